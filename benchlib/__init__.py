"""benchlib — the legs of bench.py (the driver's command and its ONE JSON line live in /bench.py; this package holds the multi-GPU legs, the launcher / supervisor,
the non-headline workloads and the host-side baselines they report)."""
