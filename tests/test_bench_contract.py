"""GPU: bench.py prints exactly one JSON line that carries the contract's keys (plus `roofline` and `cpu_baseline`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_headline_json_contract():
    d = _run("--steps", "40", "--warmup", "10")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["unit"] == "TOPS" and d["dtype"] == "s8" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2.0 * 4096**3 / (d["ms_per_step"] * 1e-3) / 1e12) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TOP/s" and r["peak"] == 5033.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.2 < r["frac"] < 1.0 and (r["traffic"] is None or r["traffic"] > 6e7)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "TOPS" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


@pytest.mark.parametrize("args,unit", [(("--workload", "mlp", "--steps", "20", "--warmup", "3"), "TOPS"),
                                       (("--workload", "llama8b", "--tokens", "16", "--steps", "3", "--norms"), "TB/s")])
def test_optional_workloads_run(args, unit):
    d = _run(*args)
    assert d["unit"] == unit and d["value"] > 0 and "roofline" in d and "workload" in d["config"]
