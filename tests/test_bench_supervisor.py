"""CPU (-m "not gpu"): what a rank's supervisor process of `bench.py --gpus N` (bench.supervise) does with its worker's reports — the decision function alone, no GPU:
the last reported line wins, a non-final line is marked crashed, a worker that died after the safe (torch.distributed) legs does not fail the rank, one that died before does."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rec(**kw):
    return json.dumps(kw) + "\n"


def test_supervisor_verdict():
    import bench
    safe = _rec(marker="safe")
    prov = _rec(final=False, line={"value": 1.0, "native_exchange": "not_attempted", "legs": {"torch_plain": {}}})
    prov2 = _rec(final=False, line={"value": 2.0, "native_exchange": "ok", "legs": {"torch_plain": {}, "native_plain": {}}})
    fin = _rec(final=True, line={"value": 3.0, "native_exchange": "ok"})
    # a normal end: the final line, untouched
    line, code, _ = bench.supervisor_verdict([safe, prov, prov2, fin], 0)
    assert (line["value"], line["native_exchange"], code) == (3.0, "ok", 0) and "fallback" not in line
    # killed inside a native leg (SIGSEGV = -11) after two provisional lines: the LAST provisional line, marked, and the rank does not fail
    line, code, was_safe = bench.supervisor_verdict([safe, prov, prov2, '{"final": false, "line": {"val'], -11)
    assert (line["value"], line["native_exchange"], code, was_safe) == (2.0, "crashed", 0, True) and "status -11" in line["fallback"]
    # the watchdog's line is final (it says "hung" itself) even though the worker then leaves through os._exit
    hung = _rec(final=True, line={"value": 1.0, "native_exchange": "hung", "hung_leg": "native_plain"})
    line, code, _ = bench.supervisor_verdict([safe, prov, hung], 0)
    assert (line["native_exchange"], code) == ("hung", 0)
    # died before the safe legs were in: nothing to print, the rank fails with the worker's status (a signal maps to 1)
    assert bench.supervisor_verdict([], 3) == (None, 3, False)
    assert bench.supervisor_verdict(["garbage\n"], -9) == (None, 1, False)
    # ranks other than 0 send only the marker: no line, but a crash after it does not fail the rank
    assert bench.supervisor_verdict([safe], -11) == (None, 0, True)


def test_survivor_line_keeps_cpu_baseline_and_roofline():
    """rank 0 reports the host baseline BEFORE the native bootstrap, so the provisional line a supervisor prints for a crashed worker is a complete
    line: non-null `cpu_baseline` and `roofline`, marked crashed, exit 0 — or the distinct status 17 under PQ_BENCH_STRICT_EXIT=1."""
    import bench
    safe = _rec(marker="safe")
    bare = _rec(final=False, line={"value": 1.0, "verified": True, "legs": {"torch_plain": {}}, "roofline": {"frac": 0.5, "traffic": 1}, "cpu_baseline": None})
    full = _rec(final=False, line={"value": 1.0, "verified": True, "legs": {"torch_plain": {}}, "roofline": {"frac": 0.5, "traffic": 1}, "cpu_baseline": {"value": 1.1, "cores": 16, "kind": "port"}})
    line, code, was_safe = bench.supervisor_verdict([safe, bare, full], -11)
    assert code == 0 and was_safe and line["native_exchange"] == "crashed"
    assert line["cpu_baseline"]["value"] == 1.1 and line["roofline"]["frac"] == 0.5 and line["verified"] is True and "legs" in line
    os.environ["PQ_BENCH_STRICT_EXIT"] = "1"
    try:
        line, code, _ = bench.supervisor_verdict([safe, bare, full], -11)
        assert code == 17 and line["cpu_baseline"] is not None
        assert bench.supervisor_verdict([safe, full, _rec(final=True, line={"value": 2.0})], 0)[1] == 0          # a clean end stays 0
    finally:
        del os.environ["PQ_BENCH_STRICT_EXIT"]


def test_self_launch_prints_the_line_of_a_launcher_that_failed():
    """torch.distributed.run ends non-zero when any rank does; if exactly one JSON line came back it is printed and the launcher's status is passed on —
    only an absent line is fatal."""
    import subprocess
    import bench
    assert bench.relay_launch(1, "banner\n{\"value\": 2.0}\n", 8) == ('{"value": 2.0}', 1)
    assert bench.relay_launch(0, "{\"value\": 2.0}\n", 8) == ('{"value": 2.0}', 0)
    assert bench.relay_launch(1, "no json here\n", 8) == (None, 1)
    assert bench.relay_launch(0, "", 8) == (None, 1)
    assert bench.relay_launch(0, "{\"a\": 1}\n{\"b\": 2}\n", 8)[0] is None                                 # ambiguous: fatal
    # end to end with a stub launcher that prints one line and exits 1 (no GPU, no torch.distributed)
    stub = "import sys; print('RCCL banner'); print('{\"value\": 3.0, \"native_exchange\": \"crashed\"}'); sys.exit(1)"
    code = ("import sys, types; sys.argv = ['bench.py', '--gpus', '2']; import bench; "
            f"bench.self_launch(types.SimpleNamespace(gpus=2), 'bench.py', launcher_cmd=[sys.executable, '-c', {stub!r}])")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 1 and r.stdout.strip() == '{"value": 3.0, "native_exchange": "crashed"}', (r.returncode, r.stdout, r.stderr[-500:])
    assert "RCCL banner" in r.stderr and "rank 0's line came back" in r.stderr


def test_bench_py_is_the_thin_front_end():
    """the driver's file stays small and holds no test hook on its command line"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert len(src.splitlines()) <= 400 and "--simulate" not in src and "add_argument(\"--supervise\"" not in src
    hooks_users = [f for f in os.listdir(os.path.join(ROOT, "benchlib")) if f.endswith(".py") and "PQ_BENCH_TEST_HOOKS" in open(os.path.join(ROOT, "benchlib", f)).read()]
    assert hooks_users == ["common.py"]


def test_only_bench_py_imports_the_oracle_outside_tests():
    """the oracle is test infrastructure: bench.py's cpu_baseline legs get the module handed in (bench.oracle_module); benchlib/ itself never imports it"""
    for f in os.listdir(os.path.join(ROOT, "benchlib")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "benchlib", f)).read()
            assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
