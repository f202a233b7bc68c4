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
