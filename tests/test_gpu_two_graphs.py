"""-m gpu: a hipGraph around a fused split-K launch must keep replaying after OTHER graphs have been captured (round 6: with the tickets zeroed by a hipMemsetAsync node it hung
on the next replay — profiles/r06_hipgraph_memset_hang.txt; the library zeroes them with a kernel of its own now, and holds no hipMemset* call).  The scenario runs in a child
process under a timeout: the failure mode is a hang."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_splitk_graph_survives_later_captures():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "two_graphs_worker.py")], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "OK two graphs" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
