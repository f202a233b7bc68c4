"""-m gpu: the native exchange (libpq_rccl.so) over REAL ranks — one child process per GPU — when the box has at least two GPUs (skipped on a 1-GPU box: there the
same code runs at world 1, tests/test_gpu_multirank.py and test_gpu_sharded_forms.py).  The file sorts LAST on purpose: no RCCL communicator with more than one
rank has run in the builder's hands, and `pytest -x` must not lose the rest of the suite to the first box that has the GPUs."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(world, port, timeout=300):
    """One child per rank, output to temporary files (a PIPE nobody reads fills up and stalls its rank).  All children are polled TOGETHER: as soon as one
    exits non-zero the others — which would otherwise sit in the collective until the timeout, holding the GPUs — are killed."""
    import tempfile
    import time
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        f = tempfile.TemporaryFile(mode="w+")
        files.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_rank_worker.py")], env=env, stdout=f, stderr=subprocess.STDOUT, text=True))
    t0 = time.time()
    failed = timed_out = False
    while True:
        codes = [p.poll() for p in procs]
        failed = any(c not in (None, 0) for c in codes)
        timed_out = time.time() - t0 > timeout
        if failed or timed_out or all(c is not None for c in codes):
            break
        time.sleep(0.2)
    if failed or timed_out:
        time.sleep(1.0)                      # let the other ranks print their own error, if they have one
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs = []
    for p, f in zip(procs, files):
        p.wait()
        f.seek(0)
        outs.append((p.returncode, f.read()))
        f.close()
    if timed_out and not failed:
        raise subprocess.TimeoutExpired("rccl_rank_worker.py", timeout, output="\n".join(o[-1500:] for _, o in outs))
    return outs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_native_exchange_over_real_ranks(world):
    """RcclColumnGather (equal and ragged shards; whole, row-chunked overlapped, transposed; graph-captured) and RcclRowReduceScatter
    across `world` processes, one per GPU, each result compared with the unsharded qlinear on its own rank."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    outs = _spawn(world, 29570 + world)
    for r, (rc, o) in enumerate(outs):
        assert rc == 0 and f"OK {r}" in o, f"rank {r} failed (rc {rc}):\n{o[-3000:]}"
