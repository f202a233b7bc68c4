"""-m gpu: the native exchange (libpq_rccl.so) at world 1 in its graph-captured forms (over REAL ranks, one child process per GPU: tests/test_zz_gpu_real_ranks.py —
it runs last, so that a first multi-GPU box cannot stop the rest of the suite under -x):
the whole tp step of bench.py (K1 + shard GEMM + RCCL all-gather + layout pass in one hipGraph) and the row-chunked exchange on the
communicator's side stream (pq_allgather_cols_rows_async / pq_comm_join) under stream capture."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pq():
    import protoquant_amd
    from protoquant_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return protoquant_amd


def test_overlapped_exchange_under_graph_capture_world1(pq):
    """pq_allgather_cols_rows_async + pq_comm_join captured into a hipGraph (the side stream forks from and joins the capturing
    stream through the communicator's events): three row blocks per forward, replayed, bit-identical to the plain qlinear."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        gather = pq.RcclColumnGather()
        torch.manual_seed(9)
        lin = torch.nn.Linear(512, 640, bias=True, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(900, 512, device="cuda", dtype=torch.bfloat16)
        y0 = pq.qlinear.from_linear(lin)(x)
        m = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather, overlap_chunks=3)
        m(x); torch.cuda.synchronize()                       # allocates the exchange workspace outside the capture
        out = torch.empty_like(y0)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out.copy_(m(x))
            for _ in range(4):
                out.zero_()
                g.replay()
            torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), y0.view(torch.int16))
        y1 = m(x)                                            # the eager form still works after a capture used the same events
        torch.cuda.synchronize()
        assert torch.equal(y1.view(torch.int16), y0.view(torch.int16))
        gather.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_bench_tp_step_is_one_graph_world1():
    """bench.py --mode tp on ONE GPU (a 1-rank RCCL communicator): the step the driver times at N > 1 — K1, the shard GEMM, the RCCL
    all-gather and the layout pass — is replayed whole from a hipGraph, and the line carries compute and exchange separately."""
    env = dict(os.environ, MASTER_PORT="29563", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "tp", "--steps", "6", "--warmup", "2", "--repeats", "3",
                        "--warmup-seconds", "0.3", "--no-cpu-baseline", "--no-gpu-context", "--no-dp-leg"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])       # (RCCL prints its version banner to stdout as well)
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["collective_in_graph"] is True, line["config"]
    assert line["config"]["rccl_ranks"] == 1 and "captured in the graph" in line["config"]["launch"]
    assert line["host_bound"] is False and line["exchange_us"] > 0 and line["compute_us"] > 0
    assert line["ms_per_step"] * 1e3 >= 0.9 * line["compute_us"]                 # the exchange is in the step
    assert line["roofline"]["frac"] > 0.3 and line["timings_consistent"] is True


def _bench_tp(extra, port):
    env = dict(os.environ, MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "tp", "--steps", "6", "--warmup", "2", "--repeats", "3", "--warmup-seconds", "0.3",
                        "--no-cpu-baseline", "--no-gpu-context", "--no-dp-leg", "--safety-net", *extra], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_bench_tp_safety_net_prints_the_torch_distributed_line_when_the_native_path_hangs():
    """A multi-GPU run must never be lost to a hung collective: before the native exchange (a second RCCL communicator, captured into the step graph) is tried,
    bench.py measures the same step with torch.distributed's all-gather; if the native path does not produce its line in time, a watchdog prints that one and the
    rank exits 0.  Here the hang is simulated (the native phase never returns)."""
    r, lines = _bench_tp(["--simulate-native-hang", "--native-timeout", "8"], "29565")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert "fallback" in line and line["config"]["collective_in_graph"] is False and line["config"]["exchange"].startswith("torch.distributed")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] > 0 and line["scaling"] == "strong" and line["roofline"]["frac"] > 0.2
    assert "did not finish" in r.stderr


def test_bench_tp_safety_net_stays_silent_when_the_native_path_finishes():
    r, lines = _bench_tp([], "29566")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert "fallback" not in line and line["config"]["collective_in_graph"] is True
    assert line["torch_distributed_exchange_ms_per_step"] > 0
