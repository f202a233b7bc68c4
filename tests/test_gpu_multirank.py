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


NATIVE_LEGS = {"native_plain", "native_transposed", "native_overlap2", "native_overlap4", "native_overlap8"}


def test_bench_tp_step_is_one_graph_world1():
    """bench.py --mode tp on ONE GPU (a 1-rank RCCL communicator): the step the driver times at N > 1.  Every exchange form the repo holds is a leg — the plain
    gather + layout pass, transposed shards (no layout pass), 2 / 4 / 8 row blocks overlapped on the communicator's side stream — each replayed whole from a
    hipGraph (K1, the shard GEMMs and the RCCL collectives captured), each verified bit for bit against the unsharded qlinear; the headline is the fastest
    verified leg and carries compute and exchange separately."""
    env = dict(os.environ, MASTER_PORT="29563", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "tp", "--steps", "6", "--warmup", "2", "--repeats", "3",
                        "--warmup-seconds", "0.3", "--no-cpu-baseline", "--no-dp-leg"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])       # (RCCL prints its version banner to stdout as well)
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["native_exchange"] == "ok" and line["config"]["rccl_ranks"] == 1
    legs = line["legs"]
    assert set(legs) == NATIVE_LEGS | {"torch_plain", "torch_transposed"}, sorted(legs)
    for name, leg in legs.items():
        assert leg["verified"] is True and leg["value"] > 0 and leg["exchange_us"] > 0 and leg["compute_us"] > 0 and leg["modelled"]["step_us"] > 0, (name, leg)
        if name in NATIVE_LEGS:
            assert leg["collective_in_graph"] is True and "captured in the graph" in leg["launch"] and leg["host_bound"] is False, (name, leg)
            assert leg["ms_per_step"] * 1e3 >= 0.9 * leg["compute_us"]           # the exchange is in the step
    head = line["config"]["headline_leg"]
    assert line["verified"] is True and line["ms_per_step"] == min(l["ms_per_step"] for l in legs.values()) == legs[head]["ms_per_step"]
    assert line["exchange_us"] > 0 and line["compute_us"] > 0 and line["config"]["exchange"] == legs[head]["exchange"]
    assert line["roofline"]["frac"] > 0.3 and line["timings_consistent"] is True and line["cpu_baseline"] is None


def _bench_tp(extra, port, hooks="", cpu_baseline=False, env_extra=None, _retried=False):
    """bench.py --mode tp at world 1; `hooks` = PQ_BENCH_TEST_HOOKS (the test hooks live in the environment, not on bench.py's command line: round 6)"""
    env = dict(os.environ, MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0", PQ_BENCH_TEST_HOOKS=hooks, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "tp", "--steps", "6", "--warmup", "2", "--repeats", "3", "--warmup-seconds", "0.3",
                        *([] if cpu_baseline else ["--no-cpu-baseline"]), "--no-dp-leg", *extra], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "bad_variant_access" in r.stderr and not _retried:
        # torch.distributed's own start-up died once in ~40 runs of round 6 with `std::bad_variant_access` before the communicator existed (0 of 160 in tools/init_stress.py):
        # not this repository's code, so one retry keeps the suite from failing on it — a second failure is reported
        return _bench_tp(extra, port, hooks, cpu_baseline, env_extra, _retried=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_bench_tp_watchdog_prints_the_best_finished_leg_when_a_native_leg_hangs():
    """A multi-GPU run must never be lost to a hung collective: the torch.distributed legs are measured (and verified) before the native exchange (a second RCCL
    communicator) is touched, and every native leg runs under its own watchdog; when one does not finish in time, rank 0 prints the fastest VERIFIED leg among
    those that finished, marked "native_exchange": "hung", and the rank exits 0.  Here the hang is simulated (the native phase never returns)."""
    r, lines = _bench_tp(["--native-timeout", "8"], "29565", hooks="native-hang")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["native_exchange"] == "hung" and "fallback" in line and line["hung_leg"] == "communicator bootstrap"
    assert set(line["legs"]) == {"torch_plain", "torch_transposed"} and line["config"]["headline_leg"] in line["legs"]
    assert line["config"]["collective_in_graph"] is False and "torch.distributed" in line["config"]["exchange"] and line["verified"] is True
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] > 0 and line["scaling"] == "strong" and line["roofline"]["frac"] > 0.2
    assert "did not finish" in r.stderr


def test_bench_tp_line_says_native_ok_when_every_leg_finishes():
    r, lines = _bench_tp([], "29566")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert "fallback" not in line and "hung_leg" not in line and line["native_exchange"] == "ok"
    assert line["torch_distributed_exchange_ms_per_step"] > 0 and NATIVE_LEGS <= set(line["legs"])


def test_bench_tp_watchdog_keeps_the_native_legs_that_finished():
    """a hang in the THIRD native leg: the line carries the two native legs that finished (verified, captured in the graph) beside the torch.distributed ones, the
    headline is the fastest of them all, and the hung leg is named."""
    r, lines = _bench_tp(["--native-timeout", "10"], "29567", hooks="leg-hang=native_overlap2")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["native_exchange"] == "hung" and line["hung_leg"] == "native_overlap2" and "fallback" in line
    assert set(line["legs"]) == {"torch_plain", "torch_transposed", "native_plain", "native_transposed"}
    assert all(l["verified"] for l in line["legs"].values()) and line["verified"] is True
    assert line["ms_per_step"] == min(l["ms_per_step"] for l in line["legs"].values())


def test_bench_tp_supervisor_keeps_the_safe_line_when_the_worker_dies():
    """tp runs over more than one rank put every rank's real work into a CHILD of a GPU-free supervisor process (bench.py: supervise): a fault inside a native
    collective — a segfault, a GPU memory fault — cannot be caught in the process it kills, so the worker reports every line it could print so far through a pipe
    and the supervisor prints the last one.  Here at one rank (PQ_BENCH_TEST_HOOKS=supervise,native-crash), the worker killing itself with SIGSEGV when it reaches the native exchange: the line of
    the verified torch.distributed legs comes out, marked "native_exchange": "crashed", and the exit status is 0."""
    r, lines = _bench_tp([], "29568", hooks="supervise,native-crash")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["native_exchange"] == "crashed" and "fallback" in line and set(line["legs"]) == {"torch_plain", "torch_transposed"}
    assert line["verified"] is True and line["value"] > 0 and line["config"]["headline_leg"] in line["legs"] and "worker ended with status" in r.stderr


def test_bench_tp_survivor_line_is_complete():
    """the line a supervisor prints after its worker died in the native exchange carries ALL of `legs`, `verified`, `cpu_baseline` and `roofline` —
    rank 0 computes the host baseline between the safe legs and the native bootstrap, so every provisional line holds it.  With PQ_BENCH_STRICT_EXIT=1 the same crash
    ends with the distinct status 17 (CI), line printed all the same."""
    r, lines = _bench_tp([], "29570", hooks="supervise,native-crash", cpu_baseline=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["native_exchange"] == "crashed" and line["verified"] is True and set(line["legs"]) == {"torch_plain", "torch_transposed"}
    c, rf = line["cpu_baseline"], line["roofline"]
    assert c is not None and c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert rf["bound"] == "mfma" and rf["frac"] > 0.2 and rf["traffic"] is not None and rf["traffic"] > 6e7
    r, lines = _bench_tp([], "29571", hooks="supervise,native-crash", env_extra={"PQ_BENCH_STRICT_EXIT": "1"})
    assert r.returncode == 17 and len(lines) == 1 and json.loads(lines[0])["native_exchange"] == "crashed", (r.returncode, r.stderr[-2000:])


def test_bench_tp_watchdog_line_carries_the_cpu_baseline():
    """... and so does the line the in-process watchdog prints when the native bootstrap hangs"""
    r, lines = _bench_tp(["--native-timeout", "8"], "29572", hooks="native-hang", cpu_baseline=True)
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    line = json.loads(lines[0])
    assert line["native_exchange"] == "hung" and line["cpu_baseline"] is not None and line["cpu_baseline"]["value"] > 0 and line["verified"] is True


def test_bench_tp_supervisor_relays_the_final_line():
    r, lines = _bench_tp([], "29569", hooks="supervise")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["native_exchange"] == "ok" and "fallback" not in line and NATIVE_LEGS <= set(line["legs"]) and line["verified"] is True
