"""benchlib.common — what every leg of bench.py shares: the roofline peaks, hipGraph capture, the ONE-JSON-line plumbing (straight to stdout, or as records
to a supervisor's pipe), the PMC traffic table, and the test hooks (environment only: PQ_BENCH_TEST_HOOKS, read by tests/ — never a command-line flag)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_INT8_TOPS = 5033.0     # 256 CU x 4 SIMD x 2048 int8 op/clk x 2.4 GHz (MI355X_MICROARCH.md:28-34,435)
PEAK_HBM_GBS = 8000.0       # spec; 6290 GB/s measured copy (MI355X_MICROARCH.md:36)


def test_hooks():
    """PQ_BENCH_TEST_HOOKS="supervise,native-crash,native-hang,leg-hang=native_overlap2" (tests only): run the tp worker under its supervisor also at one rank; the worker
    kills itself (SIGSEGV) when it reaches the native exchange; the native phase never returns; the named native leg never returns."""
    out = {}
    for item in filter(None, os.environ.get("PQ_BENCH_TEST_HOOKS", "").split(",")):
        k, _, v = item.partition("=")
        out[k.strip()] = v.strip() or True
    return out


def graph_of(fn, n, dev=None):
    """A hipGraph holding n consecutive calls of fn (launched on torch's current stream at replay)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def med(v):
    return sorted(v)[len(v) // 2]


def traffic_for(shape=None, key=None):
    """HBM bytes per launch / per step from the PMC passes (profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes).
    shape = (M, N, K) of a GEMM launch, or key = a workload name ("mlp", "llama8b", "llama70b-shard": per STEP, summed over the step's kernels).  Returns (bytes or None, source)."""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tr = json.load(open(tj))
    except Exception:
        return None, None
    if key is not None:
        w = tr.get("workload_hbm_bytes_per_step", {}).get(key)
        return (w, tr.get("source_by_workload")) if w else (None, None)
    if tuple(shape) == (4096, 4096, 4096):
        return tr.get("gemm_hbm_bytes_per_launch"), tr.get("source")
    v = tr.get("gemm_hbm_bytes_per_launch_by_shape", {}).get("x".join(str(s) for s in shape))
    return (v, tr.get("source_by_shape")) if v else (None, None)


_JSON_FD = None


def _claim_stdout():
    """The driver reads ONE JSON line from stdout.  Native libraries write there too (RCCL prints a version banner through C stdio when
    a communicator is created, flushed at exit): from here on file descriptor 1 goes to stderr, and the JSON line alone is written to
    the original stdout (emit_json)."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _pipe_fd():
    v = os.environ.get("PQ_BENCH_PIPE")
    return int(v) if v else None


def emit_json(obj, final=True):
    """The ONE JSON line.  Under a supervisor (tp runs over more than one rank: supervise()) it goes to the supervisor's pipe as a record — provisional lines too,
    so that the best line so far survives a worker that dies — and the supervisor prints the last one; otherwise straight to the original stdout."""
    fd = _pipe_fd()
    if fd is not None:
        os.write(fd, (json.dumps({"final": bool(final), "line": obj}) + "\n").encode())
        return
    if not final:
        return
    data = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def emit_marker(name):
    fd = _pipe_fd()
    if fd is not None:
        os.write(fd, (json.dumps({"marker": name}) + "\n").encode())

