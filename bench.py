#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X dynamic-int8 linear path.

Metric (BASELINE.json): int8 TOPS (+ HBM GB/s of the quant pass) for qlinear M=4096 N=K=4096.
One *step* = one qlinear forward over one batch of M=4096 synthetic bf16 tokens that is already
resident in HBM: K1 per-token row-quant (pq_quant_rowwise) -> K3/K4 int8 MFMA GEMM + fused dequant
epilogue (pq_qlinear_s8), weights pre-quantised per output channel (as a deployed qlinear holds them).
Both launches go through the C-ABI of libpq_hip.so on torch's current stream, captured in a hipGraph.

Multi-GPU (--gpus N, launched by torch.distributed.run, one process per GPU): data-parallel over
tokens — every rank runs the same step on its own 4096-token batch with replicated weights, no
data-path collective (weak scaling).  `--mode tp` instead column-shards the weight (N/G output
channels per rank) and all-gathers the transposed output shards with RCCL (BASELINE config 5 layout).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (GEMM kernel vs the
5.033 POPS dense int8 MFMA peak, timed live with HIP events) and `cpu_baseline` (the QSPEC pipeline
around torch._int_mm on the host cores, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this pool (already exported on the GPU boxes)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5033.0     # 256 CU x 4 SIMD x 2048 int8 op/clk x 2.4 GHz (MI355X_MICROARCH.md:28-34,435)
PEAK_HBM_GBS = 8000.0       # spec; 6290 GB/s measured copy (MI355X_MICROARCH.md:36)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--mode", choices=["dp", "tp"], default="dp")
    ap.add_argument("--workload", choices=["qlinear", "mlp", "llama8b"], default="qlinear",
                    help="qlinear = BASELINE configs[1] (default, the headline); mlp = configs[2]: Llama MLP block 4096->11008->4096, seq 2048; llama8b = configs[3]: every linear of Llama-3-8B at prefill seq 4096 (linears only)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-steps", type=int, default=10, help="steps captured per hipGraph replay (dp mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tp-chunks", type=int, default=1, help="--mode tp: row blocks whose all-gathers overlap the next block's GEMM")
    ap.add_argument("--tokens", type=int, default=4096, help="llama8b workload: tokens per pass (4096 = BASELINE configs[3] prefill; <= 512 = decode-like, replayed from a hipGraph)")
    ap.add_argument("--norms", action="store_true", help="llama8b workload: also run the two RMSNorms of every layer, fused into the activation quantisation (rmsnorm_quantize)")
    ap.add_argument("--unfused-silu", action="store_true", help="mlp/llama8b workloads: torch silu*mul + K1 instead of the fused producer kernel")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def ev_time_us(fn, iters):
    """Average duration of `fn` (one kernel launch) over `iters` back-to-back launches, HIP events on
    the launch stream."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        fn()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def cpu_baseline(M, N, K):
    """'protoquant's own CPU path': QSPEC around torch._int_mm on this box's host cores (oracle/torch_ref.py).
    Bounded sample: a few repetitions of the same M x N x K qlinear."""
    from oracle import torch_ref as R
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    wq, ws = R.quantize_ref(w, 1)
    reps, ts = 6, []
    R.qlinear_ref(x, wq, ws, None)
    t_all = time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        R.qlinear_ref(x, wq, ws, None)
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 25:
            break
    ts.sort()
    med = ts[len(ts) // 2]
    # the same qlinear on ONE host thread (SURVEY §8d asks for both): two repetitions, best of
    nthreads = torch.get_num_threads()
    one = None
    try:
        torch.set_num_threads(1)
        t1 = []
        for _ in range(2):
            t0 = time.perf_counter()
            R.qlinear_ref(x, wq, ws, None)
            t1.append(time.perf_counter() - t0)
        one = min(t1)
    finally:
        torch.set_num_threads(nthreads)
    return {"value": round(2.0 * M * N * K / med / 1e12, 4), "unit": "TOPS", "cores": nthreads,
            "kind": "port", "sample": f"{len(ts)} reps of the full {M}x{N}x{K} bf16 qlinear (quant+_int_mm+epilogue), median",
            "ms_per_step": round(med * 1e3, 2), "primitive": "torch._int_mm (oneDNN s8s8s32) + torch float ops",
            "value_1_thread": round(2.0 * M * N * K / one / 1e12, 4) if one else None,
            "ms_per_step_1_thread": round(one * 1e3, 1) if one else None}


def run_mlp(args):
    """BASELINE configs[2]: gate/up (horizontally fused, N = 2 x 11008) and down as qlinear at M = 2048; silu(g)*u is
    fused into the quantisation of down's input (pq_silu_mul_quant_rowwise; --unfused-silu restores the stock torch-ROCm
    elementwise + K1 pair).  One step = the whole block; 554.05 GOP of int8 GEMM."""
    import protoquant_amd as pq
    M, H, I = 2048, 4096, 11008
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(M, H, generator=g).to(torch.bfloat16).to(dev)
    mk = lambda o, i: pq.qlinear.from_linear(torch.nn.Linear(i, o, bias=False, dtype=torch.bfloat16).to(dev))
    gate_up = pq.FusedQLinear([mk(I, H), mk(I, H)])
    down = mk(H, I)

    if args.unfused_silu:
        def block():
            gt, up = gate_up(x)
            return down(torch.nn.functional.silu(gt) * up)
    else:
        mlp = pq.GatedMLP(gate_up, down)       # silu*mul fused into the quantisation of down's input (one pass, no bf16 h)

        def block():
            return mlp(x)

    for _ in range(args.warmup):
        block()
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        try:
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                block()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                block()
        except Exception as e:
            print(f"[bench] hipGraph capture failed ({e}); running eager", file=sys.stderr)
            graph = None
    run = (lambda: graph.replay()) if graph is not None else block
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops = 2.0 * M * (2 * I) * H + 2.0 * M * H * I
    print(json.dumps({"metric": "int8 TOPS, Llama MLP block (gate/up/down as qlinear)", "value": round(ops * args.steps / dt / 1e12, 2),
                      "unit": "TOPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                      "config": {"workload": "Llama MLP block 4096->11008->4096, seq 2048, gate+up fused (BASELINE configs[2])",
                                 "silu_mul": "torch elementwise + K1" if args.unfused_silu else "fused into K1 (pq_silu_mul_quant_rowwise)",
                                 "launch": "hipgraph" if graph is not None else "eager"},
                      "roofline": {"bound": "mfma", "achieved": round(ops * args.steps / dt / 1e12, 1), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                                   "frac": round(ops * args.steps / dt / 1e12 / PEAK_INT8_TOPS, 4), "traffic": None,
                                   "note": "whole block incl. both activation quantisations (the second fused with silu*mul), not a single kernel"},
                      "cpu_baseline": None}), flush=True)


def run_llama8b(args):
    """BASELINE configs[3], linears only: 32 layers x {fused qkv 4096->6144, o 4096->4096, fused gate+up 4096->28672,
    down 14336->4096} + lm_head 4096->128256 at M = 4096 tokens (bs 1, seq 4096): 61.48 TOP of int8 GEMM per pass.
    Synthetic int8 weights (gaussian codes) and scales; every activation quantisation (K1) is included; attention,
    norms and rope are NOT run (the o-projection input is a slice of the qkv output, the down input is silu(g)*u)."""
    import protoquant_amd as pq
    from protoquant_amd.qtensor import QTensor
    dev = torch.device("cuda", 0)
    M, H, I, V, L = args.tokens, 4096, 14336, 128256, 32

    def mkq(n, k):
        wq = (torch.randn(n, k, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
        return pq.qlinear.from_qtensor(QTensor(wq, torch.rand(n, device=dev) * 1e-3 + 1e-4, 1, torch.bfloat16, wq.shape))

    layers = [(mkq(6144, H), mkq(H, H), mkq(2 * I, H), mkq(H, I)) for _ in range(L)]
    head = mkq(V, H)
    x0 = torch.randn(M, H, device=dev).to(torch.bfloat16)

    norm_w = torch.ones(H, device=dev, dtype=torch.bfloat16)

    def fwd():
        x = x0
        for qkv, o, gu, down in layers:
            # --norms: the layer's two RMSNorms, fused into the quantisation of the qkv and gate/up inputs (K1n replaces K1)
            a = qkv(pq.rmsnorm_quantize(x, norm_w, 1e-5) if args.norms else x)[:, :H]      # a strided view: K1 takes the leading dimension
            x = o(a)
            g_u = gu(pq.rmsnorm_quantize(x, norm_w, 1e-5) if args.norms else x)
            if args.unfused_silu:
                x = down(torch.nn.functional.silu(g_u[:, :I]) * g_u[:, I:])
            else:
                x = down(pq.silu_mul_quantize(g_u[:, :I], g_u[:, I:]))
        return head(x)

    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    run, launch = fwd, "eager"
    if M <= 512 and not args.no_graph:        # decode-sized passes are launch-bound from Python: replay them from a hipGraph
        s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            fwd()
        torch.cuda.current_stream().wait_stream(s_)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fwd()
        run, launch = gr.replay, "hipgraph"
        for _ in range(3):
            run()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ops = L * (2.0 * M * 6144 * H + 2.0 * M * H * H + 2.0 * M * 2 * I * H + 2.0 * M * H * I) + 2.0 * M * V * H
    wbytes = L * (6144 * H + H * H + 2 * I * H + H * I) + V * H          # int8 weight bytes streamed per pass
    if M <= 512:       # decode-like: the pass is a streaming read of the weights — report it against HBM, not MFMA
        print(json.dumps({"metric": f"weight-streaming TB/s, Llama-3-8B linears at {M} tokens (decode-like)", "value": round(wbytes / dt / 1e12, 3),
                          "unit": "TB/s", "n_gpus": 1, "steps": args.steps, "warmup": 2, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                          "config": {"workload": f"Llama-3-8B every linear as qlinear (qkv and gate/up fused) at {M} tokens, linears + quant passes only",
                                     "rmsnorm": "fused into K1 (pq_rmsnorm_quant_rowwise)" if args.norms else "not run", "launch": launch,
                                     "weight_bytes_per_step": wbytes, "int8_tops": round(ops / dt / 1e12, 2)},
                          "roofline": {"bound": "hbm", "achieved": round(wbytes / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": round(wbytes / dt / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                       "note": "whole pass (290 kernels); algorithmic bytes = the int8 weights only"},
                          "cpu_baseline": None}), flush=True)
        return
    print(json.dumps({"metric": "int8 TOPS, Llama-3-8B linears at prefill seq 4096", "value": round(ops / dt / 1e12, 2), "unit": "TOPS",
                      "n_gpus": 1, "steps": args.steps, "warmup": 2, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                      "config": {"workload": "Llama-3-8B every linear as qlinear (qkv and gate/up fused), bs 1 seq 4096, linears + quant passes only (BASELINE configs[3])",
                                 "rmsnorm": "fused into K1 (pq_rmsnorm_quant_rowwise)" if args.norms else "not run",
                                 "launch": launch, "int8_ops_per_step": ops},
                      "roofline": {"bound": "mfma", "achieved": round(ops / dt / 1e12, 1), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                                   "frac": round(ops / dt / 1e12 / PEAK_INT8_TOPS, 4), "traffic": None,
                                   "note": "whole pass incl. every activation quantisation (silu*mul fused into down's) and the strided read of the qkv slice"},
                      "cpu_baseline": None}), flush=True)


def main():
    args = parse()
    if args.workload == "llama8b":
        assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "--workload llama8b is a 1-GPU measurement"
        return run_llama8b(args)
    if args.workload == "mlp":
        assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "--workload mlp is a 1-GPU measurement"
        return run_mlp(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import protoquant_amd as pq
    from protoquant_amd import _lib as L
    lib = L.lib()

    M, N, K = args.M, args.N, args.K
    tp = args.mode == "tp" and world > 1
    n_local = N // world if tp else N
    # synthetic data: seeded on the CPU generator so every box agrees; rank offsets the seed
    g = torch.Generator().manual_seed(1234 + (0 if tp else rank))
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    gw = torch.Generator().manual_seed(4321)
    w = (torch.randn(N, K, generator=gw) * 0.02).to(torch.bfloat16)
    if tp:
        w = w[rank * n_local:(rank + 1) * n_local]
    qw = pq.quantize(w.to(dev))                 # one-time weight quantisation (K1 over W's rows)
    wq, ws = qw.int_data, qw.scale
    xq = torch.empty((M, K), dtype=torch.int8, device=dev)
    xs = torch.empty((M,), dtype=torch.float32, device=dev)
    y = torch.empty((M, n_local), dtype=torch.bfloat16, device=dev)
    y_full = torch.empty((world * M * n_local,), dtype=torch.bfloat16, device=dev) if tp else None

    def k1():
        L.check(lib.pq_quant_rowwise(x.data_ptr(), 0, M, K, K, xq.data_ptr(), K, xs.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), "pq_quant_rowwise")

    wbytes = lib.pq_qlinear_workspace_bytes(M, n_local, K)        # 0 for the headline shape; > 0 for narrow tp shards (split-K)
    wsp = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)

    def k3():
        L.check(lib.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None,
                                  y.data_ptr(), n_local, 0, M, n_local, K, wsp.data_ptr() if wbytes else None, wbytes,
                                  torch.cuda.current_stream().cuda_stream), "pq_qlinear_s8")

    # --mode tp --tp-chunks k: the rows are cut into k blocks; each block's GEMM is followed at once by its asynchronous
    # all-gather, so block i's transfer overlaps block i+1's GEMM (RCCL runs on its own stream); k = 1 is the plain step
    chunks = max(1, args.tp_chunks) if tp else 1
    bounds = [(M * c // chunks, M * (c + 1) // chunks) for c in range(chunks)]
    y_parts = [torch.empty((world * (b - a) * n_local,), dtype=torch.bfloat16, device=dev) for a, b in bounds] if tp and chunks > 1 else None

    def k3_rows(a, b):
        L.check(lib.pq_qlinear_s8(xq.data_ptr() + a * K, K, xs.data_ptr() + 4 * a, wq.data_ptr(), K, ws.data_ptr(), None,
                                  y.data_ptr() + 2 * a * n_local, n_local, 0, b - a, n_local, K, None, 0,
                                  torch.cuda.current_stream().cuda_stream), "pq_qlinear_s8")

    def step_eager():
        k1()
        if tp and chunks > 1:
            works = []
            for (a, b), part in zip(bounds, y_parts):
                k3_rows(a, b)
                works.append(dist.all_gather_into_tensor(part, y[a:b].view(-1), async_op=True))
            for wk in works:
                wk.wait()
            return
        k3()
        if tp:
            dist.all_gather_into_tensor(y_full, y.view(-1))

    # hipGraph capture: S consecutive steps (2 launches each) per replay; the collective of --mode tp stays outside.
    use_graph = not args.no_graph and chunks == 1
    S = 1 if tp else max(1, min(args.graph_steps, args.steps))
    graph = None
    if use_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                k1(); k3()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(S):
                    k1(); k3()
        except Exception as e:   # report, never silently change what is measured
            print(f"[bench] hipGraph capture failed ({e}); running eager", file=sys.stderr)
            graph = None
    if graph is None:
        S = 1

    def run_steps(n):
        """exactly n steps: n // S graph replays of S steps + the remainder eagerly"""
        if graph is not None:
            for _ in range(n // S):
                graph.replay()
                if tp:
                    dist.all_gather_into_tensor(y_full, y.view(-1))
            for _ in range(n % S):
                step_eager()
        else:
            for _ in range(n):
                step_eager()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run_steps(args.steps)
    e1.record()
    fence()
    dt = time.perf_counter() - t0
    dt_ev = e0.elapsed_time(e1) * 1e-3
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel durations, live, HIP events on the launch stream: each kernel of the step replayed back-to-back from a
    # hipGraph (host-independent), on the step's own buffers, right after the timed region
    def kernel_us(fn, per_graph=10, replays=30):
        try:
            s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                fn()
            torch.cuda.current_stream().wait_stream(s2)
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk):
                for _ in range(per_graph):
                    fn()
            for _ in range(5):
                gk.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(replays):
                gk.replay()
            b.record(); b.synchronize()
            return a.elapsed_time(b) * 1e3 / (per_graph * replays)
        except Exception as e:
            print(f"[bench] per-kernel graph timing failed ({e}); eager back-to-back launches", file=sys.stderr)
            return ev_time_us(fn, 300)
    # primary figure: eager back-to-back launches (small host gaps, like the K1 gaps of a real step; agrees with the
    # rocprofv3 kernel-trace average of this command); extra figure: gap-free graph replays (sustained, lower clocks)
    it = max(50, min(args.steps, 500))
    t_gemm = ev_time_us(k3, it)
    t_k1 = ev_time_us(k1, it)
    t_gemm_sustained = kernel_us(k3)
    torch.cuda.synchronize()

    ops_step = 2.0 * M * n_local * K                       # per rank
    total_ops = ops_step * world * args.steps
    value = total_ops / dt / 1e12
    k1_bytes = 3 * M * K + 4 * M                           # read bf16, write s8 + one f32 per row
    gemm_bytes = M * K + n_local * K + 2 * M * n_local + 4 * (M + n_local)

    out = {
        "metric": "int8 TOPS for qlinear M=4096 N=K=4096 (row-quant + s8 MFMA GEMM + fused dequant); HBM GB/s of the quant pass in quant_pass",
        "value": round(value, 2), "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True,
        "scaling": "strong" if tp else "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
        "config": {"workload": f"qlinear M={M} N={N} K={K} bf16-in/int8-compute/bf16-out (BASELINE configs[1])",
                   "parallelism": (f"tp{world} column-sharded W + RCCL all-gather" + (f", {chunks} row blocks overlapped" if chunks > 1 else "") if tp else f"dp{world} over tokens, replicated int8 weights"),
                   "launch": (f"hipgraph x{S} steps/replay" if graph is not None else "eager"),
                   "gemm_variant": lib.pq_gemm_variant_name(M, n_local, K, K, K).decode()},
        "roofline": {"bound": "mfma", "kernel": "gemm_s8_sp256 (K3+K4)", "achieved": round(2.0 * M * n_local * K / t_gemm / 1e6, 1),
                     "peak": PEAK_INT8_TOPS, "unit": "TOP/s", "frac": round(2.0 * M * n_local * K / t_gemm / 1e6 / PEAK_INT8_TOPS, 4),
                     "avg_kernel_us": round(t_gemm, 2), "avg_kernel_us_gapfree_replay": round(t_gemm_sustained, 2), "traffic": None,
                     "algorithmic_bytes": gemm_bytes},
        "quant_pass": {"bound": "hbm", "kernel": "quant_rowwise_vec (K1)", "achieved": round(k1_bytes / t_k1 / 1e3, 1),
                       "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(k1_bytes / t_k1 / 1e3 / PEAK_HBM_GBS, 4),
                       "avg_kernel_us": round(t_k1, 2), "algorithmic_bytes": k1_bytes},
        "event_ms_per_step": round(dt_ev / args.steps * 1e3, 5),
    }
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        try:
            tr = json.load(open(tj))
            out["roofline"]["traffic"] = tr.get("gemm_hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")
        except Exception:
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(M, N, K)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
