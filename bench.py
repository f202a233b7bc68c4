#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X dynamic-int8 linear path.

Metric (BASELINE.json): int8 TOPS (+ HBM GB/s of the quant pass) for qlinear M=4096 N=K=4096.
One *step* = one qlinear forward over one batch of M=4096 synthetic bf16 tokens that is already
resident in HBM: K1 per-token row-quant (pq_quant_rowwise) -> K3/K4 int8 MFMA GEMM + fused dequant
epilogue (pq_qlinear_s8), weights pre-quantised per output channel (as a deployed qlinear holds them).
Both launches go through the C-ABI of libpq_hip.so on torch's current stream, captured in a hipGraph.

Protocol: W untimed warm-up steps, then warm-up BY TIME (default 1.5 s of the same steps, untimed), then R = 20
blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides; per block the max over ranks; the reported step is the
MEDIAN block (min / max alongside).  Per-kernel durations come from interleaved hipGraph replays of each kernel alone and of the
compute step, and the script asserts GEMM + K1 <= 1.05 x step.

This file holds the driver's command line, the N = 1 headline path and the dispatch; everything else lives in benchlib/ (round 6):
  benchlib/tp.py         --gpus N > 1 (launched by torch.distributed.run, one process per GPU — or by bench.py itself when no launcher set WORLD_SIZE): north_star's split — the
                         weight column-sharded over the ranks, the activation replicated, ONE all-gather of the bf16 output shards per step (strong scaling); every exchange form
                         the repo holds is a timed, VERIFIED leg (torch.distributed rows / transposed first, then libpq_rccl.so's plain gather + layout kernel, transposed shards,
                         2 / 4 / 8 overlapped row blocks, each captured whole into the step's hipGraph under its own watchdog); the headline `value` is the fastest verified leg;
                         the data-parallel figure is the extra key "dp" (`--mode dp` makes it the main line)
  benchlib/launch.py     the GPU-free per-rank supervisor (prints the best line reported so far even if a native collective kills the worker) and the self-launcher
  benchlib/workloads.py  --workload mlp | llama8b | llama8b-linears | llama70b-shard: BASELINE configs[2], [3], [4]-per-rank as 1-GPU measurements
  benchlib/cpu_baseline.py, context.py, common.py   the host-side baselines (the oracle module is handed in from HERE: the only import of oracle/ outside tests/ and smoke()),
                         the stock torch-ROCm context legs, the shared plumbing

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (GEMM kernel vs the 5.033 POPS dense int8 MFMA peak, timed live with HIP events),
`cpu_baseline` (the QSPEC pipeline around torch._int_mm on the host cores, rank 0, at every world size) and `verified`.
"""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this pool (already exported on the GPU boxes)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import PEAK_HBM_GBS, PEAK_INT8_TOPS, _claim_stdout, emit_json, graph_of, med, test_hooks, traffic_for  # noqa: E402
from benchlib.launch import relay_launch, self_launch, supervise, supervisor_verdict  # noqa: E402,F401  (re-exported: tests/test_bench_supervisor.py)


def oracle_module():
    """oracle/torch_ref.py — imported HERE and nowhere else outside tests/ and smoke(): the cpu_baseline legs get the module handed in"""
    from oracle import torch_ref
    return torch_ref


def cpu_baseline(M, N, K, budget_s=25.0):
    from benchlib.cpu_baseline import cpu_baseline as f
    return f(oracle_module(), M, N, K, budget_s=budget_s)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--mode", choices=["auto", "dp", "tp"], default="auto",
                    help="auto: N = 1 -> the qlinear on one GPU; N > 1 -> tp (north_star's column-sharded weight + RCCL all-gather); dp = replicas over tokens")
    ap.add_argument("--repeats", type=int, default=20, help="timed blocks of exactly --steps steps; the median block is reported")
    ap.add_argument("--warmup-seconds", type=float, default=1.5, help="untimed warm-up by time after the --warmup steps")
    ap.add_argument("--torch-gather", action="store_true", help="tp: exchange through torch.distributed instead of libpq_rccl.so")
    ap.add_argument("--no-dp-leg", action="store_true", help="tp: skip the extra dp figure")
    ap.add_argument("--no-consistency-check", action="store_true")
    ap.add_argument("--layers", type=int, default=32, help="llama8b workload: decoder layers (32 = the model)")
    ap.add_argument("--no-layer-fusion", action="store_true", help="llama8b workload: skip fuse_llama_layers (stock norms, separate q/k/v GEMMs)")
    ap.add_argument("--workload", choices=["qlinear", "mlp", "llama8b", "llama8b-linears", "llama70b-shard"], default="qlinear",
                    help="qlinear = BASELINE configs[1] (default, the headline); mlp = configs[2]: Llama MLP block 4096->11008->4096, seq 2048; llama8b = configs[3]: every linear of Llama-3-8B at prefill seq 4096 (linears only)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="llama70b-shard workload: the main composition only (no pairing / int8-exchange / per-shape legs): counter passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-context", action="store_true", help="skip the stock torch-ROCm legs on this GPU (torch._int_mm pipeline, bf16 linear)")
    ap.add_argument("--tokens", type=int, default=4096, help="llama8b workload: tokens per pass (4096 = BASELINE configs[3] prefill; <= 512 = decode-like, replayed from a hipGraph)")
    ap.add_argument("--norms", action="store_true", help="llama8b workload: also run the two RMSNorms of every layer, fused into the activation quantisation (rmsnorm_quantize)")
    ap.add_argument("--unfused-silu", action="store_true", help="mlp/llama8b workloads: torch silu*mul + K1 instead of the fused producer kernel")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-supervisor", action="store_true", help="tp over > 1 rank without the per-rank supervisor process (the worker prints its own line)")
    ap.add_argument("--native-timeout", type=float, default=120.0,
                    help="tp over RCCL: seconds EACH leg that drives the native exchange (libpq_rccl.so) may take; when one does not finish, the fastest verified leg "
                         "among those that did (the torch.distributed legs run first) is printed with \"native_exchange\": \"hung\" (a multi-GPU run is never lost to a hung collective)")
    return ap.parse_args()


def main():
    args = parse()
    hooks = test_hooks()           # (tests only, from the environment: benchlib.common.test_hooks)
    script = os.path.abspath(__file__)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args, script)
    tp_ranks = args.workload == "qlinear" and args.mode != "dp" and (int(os.environ.get("WORLD_SIZE", "1")) > 1 or (hooks.get("supervise") and args.mode == "tp"))
    if tp_ranks and not os.environ.get("PQ_BENCH_WORKER") and not args.no_supervisor:
        return supervise(args, script)          # (before anything here has touched the GPU: `import torch` does not)
    _claim_stdout()
    if args.workload != "qlinear":
        assert int(os.environ.get("WORLD_SIZE", "1")) == 1, f"--workload {args.workload} is a 1-GPU measurement"
        from benchlib import workloads as W
        if args.workload == "llama8b":
            return W.run_llama8b(args, oracle_module) if args.tokens > 512 else W.run_llama8b_linears(args, oracle_module)
        return {"llama70b-shard": W.run_llama70b_shard, "llama8b-linears": W.run_llama8b_linears, "mlp": W.run_mlp}[args.workload](args, oracle_module)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.mode == "tp":      # (--mode tp at world 1: the whole tp step — graph-captured RCCL exchange included — on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29561")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # N > 1: north_star's split — W column-sharded over the ranks, the activation replicated, ONE RCCL all-gather of the
    # bf16 output shards per step (strong scaling: the whole job is ONE M x N x K qlinear): run_tp.  --mode dp: every rank runs
    # the whole qlinear on its own batch (weak scaling, no collective): below; reported as the extra key "dp" in tp runs.
    mode = args.mode if args.mode != "auto" else ("tp" if world > 1 else "dp")
    if mode == "tp":
        from benchlib.tp import run_tp
        return run_tp(args, world, rank, dev, dist, cpu_fn=cpu_baseline)

    import protoquant_amd as pq
    from protoquant_amd import _lib as L
    lib = L.lib()

    M, N, K = args.M, args.N, args.K
    st = lambda: torch.cuda.current_stream().cuda_stream     # noqa: E731

    # synthetic data: seeded on the CPU generator so every box agrees; dp ranks offset the seed
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    gw = torch.Generator().manual_seed(4321)
    w = (torch.randn(N, K, generator=gw) * 0.02).to(torch.bfloat16)
    qw = pq.quantize(w.to(dev))                 # one-time weight quantisation (K1 over W's rows)
    wq, ws = qw.int_data, qw.scale
    xq = torch.empty((M, K), dtype=torch.int8, device=dev)
    xs = torch.empty((M,), dtype=torch.float32, device=dev)
    y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    wbytes = lib.pq_qlinear_workspace_bytes(M, N, K)        # 0 for the headline shape
    wsp = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)

    def k1():
        L.check(lib.pq_quant_rowwise(x.data_ptr(), 0, M, K, K, xq.data_ptr(), K, xs.data_ptr(), st()), "pq_quant_rowwise")

    def k3():
        L.check(lib.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None,
                                  y.data_ptr(), N, 0, M, N, K, wsp.data_ptr() if wbytes else None, wbytes, st()), "pq_qlinear_s8")

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the step: K1 + K3/K4 replayed from a hipGraph of S steps (host-independent)
    K_steps = max(1, args.steps)
    g_main = g_rem = None
    S = min(K_steps, 50)
    if not args.no_graph:
        try:
            g_main = graph_of(lambda: (k1(), k3()), S)
            if K_steps % S:
                g_rem = graph_of(lambda: (k1(), k3()), K_steps % S)
        except Exception as e:   # report, never silently change what is measured
            print(f"[bench] hipGraph capture failed ({e}); running eager", file=sys.stderr)
            g_main = g_rem = None

    def run_steps(n):
        """exactly n steps"""
        if g_main is not None:
            for _ in range(n // S):
                g_main.replay()
            if n % S:
                (g_rem if (g_rem is not None and n % S == K_steps % S) else graph_of(lambda: (k1(), k3()), n % S)).replay()
            return
        for _ in range(n):
            k1(); k3()

    # ---- warm-up: the W steps the caller asked for, then warm-up BY TIME (clocks and caches settle under
    # ~2 s of this very load; a fresh box otherwise times its own power ramp) — all untimed
    run_steps(args.warmup)
    fence()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.warmup_seconds:
        run_steps(K_steps)
        torch.cuda.synchronize()
    fence()

    # ---- timed: R blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides; per block the MAX over
    # ranks; the reported step time is the MEDIAN block (min and max are reported too)
    R = max(1, args.repeats)
    blocks, host_enq = [], []
    for _ in range(R):
        fence()
        t0 = time.perf_counter()
        run_steps(K_steps)
        host_enq.append(time.perf_counter() - t0)          # host time to ENQUEUE the block (the device may still be running)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        blocks.append(dt)
    blocks.sort()
    dt = blocks[len(blocks) // 2]

    # ---- verification of what was just timed (rank-local, outside the timed region): the step's y must be the bits torch's own ops give for QSPEC E1-E4 around
    # torch._int_mm on THIS GPU, from the step's own codes and scales (the int32 accumulator through hipBLASLt — exact integers, one answer; the epilogue is three
    # correctly rounded float ops, the same on every IEEE device).  The codes themselves are pinned to the CPU oracle by the -m gpu suite, not here.
    verified = None
    try:
        y.zero_(); run_steps(1); torch.cuda.synchronize()
        acc = torch._int_mm(xq, wq.t())
        y_chk = ((acc.float() * xs[:, None]) * ws[None, :]).to(torch.bfloat16)
        verified = bool(torch.equal(y_chk.view(torch.int16), y.view(torch.int16)))
        del acc, y_chk
        if not verified:
            print("[bench] WARNING: the timed step's output differs from torch._int_mm + E1-E4 on the same codes", file=sys.stderr)
    except Exception as e:      # a check that cannot run must not lose the line; it is reported as null
        print(f"[bench] verification leg failed: {e}", file=sys.stderr)

    # ---- per-kernel durations, live, on the launch stream: each kernel of the step replayed gap-free from its own hipGraph,
    # on the step's buffers, interleaved with the step graph in the same rounds (same clocks), HIP events around each replay.
    def ev_us(g, n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
        return a.elapsed_time(b) * 1e3 / n
    PG = 20
    gk1, gk3, gst = graph_of(k1, PG), graph_of(k3, PG), graph_of(lambda: (k1(), k3()), PG)
    for g_ in (gk1, gk3, gst):
        g_.replay()
    torch.cuda.synchronize()
    tk1, tk3, tst = [], [], []
    for _ in range(max(R, 20)):
        tst.append(ev_us(gst, PG)); tk3.append(ev_us(gk3, PG)); tk1.append(ev_us(gk1, PG))
    t_gemm, t_k1, t_stepc = med(tk3), med(tk1), med(tst)
    # K1 against HBM, not against the caches: replayed on ONE input, K1's 32-MB read set is served by the L2s (each XCD re-reads
    # the same eighth of x every launch, and write-through stores leave nothing dirty to evict it) and the 256-MB Infinity Cache —
    # 6.3-6.6 us, "0.95 of 8 TB/s", a cache figure.  The roofline entry therefore rotates over enough distinct input / output
    # pairs (> 600 MB in total) that every launch reads from and writes to HBM; the cache-resident replay is reported beside it.
    n_rot = max(2, -(-600 * 2**20 // (3 * M * K)))
    rot = [(torch.randn(M, K, device=dev).to(torch.bfloat16), torch.empty((M, K), dtype=torch.int8, device=dev), torch.empty((M,), dtype=torch.float32, device=dev))
           for _ in range(n_rot)]

    def k1_rot():
        for xr_, qr_, sr_ in rot:
            L.check(lib.pq_quant_rowwise(xr_.data_ptr(), 0, M, K, K, qr_.data_ptr(), K, sr_.data_ptr(), st()), "pq_quant_rowwise")
    g_rot = graph_of(k1_rot, 2)
    g_rot.replay(); torch.cuda.synchronize()
    t_k1_hot = t_k1
    t_k1 = med([ev_us(g_rot, 2 * n_rot) for _ in range(15)])
    del rot, g_rot
    # the GEMM with its weights streamed from HBM (a layer inside a model reads its weights once per pass): rotation over enough distinct
    # weight matrices that none stays in the Infinity Cache; the activation operand stays the step's (cache-resident, as in a model)
    t_gemm_hbm = t_gemm_hbm_ref = None
    try:
        n_w = max(2, -(-640 * 2**20 // (N * K)))
        wrot = [wq] + [wq.clone() for _ in range(n_w - 1)]

        def k3_rot():
            for w_ in wrot:
                L.check(lib.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), w_.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K,
                                          wsp.data_ptr() if wbytes else None, wbytes, st()), "pq_qlinear_s8")
        g_wr = graph_of(k3_rot, 1)
        g_wr.replay(); torch.cuda.synchronize()
        # interleaved with the cache-resident replay, round by round: both figures see the same clocks (measured alone, a 2-ms graph of HBM-fed launches and a
        # 1-ms graph of warm ones sit at different points of the chip's power management, and the difference reads as an "HBM penalty" that is not one)
        th, tw_ = [], []
        for _ in range(9):
            th.append(ev_us(g_wr, n_w)); tw_.append(ev_us(gk3, PG))
        t_gemm_hbm, t_gemm_hbm_ref = med(th), med(tw_)
        del wrot, g_wr
    except Exception as e:      # an extra figure must never lose the main line
        print(f"[bench] HBM-fed GEMM leg failed: {e}", file=sys.stderr)
    # the same K1 kernel on a 4x taller activation (16384 rows; 192 MiB in + out per launch, three rotating inputs): the fixed part of
    # a launch (ramp + tail, ~1.4 us) weighs less
    Mb = 4 * M
    bigs = [(torch.randn(Mb, K, device=dev).to(torch.bfloat16), torch.empty((Mb, K), dtype=torch.int8, device=dev), torch.empty((Mb,), dtype=torch.float32, device=dev))
            for _ in range(3)]

    def k1_big_rot():
        for xb_, qb_, sb_ in bigs:
            L.check(lib.pq_quant_rowwise(xb_.data_ptr(), 0, Mb, K, K, qb_.data_ptr(), K, sb_.data_ptr(), st()), "pq_quant_rowwise")
    gb_ = graph_of(k1_big_rot, 3)
    gb_.replay(); torch.cuda.synchronize()
    tb = med([ev_us(gb_, 9) for _ in range(15)])
    k1_big = {"rows": Mb, "cols": K, "avg_kernel_us": round(tb, 2), "achieved": round((3 * Mb * K + 4 * Mb) / tb / 1e3, 1), "unit": "GB/s",
              "frac": round((3 * Mb * K + 4 * Mb) / tb / 1e3 / PEAK_HBM_GBS, 4), "how": "three rotating 16384 x 4096 inputs (576 MiB per round): HBM-fed"}
    del bigs, gb_
    consistent = (t_gemm + t_k1_hot) <= 1.05 * t_stepc <= 1.05 * 1.05 * (t_gemm + t_k1)       # cache-hot K1 <= in-step K1 <= HBM-cold K1
    host_us = sorted(host_enq)[len(host_enq) // 2] / K_steps * 1e6
    step_us = dt / K_steps * 1e6
    # a step that is not replayed whole from a graph is only a device measurement while the host enqueues faster than the device runs
    host_bound = (g_main is None) and host_us > 0.9 * step_us
    if host_bound:
        print(f"[bench] WARNING: host-bound step: enqueueing takes {host_us:.1f} us per step, the step {step_us:.1f} us — this line times Python, not the device", file=sys.stderr)
        consistent = False
    msg = f"inconsistent timings: GEMM {t_gemm:.2f} us + K1 {t_k1_hot:.2f} (cache-resident) .. {t_k1:.2f} us (HBM) vs compute step {t_stepc:.2f} us"
    if world > 1 and not consistent:
        # several ranks share one host: a multi-GPU line is never lost to this check — it carries timings_consistent = false instead
        print(f"[bench] WARNING (rank {rank}): {msg}", file=sys.stderr)
    else:
        assert consistent or host_bound or args.no_consistency_check or args.share_gpu, msg

    ops_job = 2.0 * M * N * K * world                      # whole job per step
    value = ops_job * K_steps / dt / 1e12
    k1_bytes = 3 * M * K + 4 * M                           # read bf16, write s8 + one f32 per row
    gemm_bytes = M * K + N * K + 2 * M * N + 4 * (M + N)
    gemm_ops = 2.0 * M * N * K
    variant = lib.pq_gemm_variant_name(M, N, K, K, K).decode()
    kname = {"sp256": "gemm_s8_sp256 (K3+K4)", "sp128": "gemm_s8_sp256<128x256, loader/consumer> (K3+K4)", "ring128": "gemm_s8_ring128<loader/consumer> (K3+K4)",
             "skinny": "gemm_s8_skinny (K3+K4)"}.get("" if variant.startswith(("ring128x160", "ring64")) else variant.split("_")[0].split("x")[0], variant)
    if wbytes > 0:
        kname += " split-K + splitk_reduce_epilogue"

    out = {
        "metric": "int8 TOPS for qlinear M=4096 N=K=4096 (row-quant + s8 MFMA GEMM + fused dequant); HBM GB/s of the quant pass in quant_pass",
        "value": round(value, 2), "unit": "TOPS", "n_gpus": world, "steps": K_steps, "warmup": args.warmup,
        "ms_per_step": round(dt / K_steps * 1e3, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
        "config": {"workload": f"qlinear M={M} N={N} K={K} bf16-in/int8-compute/bf16-out (BASELINE configs[1])",
                   "parallelism": f"dp{world} over tokens, replicated int8 weights, no collective",
                   "launch": f"hipgraph x{S} steps/replay" if g_main is not None else "eager",
                   "collective_in_graph": None,
                   "repeats": R, "timed": f"median of {R} blocks of exactly {K_steps} steps (barrier + synchronize around each block, max over ranks)",
                   "warmup_seconds": args.warmup_seconds, "gemm_variant": variant},
        "verified": verified,
        "verified_how": "after the timed blocks: y of one more step == torch._int_mm (hipBLASLt) on the step's own codes + E1-E4 in torch ops on this GPU, bit for bit",
        "ms_per_step_min": round(blocks[0] / K_steps * 1e3, 5), "ms_per_step_max": round(blocks[-1] / K_steps * 1e3, 5),
        "roofline": {"bound": "mfma", "kernel": kname, "achieved": round(gemm_ops / t_gemm / 1e6, 1),
                     "peak": PEAK_INT8_TOPS, "unit": "TOP/s", "frac": round(gemm_ops / t_gemm / 1e6 / PEAK_INT8_TOPS, 4),
                     "avg_kernel_us": round(t_gemm, 2), "avg_kernel_us_min": round(min(tk3), 2),
                     "how": f"median of {len(tk3)} hipGraph replays of {PG} back-to-back launches, HIP events on the launch stream (includes the ~1 us kernel boundary; rocprofv3 kernel-trace: profiles/)",
                     "avg_kernel_us_weights_from_hbm": (round(t_gemm_hbm, 2) if t_gemm_hbm else None),
                     "avg_kernel_us_same_rounds_as_hbm_leg": (round(t_gemm_hbm_ref, 2) if t_gemm_hbm_ref else None),
                     "traffic": None, "algorithmic_bytes": gemm_bytes},
        "quant_pass": {"bound": "hbm", "kernel": "quant_rowwise_vec (K1)", "achieved": round(k1_bytes / t_k1 / 1e3, 1),
                       "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(k1_bytes / t_k1 / 1e3 / PEAK_HBM_GBS, 4),
                       "avg_kernel_us": round(t_k1, 2), "algorithmic_bytes": k1_bytes,
                       "how": f"gap-free hipGraph replays rotating over {n_rot} distinct input/output pairs ({n_rot * 3 * M * K // 2**20} MiB > L2 + Infinity Cache): every launch is HBM-fed",
                       "in_step_us": round(t_stepc - t_gemm, 2), "cache_resident_replay_us": round(t_k1_hot, 2),
                       "same_kernel_4x_rows": k1_big},
        "compute_step_us": round(t_stepc, 2), "timings_consistent": bool(consistent),
        "host_enqueue_us_per_step": round(host_us, 2), "host_bound": bool(host_bound),
    }
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = traffic_for((M, N, K))
    if world == 1 and not args.no_gpu_context:
        k1(); k3(); torch.cuda.synchronize()
        from benchlib.context import gpu_context
        out["gpu_context"] = gpu_context(x, wq, ws, y, med)
        t_pipe = out["gpu_context"].get("torch_rocm_int8_pipeline_us")
        if t_pipe:
            out["gpu_context"]["library_step_us"] = round(t_stepc, 2)
            out["gpu_context"]["speedup_vs_torch_rocm_int8_pipeline"] = round(t_pipe / t_stepc, 2)
            out["gpu_context"]["speedup_vs_torch_bf16_linear"] = round(out["gpu_context"]["torch_bf16_linear_us"] / t_stepc, 2)
    if rank == 0:
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(M, N, K)     # rank 0, after every timed region, at every world size
        emit_json(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
