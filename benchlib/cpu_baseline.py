"""benchlib.cpu_baseline — "protoquant's own CPU path" timed on the GPU box's host cores: the QSPEC pipeline around torch._int_mm.  The oracle module is HANDED IN by
bench.py (the only file outside tests/ and smoke() that imports oracle/): every function takes `R` = oracle.torch_ref.  A reported baseline, never the target, never the product."""
import os
import time

import torch

from .common import host_cpu_model


def _sweep_threads():
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return avail, sorted({t for t in (1, 8, 16, 32, avail) if t <= avail})


def cpu_baseline(R, M, N, K, budget_s=25.0):
    """'protoquant's own CPU path': QSPEC around torch._int_mm on this box's host cores (oracle/torch_ref.py), the same
    M x N x K bf16 qlinear as the GPU step.  Thread sweep {1, 8, 16, 32, cores this process may run on}: torch's default
    (every core of the machine) oversubscribes whatever the container is granted and ran SLOWER than one thread in round 1,
    so the stated baseline is the best of the sweep, with per-stage times (min and median) at that setting."""
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    wq, ws = R.quantize_ref(w, 1)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sweep = sorted({t for t in (1, 8, 16, 32, avail) if t <= avail})
    default_threads = torch.get_num_threads()
    ops = 2.0 * M * N * K
    t_start = time.perf_counter()
    rows = {}

    def stages():
        t0 = time.perf_counter(); xq, xs = R.quantize_ref(x, 1)
        t1 = time.perf_counter(); acc = R.int_gemm_ref(xq, wq)
        t2 = time.perf_counter(); R.epilogue_ref(acc, xs, ws, None, x.dtype)
        t3 = time.perf_counter()
        return (t1 - t0, t2 - t1, t3 - t2, t3 - t0)
    try:
        for nt in sweep:
            torch.set_num_threads(nt)
            stages()                                   # warm-up (thread pool, oneDNN primitive cache)
            reps = []
            while len(reps) < 5 and (len(reps) < 2 or time.perf_counter() - t_start < budget_s * (sweep.index(nt) + 1) / len(sweep)):
                reps.append(stages())
            tot = sorted(r[3] for r in reps)
            rows[nt] = {"reps": len(reps), "ms_median": round(tot[len(tot) // 2] * 1e3, 2), "ms_min": round(tot[0] * 1e3, 2),
                        "stage_ms_min": {"quantize": round(min(r[0] for r in reps) * 1e3, 2), "int_mm": round(min(r[1] for r in reps) * 1e3, 2),
                                         "epilogue": round(min(r[2] for r in reps) * 1e3, 2)},
                        "stage_ms_median": {"quantize": round(sorted(r[0] for r in reps)[len(reps) // 2] * 1e3, 2),
                                            "int_mm": round(sorted(r[1] for r in reps)[len(reps) // 2] * 1e3, 2),
                                            "epilogue": round(sorted(r[2] for r in reps)[len(reps) // 2] * 1e3, 2)},
                        "tops_median": round(ops / tot[len(tot) // 2] / 1e12, 4)}
    finally:
        torch.set_num_threads(default_threads)
    best = min(rows, key=lambda t: rows[t]["ms_median"])
    return {"value": rows[best]["tops_median"], "unit": "TOPS", "cores": best, "kind": "port",
            "sample": f"{rows[best]['reps']} reps of the full {M}x{N}x{K} bf16 qlinear (quantize + torch._int_mm + epilogue) per thread count, median; best of the sweep",
            "ms_per_step": rows[best]["ms_median"], "ms_per_step_min": rows[best]["ms_min"],
            "stage_ms_min": rows[best]["stage_ms_min"], "stage_ms_median": rows[best]["stage_ms_median"],
            "primitive": "torch._int_mm (oneDNN s8s8s32) + torch float ops", "host_cpu": host_cpu_model(),
            "cores_available": avail, "torch_default_threads": default_threads,
            "thread_sweep_tops_median": {str(t): rows[t]["tops_median"] for t in rows},
            "thread_sweep_ms_median": {str(t): rows[t]["ms_median"] for t in rows},
            "value_1_thread": rows[1]["tops_median"] if 1 in rows else None}



def cpu_baseline_pipeline(R, build, ops_per_run, what, budget_s=20.0, scale=1.0, scale_note=None, unit="TOPS"):
    """A multi-GEMM workload on the host: `build()` returns a zero-argument callable that runs ONE sample of the workload through the oracle's stages (its inputs prepared
    outside the timed region).  Thread sweep as in cpu_baseline; median of up to 3 runs per thread count inside the budget; value = ops_per_run / median.  `scale` (layers of a model
    extrapolated from one: stated in `sample`) multiplies nothing in the rate — it is reported so that the per-step time can be read off: ms_per_step = scale x the sample's time.  unit "TB/s": ops_per_run is
    a byte count (weight streaming of a decode-like pass); either way value = ops_per_run / median / 1e12."""
    avail, sweep = _sweep_threads()
    sweep = [t for t in sweep if t >= min(16, avail)]          # (one thread on a multi-GEMM sample would spend the whole budget: 0.1 TOPS)
    default_threads = torch.get_num_threads()
    run = build()
    rows, t_start = {}, time.perf_counter()
    try:
        for nt in sweep:
            torch.set_num_threads(nt)
            run()                                      # warm-up (thread pool, oneDNN primitive cache)
            reps = []
            while len(reps) < 3 and (len(reps) < 1 or time.perf_counter() - t_start < budget_s * (sweep.index(nt) + 1) / len(sweep)):
                t0 = time.perf_counter(); run(); reps.append(time.perf_counter() - t0)
            reps.sort()
            rows[nt] = {"reps": len(reps), "s_median": reps[len(reps) // 2], "tops_median": round(ops_per_run / reps[len(reps) // 2] / 1e12, 4)}
    finally:
        torch.set_num_threads(default_threads)
    best = min(rows, key=lambda t: rows[t]["s_median"])
    return {"value": rows[best]["tops_median"], "unit": unit, "cores": best, "kind": "port",
            "sample": what + (f"; {scale_note}" if scale_note else "") + f"; {rows[best]['reps']} timed runs per thread count, median; best of the sweep",
            "ms_per_sample": round(rows[best]["s_median"] * 1e3, 2), "ms_per_step": round(rows[best]["s_median"] * scale * 1e3, 2), "samples_per_step": scale,
            "primitive": "torch._int_mm (oneDNN s8s8s32) + torch float ops (oracle/torch_ref.py)", "host_cpu": host_cpu_model(), "cores_available": avail,
            "torch_default_threads": default_threads, "thread_sweep_tops_median": {str(t): rows[t]["tops_median"] for t in rows}}


def _w(R, n, k, g):
    return R.quantize_ref((torch.randn(n, k, generator=g) * 0.02).to(torch.bfloat16), 1)


def mlp_block(R, M=2048, H=4096, I=11008):
    """BASELINE configs[2] on the host: quantize(x) -> gate+up (one 2I-wide torch._int_mm) -> epilogue -> F.silu(g) * u -> quantize -> down -> epilogue."""
    def build():
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(M, H, generator=g).to(torch.bfloat16)
        (wgu, sgu), (wd, sd) = _w(R, 2 * I, H, g), _w(R, H, I, g)

        def run():
            y, _, _, _ = R.qlinear_ref(x, wgu, sgu)
            hq, hs, _ = R.silu_mul_quantize_ref(y[:, :I], y[:, I:])
            return R.epilogue_ref(R.int_gemm_ref(hq, wd), hs, sd, None, x.dtype)
        return run
    ops = 2.0 * M * (2 * I) * H + 2.0 * M * H * I
    return build, ops, f"the whole block at M = {M}: quantize, gate+up {M}x{2 * I}x{H}, silu*mul + quantize, down {M}x{H}x{I} (every stage through oracle/torch_ref.py)"


def llama_layer(R, M, H, I, n_qkv, n_o, n_gu, n_down, k_down, norms=True, what=""):
    """One decoder layer's int8 linear path on the host (the shapes of Llama-3-8B, or of one rank's 70B shards): [RMSNorm ->] quantize -> fused q/k/v; quantize -> o;
    [RMSNorm ->] quantize -> fused gate+up; silu*mul -> quantize -> down.  Attention itself is not part of the linear path (as in the GPU figure)."""
    def build():
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(M, H, generator=g).to(torch.bfloat16)
        a = torch.randn(M, n_o[1], generator=g).to(torch.bfloat16)           # stands for the attention output that feeds o
        hdn = torch.randn(M, k_down, generator=g).to(torch.bfloat16) if k_down != n_gu // 2 else None      # (sharded: down's input is the GATHERED intermediate)
        nw = torch.ones(H, dtype=torch.bfloat16)
        (wq_, sq_), (wo_, so_), (wg_, sg_), (wd_, sd_) = _w(R, n_qkv, H, g), _w(R, n_o[0], n_o[1], g), _w(R, n_gu, H, g), _w(R, n_down, k_down, g)

        def run():
            R.qlinear_ref(R.rmsnorm_eager_ref(x, nw, 1e-5) if norms else x, wq_, sq_)
            R.qlinear_ref(a, wo_, so_)
            y, _, _, _ = R.qlinear_ref(R.rmsnorm_eager_ref(x, nw, 1e-5) if norms else x, wg_, sg_)
            if hdn is None:
                hq, hs, _ = R.silu_mul_quantize_ref(y[:, :n_gu // 2], y[:, n_gu // 2:])
            else:
                R.silu_mul_ref(y[:, :n_gu // 2], y[:, n_gu // 2:])
                hq, hs = R.quantize_ref(hdn, 1)
            return R.epilogue_ref(R.int_gemm_ref(hq, wd_), hs, sd_, None, x.dtype)
        return run
    ops = 2.0 * M * (n_qkv * H + n_o[0] * n_o[1] + n_gu * H + n_down * k_down)
    return build, ops, what
