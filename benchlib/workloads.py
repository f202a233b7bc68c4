"""benchlib.workloads — the non-headline BASELINE configurations as 1-GPU measurements: configs[2] (Llama MLP block), configs[3] (Llama-3-8B, linears only / the whole model),
configs[4] per rank (one of 8 ranks of Llama-3-70B).  Each prints ONE JSON line with `roofline` (incl. PMC `traffic` per step where profiles/traffic.json holds it) and
`cpu_baseline` (the same workload through the oracle's stages on the host: benchlib.cpu_baseline; the oracle module is handed in by bench.py)."""
import sys
import time

import torch

from . import cpu_baseline as CB
from .common import PEAK_HBM_GBS, PEAK_INT8_TOPS, emit_json, graph_of, traffic_for


def _cpu(args, oracle, maker, scale=1.0, note=None, unit="TOPS", num=None):
    """rank 0's host baseline of the workload (after the timed GPU regions): maker(R) -> (build, ops, what) from benchlib.cpu_baseline"""
    if args.no_cpu_baseline or oracle is None:
        return None
    try:
        R = oracle()
        build, ops, what = maker(R)
        return CB.cpu_baseline_pipeline(R, build, ops if num is None else num, what, budget_s=20.0, scale=scale, scale_note=note, unit=unit)
    except Exception as e:          # a baseline that cannot run must not lose the line
        print(f"[bench] cpu baseline failed: {e}", file=sys.stderr)
        return None


def run_mlp(args, oracle=None):
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
    # The replayed block keeps its 135 MB of int8 weights in the 256-MB Infinity Cache; a layer inside a model reads its weights once per
    # pass, from HBM.  Extra key: the same block over 6 layers' worth of distinct weights (810 MB), one layer after the other.
    hbm_fed = None
    if not args.unfused_silu and not args.no_graph:
        try:
            layers = [mlp] + [pq.GatedMLP(pq.FusedQLinear([mk(I, H), mk(I, H)]), mk(H, I)) for _ in range(5)]

            def stack():
                for l_ in layers:
                    l_(x)
            stack(); torch.cuda.synchronize()
            s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                stack()
            torch.cuda.current_stream().wait_stream(s2)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                stack()
            for _ in range(5):
                g2.replay()
            torch.cuda.synchronize()
            n2 = max(5, args.steps // 6)
            t1 = time.perf_counter()
            for _ in range(n2):
                g2.replay()
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t1) / (n2 * len(layers))
            hbm_fed = {"ms_per_block": round(d2 * 1e3, 5), "value": round(ops / d2 / 1e12, 2), "unit": "TOPS",
                       "what": f"{len(layers)} blocks with distinct weights ({len(layers) * 135} MB) run one after the other: every block streams its weights from HBM"}
        except Exception as e:      # an extra figure must never lose the main line
            print(f"[bench] HBM-fed leg failed: {e}", file=sys.stderr)
    emit_json({"metric": "int8 TOPS, Llama MLP block (gate/up/down as qlinear)", "value": round(ops * args.steps / dt / 1e12, 2),
                      "unit": "TOPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                      "config": {"workload": "Llama MLP block 4096->11008->4096, seq 2048, gate+up fused (BASELINE configs[2])",
                                 "silu_mul": "torch elementwise + K1" if args.unfused_silu else "fused into K1 (pq_silu_mul_quant_rowwise)",
                                 "launch": "hipgraph" if graph is not None else "eager",
                                 "weights": "the one block replayed: its 135 MB of int8 weights stay in the Infinity Cache (see weights_from_hbm)"},
                      "weights_from_hbm": hbm_fed,
                      "roofline": {"bound": "mfma", "achieved": round(ops * args.steps / dt / 1e12, 1), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                                   "frac": round(ops * args.steps / dt / 1e12 / PEAK_INT8_TOPS, 4), "traffic": traffic_for(key="mlp")[0], "traffic_source": traffic_for(key="mlp")[1],
                                   "algorithmic_bytes": 2 * M * H + 3 * I * H + 2 * M * 2 * I * 2 + M * I + 2 * M * H + 4 * (2 * M + 3 * I + H),
                                   "note": "whole block (per step) incl. both activation quantisations (the second fused with silu*mul), not a single kernel; algorithmic bytes = x, the int8 weights, "
                                           "the bf16 gate/up written and read once, the int8 intermediate, y, the scale vectors"},
                      "cpu_baseline": _cpu(args, oracle, lambda R: CB.mlp_block(R, M, H, I))})


def run_llama8b_linears(args, oracle=None):
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
        emit_json({"metric": f"weight-streaming TB/s, Llama-3-8B linears at {M} tokens (decode-like)", "value": round(wbytes / dt / 1e12, 3),
                          "unit": "TB/s", "n_gpus": 1, "steps": args.steps, "warmup": 2, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                          "config": {"workload": f"Llama-3-8B every linear as qlinear (qkv and gate/up fused) at {M} tokens, linears + quant passes only",
                                     "rmsnorm": "fused into K1 (pq_rmsnorm_quant_rowwise)" if args.norms else "not run", "launch": launch,
                                     "weight_bytes_per_step": wbytes, "int8_tops": round(ops / dt / 1e12, 2)},
                          "roofline": {"bound": "hbm", "achieved": round(wbytes / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": round(wbytes / dt / 1e9 / PEAK_HBM_GBS, 4), "traffic": traffic_for(key=f"llama8b-linears-{M}")[0],
                                       "note": "whole pass (290 kernels); algorithmic bytes = the int8 weights only"},
                          "cpu_baseline": _cpu(args, oracle, lambda R: CB.llama_layer(R, M, H, I, 6144, (H, H), 2 * I, H, I, norms=args.norms,
                                                                                       what=f"ONE decoder layer's linear path at {M} tokens on the host"),
                                               scale=L, note=f"one layer x {L} (lm_head not included)", unit="TB/s", num=float(6144 * H + H * H + 2 * I * H + H * I))})
        return
    emit_json({"metric": "int8 TOPS, Llama-3-8B linears at prefill seq 4096", "value": round(ops / dt / 1e12, 2), "unit": "TOPS",
                      "n_gpus": 1, "steps": args.steps, "warmup": 2, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                      "config": {"workload": "Llama-3-8B every linear as qlinear (qkv and gate/up fused), bs 1 seq 4096, linears + quant passes only (BASELINE configs[3])",
                                 "rmsnorm": "fused into K1 (pq_rmsnorm_quant_rowwise)" if args.norms else "not run",
                                 "launch": launch, "int8_ops_per_step": ops},
                      "roofline": {"bound": "mfma", "achieved": round(ops / dt / 1e12, 1), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                                   "frac": round(ops / dt / 1e12 / PEAK_INT8_TOPS, 4), "traffic": traffic_for(key="llama8b-linears")[0], "traffic_source": traffic_for(key="llama8b-linears")[1],
                                   "note": "whole pass incl. every activation quantisation (silu*mul fused into down's) and the strided read of the qkv slice"},
                      "cpu_baseline": _cpu(args, oracle, lambda R: CB.llama_layer(R, M, H, I, 6144, (H, H), 2 * I, H, I, norms=args.norms,
                                                                                   what=f"ONE decoder layer's linear path of Llama-3-8B at {M} tokens on the host"),
                                           scale=L, note=f"one layer x {L} (lm_head not included)")})


def run_llama8b(args, oracle=None):
    """BASELINE configs[3] as a MODEL: transformers' LlamaForCausalLM at Llama-3-8B dims (hidden 4096, intermediate 14336, 32
    layers, 32 heads / 8 KV heads, vocab 128256) with synthetic weights initialised on the GPU, every nn.Linear swapped to qlinear
    (swap_linears: the product path quantises the bf16 weights), gated MLPs as GatedMLP, both RMSNorms of every layer fused into
    the activation quantisation and q/k/v as one fused GEMM (protoquant_amd.llama.fuse_llama_layers).  Prefill, bs 1, seq 4096.
    Reported: end-to-end latency; the time of the int8 linear path (every quantisation + GEMM launch, measured with HIP events
    around those modules) and its TOPS over the 61.48 TOP of linear work; attention / rope / residual / embedding / final norm
    (stock torch-ROCm ops) as the remainder."""
    import transformers as tr
    import protoquant_amd as pq
    from protoquant_amd.llama import RMSNormQuant, fuse_llama_layers
    dev = torch.device("cuda", 0)
    S = args.tokens
    cfg = tr.LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=args.layers, num_attention_heads=32,
                         num_key_value_heads=8, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=500000.0, attn_implementation="sdpa")
    t0 = time.perf_counter()
    torch.manual_seed(1234)
    with torch.device(dev):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.bfloat16)
        try:
            model = tr.LlamaForCausalLM(cfg).eval()
        finally:
            torch.set_default_dtype(prev)
    pq.swap_linears(model, fuse_gated_mlp=True)
    nfused = 0 if args.no_layer_fusion else fuse_llama_layers(model)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    ids = torch.randint(0, cfg.vocab_size, (1, S), device=dev)

    # HIP events around every int8-path module (fused norm+quant, qlinear / FusedQLinear / GatedMLP); attention's projections
    # sit inside self_attn, so the hooks go on the leaves
    from protoquant_amd.llama import _FusedSlice, _SharedFused
    inside = set()                       # qlinears that a timed parent already covers
    for m in model.modules():
        if isinstance(m, (pq.GatedMLP, _SharedFused)):
            inside.update(id(sub) for sub in m.modules() if sub is not m)
    timed = [m for m in model.modules()
             if isinstance(m, (RMSNormQuant, pq.GatedMLP)) or (isinstance(m, _FusedSlice) and m.index == 0)
             or (isinstance(m, (pq.qlinear, pq.FusedQLinear)) and id(m) not in inside)]
    spans, timing = [], {"on": False}

    def pre(mod, inp):
        if timing["on"]:
            mod._span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            mod._span[0].record()

    def post(mod, inp, out):
        if timing["on"]:
            mod._span[1].record()
            spans.append(mod._span)
    for m in timed:
        m.register_forward_pre_hook(pre); m.register_forward_hook(post)

    def fwd():
        with torch.no_grad():
            return model(ids, use_cache=False, logits_to_keep=0).logits
    for _ in range(max(1, args.warmup if args.warmup < 5 else 2)):
        out = fwd()
    torch.cuda.synchronize()
    assert out.shape == (1, S, cfg.vocab_size)
    lat = []
    for _ in range(max(3, min(args.steps, 10))):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fwd(); torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    lat.sort()
    e2e = lat[len(lat) // 2]
    timing["on"] = True
    lin_t = []
    for _ in range(3):
        spans.clear()
        fwd(); torch.cuda.synchronize()
        lin_t.append(sum(a.elapsed_time(b) for a, b in spans) * 1e-3)
    timing["on"] = False
    lin_t.sort()
    t_lin = lin_t[len(lin_t) // 2]
    L, H, I, V = cfg.num_hidden_layers, 4096, 14336, cfg.vocab_size
    ops = L * (2.0 * S * 6144 * H + 2.0 * S * H * H + 2.0 * S * 2 * I * H + 2.0 * S * H * I) + 2.0 * S * V * H
    emit_json({"metric": "int8 TOPS of the linear path + end-to-end prefill latency, Llama-3-8B (every nn.Linear as qlinear), bs 1 seq 4096",
                      "value": round(ops / t_lin / 1e12, 2), "unit": "TOPS", "n_gpus": 1, "steps": len(lat), "warmup": 2, "ms_per_step": round(e2e * 1e3, 3),
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                      "config": {"workload": f"transformers LlamaForCausalLM at Llama-3-8B dims ({L} layers), synthetic weights, swap_linears(fuse_gated_mlp) + fuse_llama_layers, prefill bs 1 seq {S} (BASELINE configs[3])",
                                 "layers_fused": nfused, "attention": "stock torch-ROCm (sdpa), rope / residual / embedding / final norm stock",
                                 "int8_ops_per_step": ops, "build_seconds": round(t_build, 1)},
                      "end_to_end_ms": round(e2e * 1e3, 3), "end_to_end_ms_min": round(lat[0] * 1e3, 3),
                      "linear_path_ms": round(t_lin * 1e3, 3), "other_ops_ms": round((e2e - t_lin) * 1e3, 3),
                      "roofline": {"bound": "mfma", "achieved": round(ops / t_lin / 1e12, 1), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                                   "frac": round(ops / t_lin / 1e12 / PEAK_INT8_TOPS, 4), "traffic": traffic_for(key="llama8b")[0] if L == 32 and S == 4096 else None,
                                   "traffic_source": traffic_for(key="llama8b")[1] if L == 32 and S == 4096 else None,
                                   "note": "all int8-path launches of the forward: fused RMSNorm+quant (x2 per layer), fused qkv, o, fused gate+up, silu*mul+quant, down, lm_head (its own K1); HIP events around those modules; traffic: the int8-path kernels of one forward"},
                      "cpu_baseline": _cpu(args, oracle, lambda R: CB.llama_layer(R, S, H, I, 6144, (H, H), 2 * I, H, I, norms=True,
                                                                                   what=f"ONE decoder layer's int8 linear path of Llama-3-8B at {S} tokens on the host (HF RMSNorm -> quantize -> torch._int_mm -> epilogue; attention not run, as in the GPU's linear-path figure)"),
                                           scale=L, note=f"one layer x {L} (lm_head not included)")})


def run_llama70b_shard(args, oracle=None):
    """BASELINE configs[4], ONE rank's compute at the real shapes: Llama-3-70B (hidden 8192, intermediate 28672, 80 layers, 64 heads /
    8 KV heads, vocab 128256) with every linear's int8 weight column-sharded over G = 8 GPUs, M = 4096 tokens (seq is not stated in
    BASELINE.json: assumed 4096 as in configs[3]).  This process plays rank 0 on one GPU: per layer the fused q/k/v shard
    (N = 10240 / 8 = 1280), the o shard (1024), the fused gate+up shard (2 x 3584) and the down shard (1024, K = 28672), each with
    its activation quantisation on the replicated input (RMSNorm fused for q/k/v and gate/up; silu*mul needs the GATHERED gate/up in this
    configuration, so down's input is quantised by plain K1), plus the lm_head shard (16032).  No collective runs (one GPU): the
    exchange is priced from the bytes with the xGMI link model of DESIGN.md section 6, and stated as modelled."""
    import protoquant_amd as pq
    from protoquant_amd import _lib as L_
    from protoquant_amd.qtensor import QTensor
    dev = torch.device("cuda", 0)
    G, M, H, I, V, L = 8, args.tokens, 8192, 28672, 128256, args.layers if args.layers != 32 else 80
    KVD = 1024

    def mkq(n, k):
        wq = (torch.randn(n, k, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
        return pq.qlinear.from_qtensor(QTensor(wq, torch.rand(n, device=dev) * 1e-3 + 1e-4, 1, torch.bfloat16, wq.shape))
    n_qkv, n_o, n_gu, n_down, n_head = (H + 2 * KVD) // G, H // G, 2 * I // G, H // G, V // G
    # NL distinct layers' weights take turns (105 MB of shards per layer: two sets would sit in the 256-MB Infinity Cache; eight — 840 MB — make every layer stream its
    # weights from HBM, as the 80 distinct layers of the model do)
    NL = 8
    layers = [(mkq(n_qkv, H), mkq(n_o, H), mkq(n_gu, H), mkq(n_down, I)) for _ in range(NL)]
    head = mkq(n_head, H)
    x_h = torch.randn(M, H, device=dev).to(torch.bfloat16)          # stands for a gathered hidden state
    x_i = torch.randn(M, I, device=dev).to(torch.bfloat16)          # stands for the gathered silu(g) * u
    norm_w = torch.ones(H, device=dev, dtype=torch.bfloat16)

    def fwd():
        for l in range(L):
            qkv, o, gu, down = layers[l % NL]
            qkv(pq.rmsnorm_quantize(x_h, norm_w, 1e-5))
            o(x_h)
            gu(pq.rmsnorm_quantize(x_h, norm_w, 1e-5))
            down(x_i)
        return head(x_h)
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    ts = []
    for _ in range(max(3, min(args.steps, 10))):
        t0 = time.perf_counter(); fwd(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    dt = ts[len(ts) // 2]
    ops = L * 2.0 * M * (n_qkv * H + n_o * H + n_gu * H + n_down * I) + 2.0 * M * n_head * H
    # Extra key — the ROW-SHARDED PAIRING of SURVEY section 8(f)4 (RowShardedQLinear / ShardedGatedMLP): o and down take the LOCAL shard of
    # their producer's output (this rank's heads / intermediate channels) against the matching K-slice of the weight and emit f32 partials
    # [M, H] for a reduce-scatter; no gather sits between gate/up and down, and silu*mul is fused into the local quantisation.  Same int8 ops
    # per rank, GEMM shapes 4096 x 8192 x 1024 and 4096 x 8192 x 3584 instead of the 1024-wide column shards.
    pairing = None
    try:
        if args.no_extras:
            raise RuntimeError("--no-extras")
        rl = [(mkq(H, H // G), mkq(H, I // G)) for _ in range(NL)]
        x_a = torch.randn(M, H // G, device=dev).to(torch.bfloat16)          # stands for this rank's heads of the attention output

        def fwd_row():
            for l in range(L):
                qkv, _o, gu, _d = layers[l % NL]
                o_r, d_r = rl[l % NL]
                qkv(pq.rmsnorm_quantize(x_h, norm_w, 1e-5))
                xa = pq.quantize(x_a)
                pq.qlinear_s8(xa.int_data, xa.scale, o_r.wq, o_r.ws, None, torch.float32)
                g_, u_ = gu(pq.rmsnorm_quantize(x_h, norm_w, 1e-5)).split(I // G, dim=-1)
                hq = pq.silu_mul_quantize(g_, u_)
                pq.qlinear_s8(hq.int_data, hq.scale, d_r.wq, d_r.ws, None, torch.float32)
            return head(x_h)
        fwd_row(); torch.cuda.synchronize()
        tr = []
        for _ in range(3):
            t0 = time.perf_counter(); fwd_row(); torch.cuda.synchronize()
            tr.append(time.perf_counter() - t0)
        dr = sorted(tr)[1]
        # per layer two exchanges: reduce-scatter of the f32 partials [M, H] + all-gather of the bf16 row blocks (the next column-sharded
        # linear wants the activation replicated): a rank moves 7/8 of M*H*4 + 7/8 of M*H*2 bytes each time
        moved = L * 2 * (G - 1) / G * (M * H * 4 + M * H * 2) + 2.0 * M * V * (G - 1) / G
        pairing = {"ms_per_step": round(dr * 1e3, 3), "value": round(ops / dr / 1e12, 2), "unit": "TOPS",
                   "gemm_shapes": f"4096x{n_qkv}x8192 (fused qkv, column), 4096x8192x{H // G} (o, row), 4096x{n_gu}x8192 (fused gate+up, column), 4096x8192x{I // G} (down, row)",
                   "modelled_exchange_ms": round(moved / (7 * 153e9) * 1e3, 2),
                   "model": "per layer 2 x (reduce-scatter of f32 partials [M,H] + all-gather of bf16 row blocks), 7 xGMI links x 153 GB/s; NOT measured"}
        del rl
    except Exception as e:      # an extra figure must never lose the main line
        if not args.no_extras:
            print(f"[bench] row-sharded pairing leg failed: {e}", file=sys.stderr)
    # Extra key — the INT8-CODE EXCHANGE between gate/up and down (ColumnShardedGatedMLP: everything stays column-sharded, as north_star asks): the rank computes
    # silu*mul on ITS 3584 intermediate channels only — row amax of the local block, [all-reduce(max) of 4096 32-bit patterns: not run on one GPU], encode against the
    # global amax — and the down shard's GEMM walks the all-gathered int8 blocks [8, 4096, 3584] in place.  Against the plain composition above, the re-quantisation of
    # the gathered 4096 x 28672 activation on every rank (and the gather of bf16 gate AND up: 4 bytes per intermediate element instead of 1) is gone.
    int8x = None
    try:
        if args.no_extras:
            raise RuntimeError("--no-extras")
        from protoquant_amd.qtensor import quantize_with_amax, rowamax, silu_mul_quantize_with_amax, silu_mul_rowamax
        ig = I // G
        stacked = torch.randint(-127, 128, (G, M, ig), device=dev, dtype=torch.int8)          # stands for the all-gathered code blocks
        stacked_scale = torch.rand(M, device=dev) * 1e-2 + 1e-4
        # ... and the same exchange in front of `o` (ColumnShardedQLinear.forward_sharded_input): the rank's heads of the attention output [M, H / G] are quantised
        # locally against the all-reduced amax and the int8 blocks gathered — instead of K1 on the gathered bf16 [M, H] on every rank
        x_att = torch.randn(M, H // G, device=dev).to(torch.bfloat16)
        stacked_o = torch.randint(-127, 128, (G, M, H // G), device=dev, dtype=torch.int8)

        def fwd_int8():
            for l in range(L):
                qkv, o, gu, down = layers[l % NL]
                qkv(pq.rmsnorm_quantize(x_h, norm_w, 1e-5))
                quantize_with_amax(x_att, rowamax(x_att), out=stacked_o[0])
                pq.qlinear_s8_kslabs(stacked_o, stacked_scale, o.wq, o.ws, None, torch.bfloat16)
                g_, u_ = gu(pq.rmsnorm_quantize(x_h, norm_w, 1e-5)).split(ig, dim=-1)
                am = silu_mul_rowamax(g_, u_)
                silu_mul_quantize_with_amax(g_, u_, am, out=stacked[0])
                pq.qlinear_s8_kslabs(stacked, stacked_scale, down.wq, down.ws, None, torch.bfloat16)
            return head(x_h)
        fwd_int8(); torch.cuda.synchronize()
        ti = []
        for _ in range(3):
            t0 = time.perf_counter(); fwd_int8(); torch.cuda.synchronize()
            ti.append(time.perf_counter() - t0)
        di = sorted(ti)[1]
        XG = 153e9
        hop_bf16 = 2 * (2.0 * M * ig) / XG + 2 * 2 * (2.0 * M * I * 2) / 5e12       # gather of the bf16 gate and up shards (one link per peer, all links at once) + their layout passes
        hop_int8 = (4.0 * M) / XG + (1.0 * M * ig) / XG                             # all-reduce of M amax patterns (latency-bound in practice) + gather of the int8 blocks; no layout pass
        # a layer's exchanges, modelled per rank (its shard over one direct link per peer at 153 GB/s, all 7 links at once; layout passes at 5 TB/s): the plain composition
        # gathers every projection's bf16 output (q/k/v, o, gate and up, down); the int8 form gathers nothing after q/k/v (the rank's own heads consume them), int8 codes
        # in front of o and down, and the bf16 outputs of o and down (the replicated residual stream)
        link = lambda nbytes: nbytes / XG * 1e6      # noqa: E731
        ex_bf16 = link(2.0 * M * (H + 2 * KVD) / G) + link(2.0 * M * H / G) + link(2.0 * M * 2 * I / G) + link(2.0 * M * H / G) \
            + 2 * 2.0 * M * ((H + 2 * KVD) + H + 2 * I + H) / 5e12 * 1e6
        ex_int8 = link(1.0 * M * H / G) + link(2.0 * M * H / G) + link(1.0 * M * I / G) + link(2.0 * M * H / G) + 2 * link(4.0 * M) + 2 * 2 * 2.0 * M * H / 5e12 * 1e6
        int8x = {"ms_per_step": round(di * 1e3, 3), "value": round(ops / di / 1e12, 2), "unit": "TOPS", "us_per_layer": round((di - 0) / L * 1e6, 1),
                 "modelled_exchange_us_per_layer": {"bf16_gather_of_every_output": round(ex_bf16, 1), "int8_code_exchange": round(ex_int8, 1),
                                                    "model": "per rank: its shard over one direct xGMI link per peer at 153 GB/s, all 7 links at once, + layout passes at 5 TB/s; NOT measured, not overlapped with compute"},
                 "what": "per layer: rmsnorm x2, qkv shard, row amax + encode of the LOCAL 1024 attention features, o shard on the stacked int8 blocks, gate+up shard, "
                         "silu*mul row amax + encode on the LOCAL 3584 channels, down shard on the stacked int8 blocks (slabs walked in place)",
                 "modelled_gate_up_to_down_hop_us": {"bf16_gather_of_gate_and_up_plus_layout": round(hop_bf16 * 1e6, 1), "int8_code_exchange": round(hop_int8 * 1e6, 1),
                                                     "model": "bytes over one direct xGMI link per peer at 153 GB/s, all 7 links at once; layout passes at 5 TB/s; NOT measured"}}
    except Exception as e:      # an extra figure must never lose the main line
        if not args.no_extras:
            print(f"[bench] int8-code exchange leg failed: {e}", file=sys.stderr)
    # ---- where the step goes: every distinct kernel of a layer by itself, gap-free from its own hipGraph (all NL weight sets in turn: HBM-fed like the step),
    # with its share of the layer and — for the GEMMs —
    # its fraction of the int8 peak: the per-shape account of the distance to 0.50
    per_shape = None
    try:
        if args.no_extras:
            raise RuntimeError("--no-extras")
        def ev_graph(fn, n=8):
            g = graph_of(fn, n)
            g.replay(); torch.cuda.synchronize()
            v = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); b.synchronize()
                v.append(a.elapsed_time(b) * 1e3 / n)
            return sorted(v)[len(v) // 2]
        qn = pq.rmsnorm_quantize(x_h, norm_w, 1e-5)
        qh = pq.quantize(x_h)
        qi = pq.quantize(x_i)
        items = [("rmsnorm -> int8 (input of qkv; again for gate+up)", None, lambda l: pq.rmsnorm_quantize(x_h, norm_w, 1e-5), 2),
                 ("fused qkv shard", (M, n_qkv, H), lambda l: layers[l][0](qn), 1),
                 ("K1 of o's input (replicated attention output)", None, lambda l: pq.quantize(x_h), 1),
                 ("o shard", (M, n_o, H), lambda l: layers[l][1](qh), 1),
                 ("fused gate+up shard", (M, n_gu, H), lambda l: layers[l][2](qn), 1),
                 ("K1 of down's input (the GATHERED silu(g)*u, 4096 x 28672)", None, lambda l: pq.quantize(x_i), 1),
                 ("down shard", (M, n_down, I), lambda l: layers[l][3](qi), 1)]
        # the int8-code exchange's three kernels (reported beside the account, not summed into it)
        extra_items = []
        if int8x is not None:
            gu_out = layers[0][2](qn)
            g0, u0 = gu_out.split(I // G, dim=-1)
            am0 = silu_mul_rowamax(g0, u0)
            extra_items = [("row amax + encode of the local attention features 4096 x 1024 (two launches)", None, lambda l: quantize_with_amax(x_att, rowamax(x_att), out=stacked_o[0]), 1),
                           ("o shard on stacked int8 blocks [8, 4096, 1024]", (M, n_o, H), lambda l: pq.qlinear_s8_kslabs(stacked_o, stacked_scale, layers[l][1].wq, layers[l][1].ws, None, torch.bfloat16), 1),
                           ("silu*mul row amax, local 4096 x 3584 (int8-code exchange)", None, lambda l: silu_mul_rowamax(g0, u0), 1),
                           ("silu*mul encode against the global amax, local 4096 x 3584", None, lambda l: silu_mul_quantize_with_amax(g0, u0, am0, out=stacked[0]), 1),
                           ("down shard on stacked int8 blocks [8, 4096, 3584]", (M, n_down, I), lambda l: pq.qlinear_s8_kslabs(stacked, stacked_scale, layers[l][3].wq, layers[l][3].ws, None, torch.bfloat16), 1)]
        per_shape, tot = [], 0.0
        for name, shp, fn, mult in items:
            us = ev_graph(lambda: [fn(l) for l in range(NL)], 2) / NL
            tot += us * mult
            d = {"kernel": name, "us": round(us, 1), "per_layer": mult}
            if shp is not None:
                d["shape"] = "x".join(str(v) for v in shp)
                d["frac_of_int8_peak"] = round(2.0 * shp[0] * shp[1] * shp[2] / us / 1e6 / PEAK_INT8_TOPS, 3)
                d["dispatch"] = L_.lib().pq_gemm_variant_name(shp[0], shp[1], shp[2], shp[2], shp[2]).decode() + (" + workspace" if L_.lib().pq_qlinear_workspace_bytes(*shp) else "")
            per_shape.append(d)
        for d in per_shape:
            d["share_of_layer"] = round(d["us"] * d["per_layer"] / tot, 3)
        per_shape.append({"sum_per_layer_us": round(tot, 1), "step_per_layer_us": round(dt / L * 1e6, 1)})
        # one layer of each composition replayed gap-free from a hipGraph over the NL weight sets (host-independent; the eager steps above carry Python's launch overhead)
        def layer_plain(l):
            qkv, o, gu, down = layers[l]
            qkv(pq.rmsnorm_quantize(x_h, norm_w, 1e-5)); o(x_h); gu(pq.rmsnorm_quantize(x_h, norm_w, 1e-5)); down(x_i)
        lg = {"column_sharded_bf16_gather": round(ev_graph(lambda: [layer_plain(l) for l in range(NL)], 2) / NL, 1)}
        if int8x is not None:
            def layer_int8(l):
                qkv, o, gu, down = layers[l]
                qkv(pq.rmsnorm_quantize(x_h, norm_w, 1e-5))
                quantize_with_amax(x_att, rowamax(x_att), out=stacked_o[0])
                pq.qlinear_s8_kslabs(stacked_o, stacked_scale, o.wq, o.ws, None, torch.bfloat16)
                g_, u_ = gu(pq.rmsnorm_quantize(x_h, norm_w, 1e-5)).split(I // G, dim=-1)
                silu_mul_quantize_with_amax(g_, u_, silu_mul_rowamax(g_, u_), out=stacked[0])
                pq.qlinear_s8_kslabs(stacked, stacked_scale, down.wq, down.ws, None, torch.bfloat16)
            lg["int8_code_exchange"] = round(ev_graph(lambda: [layer_int8(l) for l in range(NL)], 2) / NL, 1)
            int8x["us_per_layer_graph"] = lg["int8_code_exchange"]
        per_shape.append({"layer_us_from_hipgraph": lg})
        if extra_items:
            ex = []
            for name, shp, fn, mult in extra_items:
                us = ev_graph(lambda: [fn(l) for l in range(NL)], 2) / NL
                d = {"kernel": name, "us": round(us, 1)}
                if shp is not None:
                    d["frac_of_int8_peak"] = round(2.0 * shp[0] * shp[1] * shp[2] / us / 1e6 / PEAK_INT8_TOPS, 3)
                ex.append(d)
            int8x["kernels"] = ex
    except Exception as e:
        if not args.no_extras:
            print(f"[bench] per-shape leg failed: {e}", file=sys.stderr)
    # exchange model: every linear's bf16 output is all-gathered after dequant; a rank receives (G-1)/G of it over 7 xGMI links x ~153 GB/s
    gathered = L * 2.0 * M * (H + 2 * KVD + H + 2 * I + H) + 2.0 * M * V
    t_gather = gathered * (G - 1) / G / (7 * 153e9)
    emit_json({"metric": "int8 TOPS per GPU, Llama-3-70B column-sharded over 8 GPUs: one rank's linears at M=4096 (exchange modelled)",
                      "value": round(ops / dt / 1e12, 2), "unit": "TOPS", "n_gpus": 1, "steps": len(ts), "warmup": 2, "ms_per_step": round(dt * 1e3, 3),
                      "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "s8", "data": "synthetic",
                      "config": {"workload": f"one of 8 ranks of Llama-3-70B ({L} layers + lm_head), weights column-sharded: per-GPU shards 4096x{n_qkv}x8192 (fused qkv), "
                                             f"4096x{n_o}x8192 (o), 4096x{n_gu}x8192 (fused gate+up), 4096x{n_down}x28672 (down), 4096x{n_head}x8192 (lm_head) (BASELINE configs[4])",
                                 "int8_ops_per_rank": ops, "gathered_bytes_per_pass": gathered,
                                 "modelled_allgather_ms": round(t_gather * 1e3, 2), "row_sharded_pairing": pairing, "int8_code_exchange": int8x, "per_shape": per_shape,
                                 "model": "all-gather after dequant of every linear's bf16 output; a rank receives 7/8 of it over 7 xGMI links x 153 GB/s (fully connected, direct); NOT measured"},
                      "roofline": {"bound": "mfma", "achieved": round(ops / dt / 1e12, 1), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                                   "frac": round(ops / dt / 1e12 / PEAK_INT8_TOPS, 4), "traffic": traffic_for(key="llama70b-shard")[0] if L == 80 and M == 4096 else None,
                                   "traffic_source": traffic_for(key="llama70b-shard")[1] if L == 80 and M == 4096 else None,
                                   "note": "one rank's compute only: every activation quantisation (RMSNorm fused for q/k/v and gate/up) + the shard GEMMs; traffic: per step (all layers + lm_head shard)"},
                      "cpu_baseline": _cpu(args, oracle, lambda R: CB.llama_layer(R, M, H, I, n_qkv, (n_o, H), n_gu, n_down, I, norms=True,
                                                                                   what=f"ONE layer of one rank's shards at {M} tokens on the host (the bf16-gather composition: down's input is the gathered [M, {I}] intermediate)"),
                                           scale=L, note=f"one layer x {L} (lm_head shard not included)")})

