"""qlinear: the nn.Linear replacement of the contract (BASELINE.json north_star) + swap_linears().

forward(x[..., K]) = K1 row-quant(x) -> K3 s8xs8->s32 MFMA GEMM -> K4 fused (row-scale x col-scale
(+bias)) epilogue, all inside libpq_hip.so.  Weights are quantised once, per output channel, when
the module is built.  Reference definitions are absent from the mount (/root/reference holds only
CODE_OF_CONDUCT.md:1-80); semantics follow QSPEC v2."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L
from .qtensor import QTensor, quantize, silu_mul_quantize


def int_mm(xq: torch.Tensor, wq: torch.Tensor) -> torch.Tensor:
    """acc[M,N] = xq[M,K] . wq[N,K]^T, exact int32 — the GPU twin of torch._int_mm(xq, wq.t())."""
    L.require_gpu(xq, "int_mm(xq)")
    xq, wq = L.row_major_2d(xq), L.row_major_2d(wq)
    M, K = xq.shape
    N, K2 = wq.shape
    if K != K2 or xq.dtype != torch.int8 or wq.dtype != torch.int8:
        raise ValueError("int_mm expects int8 [M,K] and int8 [N,K]")
    if _pad_pays(M, N, K):
        K = _round_k(K)
        xq, wq = _pad_k(xq, K), _pad_k(wq, K)
    acc = torch.empty((M, N), dtype=torch.int32, device=xq.device)
    with torch.cuda.device(xq.device):
        L.check(L.lib().pq_gemm_s8s8s32(xq.data_ptr(), L.ld(xq), wq.data_ptr(), L.ld(wq), acc.data_ptr(), max(N, 1),
                                        M, N, K, L.stream_ptr(xq)), "int_mm")
    return acc


_WORKSPACES: dict = {}     # (device, stream) -> [buffer, handed out under a hipGraph capture?]
_RETIRED: list = []        # outgrown buffers whose address a captured hipGraph holds


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Caller-owned scratch (split-K slabs, the one-call path's codes and scales): one buffer per (device, stream) — reuse is
    ordered by the stream, and two streams never share a buffer.  It grows geometrically (at least 2x), so a run with rising M
    reallocates O(log) times.  An outgrown buffer is simply dropped — the caching allocator hands its memory out again in stream
    order, which is safe because every use of it was enqueued on this stream before the free — UNLESS it was ever handed out while
    a hipGraph was being captured (whether it was allocated during that capture or eagerly before it, the usual warm-up): a captured
    launch keeps the raw address, so such a buffer is parked until clear_workspaces()."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    capturing = torch.cuda.is_current_stream_capturing()
    ent = _WORKSPACES.get(key)
    if ent is None or ent[0].numel() < nbytes:
        if ent is not None and ent[1]:
            _RETIRED.append(ent[0])
        size = nbytes if ent is None else max(nbytes, 2 * ent[0].numel())
        ent = [torch.empty((size,), dtype=torch.uint8, device=device), False]
        _WORKSPACES[key] = ent
    if capturing:
        ent[1] = True          # graph-pinned from now on: replays write through this address
    return ent[0]


def clear_workspaces() -> None:
    """Drop every cached workspace (call when no captured hipGraph that used this module will be replayed again)."""
    _WORKSPACES.clear()
    _RETIRED.clear()


def _check_operand(t, name: str, dev, dtype, numel=None):
    """The C-ABI takes raw pointers: a CPU tensor, a wrong dtype or a strided vector would be read as garbage (or fault)."""
    if t.device != dev:
        raise L.PQError(f"{name} is on {t.device}, expected {dev}: protoquant_amd has no CPU fallback — move the module / tensor to the GPU")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if numel is not None and (t.dim() != 1 or t.numel() != numel or (numel > 1 and t.stride(0) != 1)):
        raise ValueError(f"{name} must be a contiguous vector of {numel} elements, got shape {tuple(t.shape)} stride {t.stride()}")


def _round_k(K: int) -> int:
    return -(-K // 128) * 128


def _pad_k(t: torch.Tensor, kp: int) -> torch.Tensor:
    """[rows, K] int8 -> a fresh [rows, kp] copy with a zeroed tail."""
    K = t.shape[1]
    buf = torch.empty((t.shape[0], kp), dtype=torch.int8, device=t.device)
    buf[:, :K].copy_(t)
    buf[:, K:].zero_()
    return buf


def _pad_pays(M: int, N: int, K: int) -> bool:
    """The functional entry points with a K that is not a multiple of 128: the C-ABI would run its generic kernel (~300 TOPS, and ~50 when the rows are not 16-byte
    aligned) where the MFMA tiles do 2000+.  Copying BOTH operands into zero-tailed buffers of the next multiple of 128 costs two int8 copies and four small launches
    (~10 us + (M + N) K bytes); it pays from about 2^31 multiply-adds on (4096 x 4096 x 4000: 429 -> ~80 us).  Same bits: zeros add nothing to an integer sum.
    (The modules keep a padded weight copy instead of making one per call: _KPadded.)"""
    return K % 128 != 0 and K > 0 and M * N * K >= (1 << 31)


def qlinear_s8(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias, out_dtype,
               out: torch.Tensor | None = None) -> torch.Tensor:
    """Fused int8 GEMM + dequant epilogue on pre-quantised operands (C-ABI pq_qlinear_s8)."""
    L.require_gpu(xq, "qlinear_s8(xq)")
    xq, wq = L.row_major_2d(xq), L.row_major_2d(wq)
    M, K = xq.shape
    N = wq.shape[0]
    if wq.shape[1] != K:
        raise ValueError(f"shape mismatch: x has K={K}, weight has K={wq.shape[1]}")
    code = L.dtype_code(out_dtype)
    if bias is not None and bias.dtype != out_dtype:
        bias = bias.to(out_dtype)
    dev = xq.device
    _check_operand(xq, "xq", dev, torch.int8); _check_operand(wq, "wq", dev, torch.int8)
    _check_operand(xs, "xs", dev, torch.float32, M); _check_operand(ws, "ws", dev, torch.float32, N)
    if _pad_pays(M, N, K):
        K = _round_k(K)
        xq, wq = _pad_k(xq, K), _pad_k(wq, K)
    if bias is not None:
        _check_operand(bias, "bias", dev, out_dtype, N)
    if out is not None:
        _check_operand(out, "out", dev, out_dtype)
        if out.dim() != 2 or out.shape != (M, N) or (N > 1 and out.stride(1) != 1):
            raise ValueError(f"out must be a row-major [{M}, {N}] tensor, got {tuple(out.shape)} stride {out.stride()}")
    y = out if out is not None else torch.empty((M, N), dtype=out_dtype, device=xq.device)
    with torch.cuda.device(xq.device):          # (the planner's thresholds follow the CURRENT device's CU count: query and launch under the tensor's device)
        wbytes = L.lib().pq_qlinear_workspace_bytes(M, N, K)        # > 0: split-K pays for this shape
        wsp = _workspace(xq.device, wbytes) if wbytes else None
        L.check(L.lib().pq_qlinear_s8(xq.data_ptr(), L.ld(xq), xs.data_ptr(), wq.data_ptr(), L.ld(wq), ws.data_ptr(),
                                      bias.data_ptr() if bias is not None else None, y.data_ptr(), L.ld(y), code,
                                      M, N, K, wsp.data_ptr() if wsp is not None else None, wbytes,
                                      L.stream_ptr(xq)), "qlinear_s8")
    return y


def qlinear_s8_t(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias, out_dtype,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """The same fused GEMM + epilogue with the output TRANSPOSED: returns yt[N, M], yt[n][m] bit-identical to
    qlinear_s8(...)[m][n] (C-ABI pq_qlinear_s8_t).  The column-sharded configuration gathers these row blocks contiguously."""
    L.require_gpu(xq, "qlinear_s8_t(xq)")
    xq, wq = L.row_major_2d(xq), L.row_major_2d(wq)
    M, K = xq.shape
    N = wq.shape[0]
    if wq.shape[1] != K:
        raise ValueError(f"shape mismatch: x has K={K}, weight has K={wq.shape[1]}")
    code = L.dtype_code(out_dtype)
    if bias is not None and bias.dtype != out_dtype:
        bias = bias.to(out_dtype)
    dev = xq.device
    _check_operand(xq, "xq", dev, torch.int8); _check_operand(wq, "wq", dev, torch.int8)
    _check_operand(xs, "xs", dev, torch.float32, M); _check_operand(ws, "ws", dev, torch.float32, N)
    if _pad_pays(M, N, K):
        K = _round_k(K)
        xq, wq = _pad_k(xq, K), _pad_k(wq, K)
    if bias is not None:
        _check_operand(bias, "bias", dev, out_dtype, N)
    if out is not None and (out.device != dev or out.dtype != out_dtype or out.shape != (N, M) or (M > 1 and out.stride(1) != 1)):
        raise ValueError(f"out must be a row-major [{N}, {M}] {out_dtype} tensor on {dev}")
    yt = out if out is not None else torch.empty((N, M), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        wbytes = L.lib().pq_qlinear_t_workspace_bytes(M, N, K)
        wsp = _workspace(dev, wbytes) if wbytes else None
        L.check(L.lib().pq_qlinear_s8_t(xq.data_ptr(), L.ld(xq), xs.data_ptr(), wq.data_ptr(), L.ld(wq), ws.data_ptr(),
                                        bias.data_ptr() if bias is not None else None, yt.data_ptr(), L.ld(yt), code,
                                        M, N, K, wsp.data_ptr() if wsp is not None else None, wbytes, L.stream_ptr(xq)), "qlinear_s8_t")
    return yt


def qlinear_s8_kslabs(xq_stacked: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias, out_dtype,
                      out: torch.Tensor | None = None) -> torch.Tensor:
    """qlinear_s8 on STACKED activation codes xq_stacked[G, M, K/G] (contiguous: what an all-gather of the ranks' int8 column blocks leaves): slab s holds the
    columns [s*K/G, (s+1)*K/G) of the logical xq[M, K].  Same bits as qlinear_s8 on the row-major matrix (an integer sum has no order); the 128 x 128 ring tile
    walks the slabs in place, and so does the fused split-K of the 256 x 256 tile where the planner runs it (the Llama-70B `down` shard); every other shape takes one
    layout pass inside the call (C-ABI pq_qlinear_s8_kslabs)."""
    L.require_gpu(xq_stacked, "qlinear_s8_kslabs(xq)")
    if xq_stacked.dim() != 3 or not xq_stacked.is_contiguous():
        raise ValueError("qlinear_s8_kslabs: activation codes must be a contiguous [G, M, K/G] int8 tensor")
    G, M, kps = xq_stacked.shape
    K = G * kps
    wq = L.row_major_2d(wq)
    N = wq.shape[0]
    if wq.shape[1] != K:
        raise ValueError(f"shape mismatch: stacked x has K={K}, weight has K={wq.shape[1]}")
    code = L.dtype_code(out_dtype)
    if bias is not None and bias.dtype != out_dtype:
        bias = bias.to(out_dtype)
    dev = xq_stacked.device
    _check_operand(xq_stacked, "xq", dev, torch.int8); _check_operand(wq, "wq", dev, torch.int8)
    _check_operand(xs, "xs", dev, torch.float32, M); _check_operand(ws, "ws", dev, torch.float32, N)
    if bias is not None:
        _check_operand(bias, "bias", dev, out_dtype, N)
    if out is not None and (out.device != dev or out.dtype != out_dtype or out.shape != (M, N) or (N > 1 and out.stride(1) != 1)):
        raise ValueError(f"out must be a row-major [{M}, {N}] {out_dtype} tensor on {dev}")
    y = out if out is not None else torch.empty((M, N), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        # the exact query (it looks at the operands' alignment and strides): the fused split-K's hand-over slabs where the 256 x 256 tile walks the blocks in place,
        # nothing where a ring tile does, the layout pass's M * K bytes otherwise
        wbytes = L.lib().pq_qlinear_kslabs_workspace_bytes_for(xq_stacked.data_ptr(), kps, M * kps, kps, wq.data_ptr(), L.ld(wq), M, N, K)
        wsp = _workspace(dev, wbytes) if wbytes else None
        L.check(L.lib().pq_qlinear_s8_kslabs(xq_stacked.data_ptr(), kps, M * kps, kps, xs.data_ptr(), wq.data_ptr(), L.ld(wq), ws.data_ptr(),
                                             bias.data_ptr() if bias is not None else None, y.data_ptr(), L.ld(y), code, M, N, K,
                                             wsp.data_ptr() if wsp is not None else None, wbytes, L.stream_ptr(xq_stacked)), "qlinear_s8_kslabs")
    return y


def qlinear_dyn(x: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias=None) -> torch.Tensor:
    """x[..., K] -> y[..., N]: dynamic per-token quant + int8 GEMM + fused dequant in ONE C-ABI call (pq_qlinear_dyn);
    the scratch (xq, xs, split-K slabs) lives in the per-device workspace."""
    L.require_gpu(x, "qlinear(x)")
    code = L.dtype_code(x.dtype)
    K = x.shape[-1]
    lead = 1
    for d in x.shape[:-1]:
        lead *= d
    x2 = L.row_major_2d(x.reshape(lead, K))
    wq = L.row_major_2d(wq)
    N = wq.shape[0]
    if wq.shape[1] != K:
        raise ValueError(f"shape mismatch: x has K={K}, weight has K={wq.shape[1]}")
    if bias is not None and bias.dtype != x.dtype:
        bias = bias.to(x.dtype)
    _check_operand(wq, "wq", x.device, torch.int8); _check_operand(ws, "ws", x.device, torch.float32, N)
    if bias is not None:
        _check_operand(bias, "bias", x.device, x.dtype, N)
    if _pad_pays(lead, N, K):
        xq = quantize(x2, axis=-1)
        return qlinear_s8(xq.int_data, xq.scale, wq, ws, bias, x.dtype).reshape(*x.shape[:-1], N)
    y = torch.empty((lead, N), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        wbytes = L.lib().pq_qlinear_dyn_workspace_bytes(lead, N, K)
        wsp = _workspace(x.device, max(wbytes, 256))
        L.check(L.lib().pq_qlinear_dyn(x2.data_ptr(), code, L.ld(x2), wq.data_ptr(), L.ld(wq), ws.data_ptr(),
                                       bias.data_ptr() if bias is not None else None, y.data_ptr(), max(N, 1),
                                       lead, N, K, wsp.data_ptr(), wsp.numel(), L.stream_ptr(x)), "qlinear_dyn")
    return y.reshape(*x.shape[:-1], N)


class _KPadded:
    """in_features that is not a multiple of 128 (GPT-2 XL's 1600, a 4000-wide projection): the MFMA tiles step through K in 128-byte K-tiles, and the C-ABI sends any
    other K to the generic kernel — 4 to 25x slower than the bf16 nn.Linear it replaces (measured: 4096 x 4096 x 4000 429 us against 99).  The modules therefore keep a
    zero-padded copy of their int8 weight, [N, round_up(K, 128)], made lazily and remade whenever `wq` changes (load_state_dict, .to()), and quantise the activation
    into a buffer of the same width whose tail is zeroed: zeros add nothing to an integer sum, so every accumulator — and every output bit — is the unpadded problem's.
    `wq` itself stays [N, K]: state dicts, fusion and sharding see the unpadded weight."""

    def _wq_for_gemm(self):
        K = self.in_features
        kp = _round_k(K)
        if kp == K or K == 0:
            return self.wq, K
        # keyed on the tensor OBJECT (held by a weak reference) and its version counter, not on its address: a weight replaced by a fresh tensor (module.to('cpu') ->
        # load_state_dict -> .to('cuda'), load_state_dict(assign=True) after a del) starts again at version 0 and can be handed the very address the caching allocator
        # just freed — an address-based key would then match and the GEMM would run on the OLD padded weight
        import weakref
        wq = self.wq
        # ... and ALSO on the address: `module.wq.data = other` / `wq.set_(...)` keep the Python object and do not bump its version
        key = (wq._version, wq.device, tuple(wq.shape), wq.data_ptr(), wq.storage_offset())
        c = self.__dict__.get("_wq_pad")
        if c is None or c[0] != key or c[2]() is not wq:
            w = wq.new_zeros((wq.shape[0], kp))
            w[:, :K].copy_(wq)
            c = (key, w, weakref.ref(wq))
            self.__dict__["_wq_pad"] = c
        return c[1], kp

    def __getstate__(self):
        """pickling / torch.save of a module: the lazily built padded copy (and the weak reference that keys it) is a cache, not state"""
        state = super().__getstate__()
        if "_wq_pad" in state:
            state = dict(state)
            del state["_wq_pad"]
        return state

    def _padded_forward(self, x, wq_pad, kp):
        """x: float [..., K] or a per-token QTensor; the GEMM runs over kp = round_up(K, 128) with zero tails on both operands."""
        K, N = self.in_features, self.out_features
        if isinstance(x, QTensor):
            src = L.row_major_2d(x.int_data.reshape(-1, K))
            M = src.shape[0]
            buf = torch.empty((M, kp), dtype=torch.int8, device=src.device)
            buf[:, :K].copy_(src)
            buf[:, K:].zero_()
            xs, out_dtype = x.scale, x.orig_dtype
        else:
            L.require_gpu(x, "qlinear(x)")
            code = L.dtype_code(x.dtype)
            lead = 1
            for d in x.shape[:-1]:
                lead *= d
            x2 = L.row_major_2d(x.reshape(lead, K))
            M = lead
            buf = torch.empty((M, kp), dtype=torch.int8, device=x.device)
            buf[:, K:].zero_()
            xs = torch.empty((M,), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                L.check(L.lib().pq_quant_rowwise(x2.data_ptr(), code, M, K, L.ld(x2), buf.data_ptr(), kp, xs.data_ptr(), L.stream_ptr(x)), "quantize")
            out_dtype = x.dtype
        y = qlinear_s8(buf, xs, wq_pad, self.ws, self.bias, out_dtype)
        return y.reshape(*x.shape[:-1], N)


def gemm_operands(local: "_KPadded", codes: torch.Tensor):
    """(codes, wq) for one GEMM of a module that owns `wq`: as they are when in_features is a multiple of 128, else both zero-padded to the next one (_KPadded).
    codes: int8 [M, in_features]."""
    K = local.in_features
    if K % 128 == 0 or not codes.is_cuda:      # (CPU stand-ins of the gloo host-logic tests pass through)
        return codes, local.wq
    wq, kp = local._wq_for_gemm()
    buf = torch.empty((codes.shape[0], kp), dtype=torch.int8, device=codes.device)
    buf[:, :K].copy_(codes)
    buf[:, K:].zero_()
    return buf, wq


class qlinear(_KPadded, nn.Module):
    """Drop-in for nn.Linear with dynamic per-token int8 activations and per-channel int8 weights."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        dtype = dtype or torch.bfloat16
        self.register_buffer("wq", torch.zeros((out_features, in_features), dtype=torch.int8, device=device))
        self.register_buffer("ws", torch.ones((out_features,), dtype=torch.float32, device=device))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features,), dtype=dtype, device=device))
        else:
            self.bias = None

    @classmethod
    def from_linear(cls, lin: nn.Linear) -> "qlinear":
        """Quantise lin.weight [N,K] per output channel on the GPU (kernel K1 over W's rows)."""
        w = lin.weight.detach()
        L.require_gpu(w, "qlinear.from_linear(weight)")
        m = cls.__new__(cls)
        nn.Module.__init__(m)
        m.in_features, m.out_features = lin.in_features, lin.out_features
        qw = quantize(w, axis=-1)
        m.register_buffer("wq", qw.int_data)
        m.register_buffer("ws", qw.scale)
        if lin.bias is not None:
            m.register_buffer("bias", lin.bias.detach().clone())
        else:
            m.bias = None
        return m

    @classmethod
    def from_kn_weight(cls, weight_kn: torch.Tensor, bias=None) -> "qlinear":
        """A weight STORED [in_features, out_features] (y = x @ W + b: Hugging Face GPT-2's Conv1D): per-output-channel quantisation along the strided axis of the
        matrix as it lies in memory (kernel K2 — no transposed bf16 copy is made), then ONE transpose of the int8 codes into the [N, K] layout the GEMM streams.
        Codes and scales are those of from_linear on the transposed weight, bit for bit (the same max and the same quotients, reduced along the other axis)."""
        w = weight_kn.detach()
        L.require_gpu(w, "qlinear.from_kn_weight(weight)")
        if w.dim() != 2:
            raise ValueError("from_kn_weight expects a 2-D [in_features, out_features] weight")
        qw = quantize(w, axis=0)                                    # int_data [K, N], scale [N]
        K, N = w.shape
        sub = QTensor(qw.int_data.t().contiguous(), qw.scale, 1, w.dtype, torch.Size((N, K)))
        return cls.from_qtensor(sub, bias.detach().clone() if bias is not None else None)

    @classmethod
    def from_qtensor(cls, qw: QTensor, bias=None) -> "qlinear":
        if qw.axis != 1 or qw.int_data.dim() != 2:
            raise ValueError("weight QTensor must be [N,K] quantised per output channel (axis=-1)")
        m = cls.__new__(cls)
        nn.Module.__init__(m)
        m.out_features, m.in_features = qw.int_data.shape
        m.register_buffer("wq", qw.int_data)
        m.register_buffer("ws", qw.scale)
        if bias is not None:
            m.register_buffer("bias", bias)
        else:
            m.bias = None
        return m

    def forward(self, x) -> torch.Tensor:
        """x: a float tensor [..., K] (quantised per token here, K1), or a per-token QTensor made by quantize() /
        silu_mul_quantize() (its codes and scales go straight to the GEMM; output dtype = its orig_dtype)."""
        if x.shape[-1] != self.in_features:
            raise ValueError(f"qlinear: expected last dim {self.in_features}, got {x.shape[-1]}")
        if isinstance(x, QTensor) and x.axis != 1:
            raise ValueError("qlinear: a QTensor input must be quantised per token (axis=-1)")
        wq, kp = self._wq_for_gemm()
        if kp != self.in_features:
            return self._padded_forward(x, wq, kp)
        if isinstance(x, QTensor):
            y = qlinear_s8(x.int_data.reshape(-1, self.in_features), x.scale, self.wq, self.ws, self.bias, x.orig_dtype)
            return y.reshape(*x.shape[:-1], self.out_features)
        return qlinear_dyn(x, self.wq, self.ws, self.bias)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class FusedQLinear(_KPadded, nn.Module):
    """Horizontal fusion of projections that share one input (q/k/v, gate/up): the int8 weights are concatenated
    along N, the activation is quantised ONCE and one GEMM launch fills all outputs (better tile count on 256 CUs
    than separate N=1024 GEMMs).  Exact: weight scales are per output row, so concatenation changes no value.
    forward() returns one tensor per fused projection (views of the fused output)."""

    def __init__(self, parts):
        super().__init__()
        parts = list(parts)
        if not parts or any(p.in_features != parts[0].in_features for p in parts):
            raise ValueError("FusedQLinear: projections must share in_features")
        if any((p.bias is None) != (parts[0].bias is None) for p in parts):
            raise ValueError("FusedQLinear: either all or none of the projections may have a bias")
        self.in_features = parts[0].in_features
        self.splits = [p.out_features for p in parts]
        self.out_features = sum(self.splits)
        self.register_buffer("wq", torch.cat([p.wq for p in parts], dim=0).contiguous())
        self.register_buffer("ws", torch.cat([p.ws for p in parts], dim=0).contiguous())
        if parts[0].bias is not None:
            self.register_buffer("bias", torch.cat([p.bias for p in parts], dim=0).contiguous())
        else:
            self.bias = None

    @classmethod
    def from_linears(cls, *linears: nn.Linear) -> "FusedQLinear":
        return cls([qlinear.from_linear(l) for l in linears])

    def forward(self, x):
        """x: a float tensor [..., K] or a per-token QTensor (e.g. from rmsnorm_quantize)."""
        if x.shape[-1] != self.in_features or (isinstance(x, QTensor) and x.axis != 1):
            raise ValueError("FusedQLinear: input must be [..., in_features], quantised per token if a QTensor")
        wq, kp = self._wq_for_gemm()
        if kp != self.in_features:
            return torch.split(self._padded_forward(x, wq, kp), self.splits, dim=-1)
        xq = x if isinstance(x, QTensor) else quantize(x, axis=-1)
        y = qlinear_s8(xq.int_data.reshape(-1, self.in_features), xq.scale, self.wq, self.ws, self.bias, xq.orig_dtype)
        y = y.reshape(*x.shape[:-1], self.out_features)
        return torch.split(y, self.splits, dim=-1)


class GatedMLP(nn.Module):
    """The gated MLP block of BASELINE config 3 on the int8 path: down(silu(gate(x)) * up(x)).
    gate and up share one activation quantisation and one GEMM launch (FusedQLinear); silu*mul is fused into the
    quantisation of down's input (silu_mul_quantize), so between the two GEMMs only int8 codes + row scales exist."""

    def __init__(self, gate_up: FusedQLinear, down: qlinear):
        super().__init__()
        if len(gate_up.splits) != 2 or gate_up.splits[0] != gate_up.splits[1] or gate_up.splits[0] != down.in_features:
            raise ValueError("GatedMLP: gate and up must both map to down.in_features")
        self.gate_up, self.down = gate_up, down

    @classmethod
    def from_linears(cls, gate: nn.Linear, up: nn.Linear, down: nn.Linear) -> "GatedMLP":
        return cls(FusedQLinear.from_linears(gate, up), qlinear.from_linear(down))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        g, u = self.gate_up(x)
        return self.down(silu_mul_quantize(g, u))


def _is_silu(act) -> bool:
    return isinstance(act, nn.SiLU) or type(act).__name__ in ("SiLU", "SiLUActivation") or act is torch.nn.functional.silu


def _as_gated_mlp(mod: nn.Module):
    """A module shaped like a Llama MLP — Linear children gate_proj / up_proj / down_proj and a SiLU act_fn — as GatedMLP."""
    g, u, d = (getattr(mod, n, None) for n in ("gate_proj", "up_proj", "down_proj"))
    if not all(isinstance(l, nn.Linear) for l in (g, u, d)) or not _is_silu(getattr(mod, "act_fn", None)):
        return None
    if g.in_features != u.in_features or g.out_features != u.out_features or d.in_features != g.out_features:
        return None
    if (g.bias is None) != (u.bias is None):
        return None
    return GatedMLP.from_linears(g, u, d)


def _is_conv1d(mod: nn.Module) -> bool:
    """transformers.pytorch_utils.Conv1D (GPT-2 family): a linear layer whose weight is stored [in_features, out_features] (recognised by shape, not by import)."""
    w = getattr(mod, "weight", None)
    return (type(mod).__name__ == "Conv1D" and isinstance(getattr(mod, "nf", None), int) and isinstance(w, torch.Tensor) and w.dim() == 2
            and w.shape[1] == mod.nf and not list(mod.children()))


def swap_linears(model: nn.Module, predicate=None, fuse_gated_mlp: bool = False) -> nn.Module:
    """Replace every nn.Linear (for which predicate(name, module) is true) by qlinear, in place; Hugging Face Conv1D layers (GPT-2: weight stored [K, N]) likewise,
    through qlinear.from_kn_weight (per-channel quantisation along the strided axis, kernel K2).
    fuse_gated_mlp=True additionally replaces whole gated-MLP blocks (gate_proj / up_proj / down_proj + SiLU, the
    Llama-family MLP) by GatedMLP: one fused gate+up GEMM, silu*mul fused into the quantisation, the down GEMM."""
    for name, child in list(model.named_children()):
        if fuse_gated_mlp and (predicate is None or predicate(name, child)):
            fused = _as_gated_mlp(child)
            if fused is not None:
                setattr(model, name, fused)
                continue
        if isinstance(child, nn.Linear) and (predicate is None or predicate(name, child)):
            setattr(model, name, qlinear.from_linear(child))
        elif _is_conv1d(child) and (predicate is None or predicate(name, child)):
            setattr(model, name, qlinear.from_kn_weight(child.weight, child.bias))      # Hugging Face GPT-2's Conv1D: y = x @ W[K, N] + b
        else:
            swap_linears(child, predicate, fuse_gated_mlp)
    return model
