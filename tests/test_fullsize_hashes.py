"""Full-size parity against torch-generated data: tests/golden/fullsize_hashes.json holds SHA-256 digests of the WHOLE
xq / xs / wq / ws / int32 accumulator / y that oracle/torch_ref.py (QSPEC around torch._int_mm) produces at BASELINE.json's
full sizes (generated in the builder container by oracle/gen_fullsize_hashes.py from tests/synth.py's integer-only inputs).

CPU (-m "not gpu"): the input generator has not drifted; the C + numpy restatements reproduce the torch form's digests at
4096^3.  GPU (-m gpu): the HIP path through the C-ABI reproduces every digest of every case — every bit of every output at
the sizes the bench runs (4096^3: the 256 x 256 asm tile with 32 K-tiles; the cfg-3 `down` and 70B `down` shard: fused split-K)."""
import json
import os

import numpy as np
import pytest

from tests.synth import FULLSIZE_CASES, bell_bf16, fullsize_inputs, sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json")) as _f:
    _ALL = json.load(_f)
HASHES, CHAINS = _ALL["cases"], _ALL["chains"]


def test_fixture_covers_every_case():
    assert set(HASHES) == set(FULLSIZE_CASES)
    for name, (M, N, K, seed, bias) in FULLSIZE_CASES.items():
        h = HASHES[name]
        assert (h["M"], h["N"], h["K"], h["seed"], h["bias"]) == (M, N, K, seed, bias)
        assert all(len(h[k]) == 64 for k in ("x", "w", "xq", "xs", "wq", "ws", "acc", "y"))


def test_generator_is_portable_and_bell_shaped():
    a = bell_bf16(64, 4096, 7, -15)
    assert sha(a) == sha(bell_bf16(64, 4096, 7, -15))
    f = (a.astype(np.uint32) << 16).view(np.float32)
    assert abs(float(f.mean())) < 0.01 and 1.10 < float(f.std()) < 1.20 and float(np.abs(f).max()) <= 4.0
    x, w, b = fullsize_inputs("l70b_qo_shard_4096x1024x8192")
    h = HASHES["l70b_qo_shard_4096x1024x8192"]
    assert sha(x) == h["x"] and sha(w) == h["w"], "tests/synth.py drifted from the committed digests"


def test_c_and_numpy_oracles_reproduce_torch_digests_at_4096_cubed():
    """Three-way pin at the metric's size: torch form (committed digests) == C form (quantisation) + numpy form (exact float64 BLAS GEMM,
    QSPEC epilogue) on every bit of every output."""
    from oracle import c_oracle as C
    from oracle import qspec_numpy as Q
    name = "cfg2_4096x4096x4096"
    h = HASHES[name]
    x, w, b = fullsize_inputs(name)
    assert sha(x) == h["x"] and sha(w) == h["w"]
    xq, xs = C.quant_rowwise(x, 0)
    wq, ws = C.quant_rowwise(w, 0)
    assert (sha(xq), sha(xs), sha(wq), sha(ws)) == (h["xq"], h["xs"], h["wq"], h["ws"])
    acc = Q.gemm_s8s8s32(xq, wq)
    assert sha(acc) == h["acc"]
    assert sha(Q.epilogue(acc, xs, ws, None, 0)) == h["y"]


def test_c_oracle_silu_mul_reproduces_torch_eager_digests_at_cfg3_size():
    """QSPEC S1-S6 in plain C == torch's eager quantize(F.silu(g) * u) (oracle/torch_ref.py::silu_mul_quantize_ref, committed digests) on every bit of the stored
    product, the codes and the scales of a whole 2048 x 11008 bf16 intermediate — the producer-fused quantisation is pinned to torch-generated data at full size."""
    from oracle import c_oracle as C
    h = CHAINS["silu_mul_quant_2048x11008"]
    g, u = bell_bf16(2048, 11008, 41, -13), bell_bf16(2048, 11008, 42, -15)
    assert sha(g) == h["g"] and sha(u) == h["u"], "tests/synth.py drifted from the committed digests"
    q, s, hh = C.silu_mul_quant_rowwise(g, u, 0)
    assert (sha(hh), sha(q), sha(s)) == (h["h"], h["hq"], h["hs"])


@pytest.mark.gpu
def test_hip_silu_mul_quant_reproduces_torch_eager_digests():
    """K1s (pq_silu_mul_quant_rowwise) on the GPU == torch's eager quantize(F.silu(g) * u) on the CPU: whole-output digests of codes and scales."""
    import protoquant_amd as pq
    from tests.gpu_util import bits, to_gpu
    h = CHAINS["silu_mul_quant_2048x11008"]
    g, u = bell_bf16(2048, 11008, 41, -13), bell_bf16(2048, 11008, 42, -15)
    assert sha(g) == h["g"] and sha(u) == h["u"], "input generator drifted"
    hq = pq.silu_mul_quantize(to_gpu(g, 0), to_gpu(u, 0))
    assert sha(bits(hq.int_data)) == h["hq"] and sha(bits(hq.scale)) == h["hs"]


@pytest.mark.gpu
def test_hip_gated_mlp_reproduces_torch_eager_chain_digests():
    """BASELINE configs[2] end to end against torch-generated data: GatedMLP (fused gate+up GEMM -> K1s -> down) on the GPU reproduces the digests of gate, up,
    the int8 codes / scales of the intermediate and the block output that torch's own eager ops around torch._int_mm produce on the CPU
    (oracle/gen_fullsize_hashes.py: qlinear_ref x 2 -> F.silu(gate) * up -> quantize_ref -> torch._int_mm -> epilogue)."""
    import torch

    import protoquant_amd as pq
    from tests.gpu_util import bits, to_gpu
    h = CHAINS["cfg3_mlp_block_2048x4096x11008"]
    x, wg, _ = fullsize_inputs("cfg3_gate_2048x11008x4096")
    _, wu, _ = fullsize_inputs("cfg3_up_2048x11008x4096")
    _, wd, _ = fullsize_inputs("cfg3_down_2048x4096x11008")
    assert sha(x) == h["x"], "input generator drifted"
    lins = []
    for w in (wg, wu, wd):
        lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, device="cuda", dtype=torch.bfloat16)
        with torch.no_grad():
            lin.weight.copy_(to_gpu(w, 0))
        lins.append(lin)
    mlp = pq.GatedMLP.from_linears(*lins)
    xg = to_gpu(x, 0)
    gate, up = mlp.gate_up(xg)
    assert sha(bits(gate.contiguous())) == h["gate"] and sha(bits(up.contiguous())) == h["up"], "gate / up"
    hq = pq.silu_mul_quantize(gate, up)
    assert sha(bits(hq.int_data)) == h["hq"] and sha(bits(hq.scale)) == h["hs"], "K1s codes / scales"
    assert sha(bits(mlp.down(hq))) == h["y"], "down on the fused codes"
    assert sha(bits(mlp(xg))) == h["y"], "GatedMLP.forward"


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(FULLSIZE_CASES))
def test_hip_path_reproduces_torch_digests(name):
    import torch

    import protoquant_amd as pq
    from protoquant_amd import _lib
    from tests.gpu_util import bits, to_gpu
    _lib.lib()
    h = HASHES[name]
    x, w, b = fullsize_inputs(name)
    assert sha(x) == h["x"] and sha(w) == h["w"] and (b is None or sha(b) == h["b"]), "input generator drifted"
    xg, wg = to_gpu(x, 0), to_gpu(w, 0)
    bg = to_gpu(b, 0) if b is not None else None
    qw = pq.quantize(wg, axis=-1)
    qx = pq.quantize(xg, axis=-1)
    assert sha(bits(qw.int_data)) == h["wq"] and sha(bits(qw.scale)) == h["ws"], "weight codes / scales"
    assert sha(bits(qx.int_data)) == h["xq"] and sha(bits(qx.scale)) == h["xs"], "activation codes / scales"
    acc = pq.int_mm(qx.int_data, qw.int_data)
    assert sha(bits(acc)) == h["acc"], f"int32 accumulator (dispatch {_lib.lib().pq_gemm_variant_name(h['M'], h['N'], h['K'], h['K'], h['K']).decode()})"
    del acc
    y = pq.qlinear_s8(qx.int_data, qx.scale, qw.int_data, qw.scale, bg, torch.bfloat16)
    assert sha(bits(y)) == h["y"], "y (fused epilogue)"
    # the module path (K1 + K3/K4 behind qlinear.forward), whole output again
    lin = torch.nn.Linear(h["K"], h["N"], bias=b is not None, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(wg)
        if b is not None:
            lin.bias.copy_(bg)
    m = pq.qlinear.from_linear(lin)
    assert sha(bits(m(xg))) == h["y"], "qlinear module y"
