"""GPU parity at MAXIMUM sizes (-m gpu): tensors with more than 2^31 elements — byte and element offsets past the 32-bit range in
every kernel family (K1/K2/dequant, the producers, every GEMM tile's epilogue, the int32 twin) — checked against the oracle on the
rows and columns around the 2^31-element boundary, the first and the last ones.  288 GB of HBM hold these comfortably; the host
only ever sees the sampled rows."""
import numpy as np
import pytest
import torch

from oracle import qspec_numpy as Q
from tests.gpu_util import bits

pytestmark = pytest.mark.gpu

COLS = 4096
ROWS = (1 << 31) // COLS + 48          # 524 336 rows x 4096 = 2^31 + 196 608 elements
EDGE = (1 << 31) // COLS               # the row whose first element is element 2^31
SAMPLE = sorted({0, 1, 77, EDGE - 2, EDGE - 1, EDGE, EDGE + 1, EDGE + 17, ROWS - 2, ROWS - 1})


@pytest.fixture(scope="module")
def pq():
    import protoquant_amd
    from protoquant_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return protoquant_amd


@pytest.fixture(scope="module")
def big_x():
    g = torch.Generator(device="cuda"); g.manual_seed(99)
    x = torch.empty((ROWS, COLS), dtype=torch.bfloat16, device="cuda")
    step = 1 << 16
    for r0 in range(0, ROWS, step):          # in pieces: torch.randn of 2^31 elements in one call needs 8 GB of f32 on the side
        n = min(step, ROWS - r0)
        x[r0:r0 + n] = (torch.randn((n, COLS), device="cuda", generator=g) * (1.0 + (r0 // step) % 5)).to(torch.bfloat16)
    return x


def _rows(t, rows):
    return bits(t[torch.tensor(rows, device=t.device)])


def test_rowwise_quant_and_dequant_past_2g_elements(pq, big_x):
    q = pq.quantize(big_x, axis=-1)
    assert q.int_data.shape == (ROWS, COLS) and q.scale.shape == (ROWS,)
    xs = _rows(big_x, SAMPLE)
    wq, ws = Q.quantize(xs, 0, 1)
    assert np.array_equal(_rows(q.int_data, SAMPLE), wq)
    assert np.array_equal(bits(q.scale[torch.tensor(SAMPLE, device="cuda")]), ws.view(np.uint32))
    d = pq.dequantize(q)
    assert d.shape == (ROWS, COLS) and d.dtype == torch.bfloat16
    assert np.array_equal(_rows(d, SAMPLE), Q.dequantize(wq, ws, 1, 0))
    # nothing between the samples was skipped: every row has a scale > 0 and a code of magnitude 127
    assert bool((q.scale > 0).all())
    assert bool((q.int_data.abs().amax(dim=1) == 127).all())


def test_colwise_quant_past_2g_elements(pq, big_x):
    q = pq.quantize(big_x, axis=0)
    assert q.int_data.shape == (ROWS, COLS) and q.scale.shape == (COLS,)
    cols = [0, 1, 63, 64, 2047, COLS - 1]
    xc = bits(big_x[:, cols].contiguous())                     # [ROWS, 6]
    wq, ws = Q.quantize(xc, 0, 0)
    assert np.array_equal(q.int_data[:, cols].contiguous().cpu().numpy(), wq)
    assert np.array_equal(bits(q.scale[cols]), ws.view(np.uint32))
    assert bool((q.int_data.abs().amax(dim=0) == 127).all())


def test_producers_past_2g_elements(pq, big_x):
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    u = big_x.flip(1)                                           # a second operand without another 4 GB of random numbers
    q = pq.silu_mul_quantize(big_x, u)
    gs, us = _rows(big_x, SAMPLE), _rows(u, SAMPLE)
    wq, ws, _ = Q.silu_mul_quantize(gs, us, 0)
    assert np.array_equal(_rows(q.int_data, SAMPLE), wq)
    assert np.array_equal(bits(q.scale[torch.tensor(SAMPLE, device="cuda")]), ws.view(np.uint32))
    w = (torch.rand(COLS, device="cuda", generator=g) + 0.5).to(torch.bfloat16)
    qn = pq.rmsnorm_quantize(big_x, w, 1e-5)
    wq, ws, _, _ = Q.rmsnorm_quantize(gs, bits(w), 1e-5, 0)
    assert np.array_equal(_rows(qn.int_data, SAMPLE), wq)
    assert np.array_equal(bits(qn.scale[torch.tensor(SAMPLE, device="cuda")]), ws.view(np.uint32))


def _every_element(y, xq, wq, xs, ws, bias):
    """All of y against the same arithmetic in stock torch ops on the GPU, in row blocks: with K = 256 the f32 matmul is exact (|acc| < 2^24) and the epilogue is
    two f32 multiplies, one add and one RNE cast — no division, nothing torch-ROCm rounds differently (the oracle pins the sampled rows; this pins the rest to them)."""
    wf = wq.float()
    for r0 in range(0, y.shape[0], 1 << 15):
        r1 = min(y.shape[0], r0 + (1 << 15))
        ref = (((xq[r0:r1].float() @ wf.T) * xs[r0:r1, None]) * ws[None, :] + bias.float()[None, :]).to(torch.bfloat16)
        assert torch.equal(ref.view(torch.int16), y[r0:r1].view(torch.int16)), f"rows {r0}..{r1}"


# y of more than 2^31 elements, both ways round: tall (M large) and wide (N large); K small so the int matmul of the sampled rows is cheap
@pytest.mark.parametrize("variant", ["auto", "sp256_16", "sp128_16", "ring128", "ring64x128", "ring64x64", "ring128x160", "generic"])
@pytest.mark.parametrize("tall", [True, False])
def test_gemm_outputs_past_2g_elements(pq, pq_opt, variant, tall):
    pq_opt("PQ_FORCE_VARIANT", "" if variant == "auto" else variant)
    big, small, K = ROWS + 5, 4100, 256                         # ragged on both edges
    M, N = (big, small) if tall else (small, big)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    xq = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    wq = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    xs = torch.rand(M, device="cuda", generator=g) * 0.01 + 1e-4
    ws = torch.rand(N, device="cuda", generator=g) * 0.01 + 1e-4
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    y = pq.qlinear_s8(xq, xs, wq, ws, bias, torch.bfloat16)
    assert y.shape == (M, N)
    if tall:
        rows = sorted({0, 255, EDGE * COLS // N - 1, EDGE * COLS // N, EDGE * COLS // N + 1, M - 6, M - 1})
        acc = Q.gemm_s8s8s32(xq[rows].cpu().numpy(), wq.cpu().numpy())
        want = Q.epilogue(acc, xs[rows].cpu().numpy(), ws.cpu().numpy(), bits(bias), 0)
        assert np.array_equal(bits(y[rows]), want)
    else:
        cols = sorted({0, 255, 256, (1 << 31) // M - 1, (1 << 31) // M, N // 2 + 3, N - 6, N - 1})
        acc = Q.gemm_s8s8s32(xq.cpu().numpy(), wq[cols].cpu().numpy())
        want = Q.epilogue(acc, xs.cpu().numpy(), ws[cols].cpu().numpy(), bits(bias[cols]), 0)
        assert np.array_equal(bits(y[:, cols].contiguous()), want)
    _every_element(y, xq, wq, xs, ws, bias)
    del y
    if variant in ("auto", "sp256_16", "generic"):              # the int32 twin: 8.6 GB of accumulators
        acc_gpu = pq.int_mm(xq, wq)
        if tall:
            assert np.array_equal(acc_gpu[rows].cpu().numpy(), Q.gemm_s8s8s32(xq[rows].cpu().numpy(), wq.cpu().numpy()))
        else:
            assert np.array_equal(acc_gpu[:, cols].contiguous().cpu().numpy(), Q.gemm_s8s8s32(xq.cpu().numpy(), wq[cols].cpu().numpy()))


def test_transposed_output_past_2g_elements(pq):
    """pq_qlinear_s8_t (the column-sharded configuration's y^T): y^T[n][m] == y[m][n] with both tensors past 2^31 elements."""
    M, N, K = ROWS + 5, 4100, 256
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    xq = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    wq = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    xs = torch.rand(M, device="cuda", generator=g) * 0.01 + 1e-4
    ws = torch.rand(N, device="cuda", generator=g) * 0.01 + 1e-4
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    y = pq.qlinear_s8(xq, xs, wq, ws, bias, torch.bfloat16)
    _every_element(y, xq, wq, xs, ws, bias)
    yt = pq.qlinear_s8_t(xq, xs, wq, ws, bias, torch.bfloat16)
    assert yt.shape == (N, M)
    for n0 in range(0, N, 512):
        assert torch.equal(yt[n0:n0 + 512].t().contiguous().view(torch.int16), y[:, n0:n0 + 512].contiguous().view(torch.int16)), f"columns {n0}.."


@pytest.mark.parametrize("M", [1, 8, 16])
def test_weight_streaming_gemm_past_2g_weight_bytes(pq, M):
    """Decode-like passes (the weight-streaming kernel): a weight matrix of 34 GB — row offsets n * ldw far past 2^31 bytes — and, at M = 8 / 16, an output past
    2^31 elements.  Every element against stock torch ops in column blocks (K = 128: exact in f32)."""
    N, K = (1 << 28) + 37, 128
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    xq = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    wq = torch.empty((N, K), dtype=torch.int8, device="cuda")
    step = 1 << 24
    for n0 in range(0, N, step):
        n = min(step, N - n0)
        wq[n0:n0 + n] = torch.randint(-127, 128, (n, K), dtype=torch.int8, device="cuda", generator=g)
    xs = torch.rand(M, device="cuda", generator=g) * 0.01 + 1e-4
    ws = torch.rand(N, device="cuda", generator=g) * 0.01 + 1e-4
    y = pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16)
    assert y.shape == (M, N)
    xf = xq.float()
    for n0 in range(0, N, step):
        n1 = min(N, n0 + step)
        ref = (((xf @ wq[n0:n1].float().T) * xs[:, None]) * ws[None, n0:n1]).to(torch.bfloat16)
        assert torch.equal(ref.view(torch.int16), y[:, n0:n1].contiguous().view(torch.int16)), f"columns {n0}..{n1}"
    # the last rows of the weight matrix against the oracle as well
    cols = [0, (1 << 24) - 1, 1 << 24, N - 38, N - 1]
    acc = Q.gemm_s8s8s32(xq.cpu().numpy(), wq[cols].cpu().numpy())
    want = Q.epilogue(acc, xs.cpu().numpy(), ws[cols].cpu().numpy(), None, 0)
    assert np.array_equal(bits(y[:, cols].contiguous()), want)
