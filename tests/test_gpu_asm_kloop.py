"""-m gpu: the hand-allocated asm K-loop of the 256 x 256 split-ring tile (protoquant_amd/csrc/kloop_p3_asm.inc) at every place it can be entered
and left: K = NT x 128 for NT = 5 .. 17 covers 1 .. 13 loop tiles, i.e. every exit position of the six-tile ring turn (twice) and the three closing
tiles in each ring phase.  Per K: the int32 accumulator of sampled rows == torch._int_mm on the host (the primitive the contract names), the whole
accumulator and the fused outputs (bf16 without bias, f32 and fp16 with a bias, the transposed form) == the HIP K-loop of the same kernel
(PQ_SP256_ASM=0) bit for bit, on a grid with ragged edge tiles in both directions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pq():
    import protoquant_amd
    from protoquant_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return protoquant_amd


@pytest.mark.parametrize("NT", list(range(5, 18)))
def test_asm_kloop_every_entry_and_exit(pq, pq_opt, NT):
    from protoquant_amd import _lib
    M, N, K = 3400, 3333, NT * 128                    # 14 x 14 tiles of 256 (ragged last row and column): more than 160 -> the 256 x 256 tile
    assert b"sp256" in _lib.lib().pq_gemm_variant_name(M, N, K, K, K)
    g = torch.Generator(device="cuda").manual_seed(NT)
    a = torch.randint(-128, 128, (M, K), device="cuda", generator=g, dtype=torch.int8)
    b = torch.randint(-128, 128, (N, K), device="cuda", generator=g, dtype=torch.int8)
    xs = torch.rand(M, device="cuda", generator=g) * 1e-2 + 1e-4
    ws = torch.rand(N, device="cuda", generator=g) * 1e-2 + 1e-4
    outs = {}
    for mode in ("1", "0"):
        pq_opt("PQ_SP256_ASM", mode)
        bias16 = (torch.randn(N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)) * 0.05)
        outs[mode] = (pq.int_mm(a, b),
                      pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16),
                      pq.qlinear_s8(a, xs, b, ws, bias16, torch.float32),
                      pq.qlinear_s8(a, xs, b, ws, bias16.to(torch.float16), torch.float16),
                      pq.qlinear_s8_t(a, xs, b, ws, bias16.to(torch.bfloat16), torch.bfloat16))
    torch.cuda.synchronize()
    rows = np.sort(np.random.default_rng(NT).choice(M, 48, replace=False))
    want = torch._int_mm(a[torch.from_numpy(rows).cuda()].cpu(), b.cpu().t()).numpy()
    assert np.array_equal(outs["1"][0][torch.from_numpy(rows).cuda()].cpu().numpy(), want), f"asm K-loop, NT={NT}: accumulator differs from torch._int_mm"
    for x1, x0, what in zip(outs["1"], outs["0"], ("int32", "bf16", "f32+bias", "fp16+bias", "transposed bf16+bias")):
        assert torch.equal(x1.contiguous().view(torch.uint8), x0.contiguous().view(torch.uint8)), f"asm vs HIP K-loop, NT={NT}: {what} differs"
