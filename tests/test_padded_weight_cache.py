"""CPU (-m "not gpu"): the lazily built, zero-padded copy of a module's int8 weight (in_features not a multiple of 128: protoquant_amd/qlinear.py _KPadded) must be rebuilt whenever
the weight it was made from changes — by version (in-place edits), by object (a fresh tensor that lands on a freed address), and by ADDRESS:
`module.wq.data = other` and `wq.set_(...)` keep the Python object and do not bump its version counter."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _module(K=200, N=24):
    import protoquant_amd as pq
    m = pq.qlinear(K, N, bias=False)
    m.wq = torch.nn.Parameter(torch.randint(-127, 128, (N, K), dtype=torch.int8), requires_grad=False) if isinstance(m.wq, torch.nn.Parameter) else torch.randint(-127, 128, (N, K), dtype=torch.int8)
    return m


def test_padded_copy_follows_the_weight():
    m = _module()
    K = m.in_features
    w1, kp = m._wq_for_gemm()
    assert kp == 256 and w1.shape == (m.out_features, 256) and torch.equal(w1[:, :K], m.wq) and int(w1[:, K:].abs().sum()) == 0
    assert m._wq_for_gemm()[0] is w1                                    # cached while nothing changes
    # 1. `.data =` assignment: same Python object, version NOT bumped, another address
    other = torch.randint(-127, 128, tuple(m.wq.shape), dtype=torch.int8)
    v0 = m.wq._version
    m.wq.data = other
    assert m.wq._version == v0
    w2, _ = m._wq_for_gemm()
    assert w2 is not w1 and torch.equal(w2[:, :K], other)
    # 2. set_(): same object, storage swapped
    third = torch.randint(-127, 128, tuple(m.wq.shape), dtype=torch.int8)
    m.wq.set_(third)
    w3, _ = m._wq_for_gemm()
    assert w3 is not w2 and torch.equal(w3[:, :K], third)
    # 3. an in-place edit bumps the version
    m.wq.add_(1)
    w4, _ = m._wq_for_gemm()
    assert w4 is not w3 and torch.equal(w4[:, :K], m.wq)
    # 4. a view into a larger buffer at another storage offset (same base address)
    big = torch.randint(-127, 128, (2 * m.out_features, K), dtype=torch.int8)
    m.wq.set_(big[:m.out_features])
    wa, _ = m._wq_for_gemm()
    m.wq.set_(big[m.out_features:])
    wb, _ = m._wq_for_gemm()
    assert wb is not wa and torch.equal(wb[:, :K], big[m.out_features:])


def test_multiple_of_128_needs_no_copy():
    m = _module(K=256)
    w, kp = m._wq_for_gemm()
    assert kp == 256 and w is m.wq
