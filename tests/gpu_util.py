"""Helpers for the -m gpu parity tests: numpy <-> torch bit plumbing and oracle comparison."""
import numpy as np
import torch

TD = {0: torch.bfloat16, 1: torch.float16, 2: torch.float32}


def to_gpu(a: np.ndarray, code: int) -> torch.Tensor:
    """numpy storage (uint16 bit patterns for half types) -> cuda tensor of the real dtype."""
    if code == 2:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda()
    return t.view(TD[code])


def bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach().contiguous().cpu()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float32:
        return t.numpy().view(np.uint32)
    return t.numpy()


def same(got: torch.Tensor, want: np.ndarray, what=""):
    g = bits(got)
    w = np.asarray(want)
    if w.dtype == np.float32:
        w = w.view(np.uint32)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    nbad = int(np.count_nonzero(g != w))
    assert nbad == 0, f"{what}: {nbad} of {g.size} elements differ (first at {np.argwhere(g != w)[:3].tolist()})"


def same_f(got: torch.Tensor, want: np.ndarray, code: int, what=""):
    """Float outputs (y, dequantised tensors): NaNs compared as a CLASS (QSPEC v2 leaves the payload and sign of a NaN that
    arithmetic produces open — 0 * Inf is 0xFFC00000 on x86 and 0x7FC00000 on gfx950), everything else bit for bit."""
    from oracle import qspec_numpy as Q
    g = bits(got)
    w = np.asarray(want)
    wf = Q.to_f32(w, code)
    if w.dtype == np.float32:
        w = w.view(np.uint32)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    nan_w = np.isnan(wf)
    nan_g = np.isnan(got.detach().float().cpu().numpy())
    assert np.array_equal(nan_g, nan_w), f"{what}: NaN positions differ ({int(nan_g.sum())} vs {int(nan_w.sum())})"
    bad = (g != w) & ~nan_w
    assert not bad.any(), f"{what}: {int(bad.sum())} of {g.size} elements differ (first at {np.argwhere(bad)[:3].tolist()})"
