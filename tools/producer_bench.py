"""Times the producer-fused quantisation silu(g)*u -> int8 (pq_silu_mul_quant_rowwise) against the unfused pair
(torch-ROCm silu*mul, then K1) on BASELINE config 3's intermediate (2048 x 11008, bf16) and the 8B/70B widths.
HIP events over back-to-back launches after a warm-up; algorithmic bytes: fused 2*2 B read + 1 B write per element."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import protoquant_amd as pq  # noqa: E402


QUICK = '--quick' in sys.argv


def timeit(fn, iters=200, warm=50):
    if QUICK:
        iters, warm = 20, 5
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    for M, I in ((2048, 11008), (4096, 14336), (4096, 28672), (4096, 4096)):
        gu = torch.randn(M, 2 * I, device="cuda").to(torch.bfloat16)
        g, u = gu[:, :I], gu[:, I:]
        t_f = timeit(lambda: pq.silu_mul_quantize(g, u))
        t_fh = timeit(lambda: pq.silu_mul_quantize(g, u, return_h=True))
        t_e = timeit(lambda: torch.nn.functional.silu(g) * u)
        h = torch.nn.functional.silu(g) * u
        t_q = timeit(lambda: pq.quantize(h))
        alg = M * I * 5 + 4 * M
        print(f"{M:5d} x {I:6d} bf16  fused {t_f:7.1f} us ({alg / t_f / 1e6:6.2f} TB/s of algorithmic bytes)   fused+h {t_fh:7.1f} us   "
              f"unfused: torch silu*mul {t_e:7.1f} us + K1 {t_q:6.1f} us = {t_e + t_q:7.1f} us   speed-up {(t_e + t_q) / t_f:4.2f}x")


def rms():
    for M, H in ((4096, 4096), (4096, 8192), (16384, 4096)):
        x = torch.randn(M, H, device="cuda").to(torch.bfloat16)
        w = torch.ones(H, device="cuda", dtype=torch.bfloat16)

        def eager():
            xf = x.float()
            return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16)

        t_f = timeit(lambda: pq.rmsnorm_quantize(x, w, 1e-5))
        t_e = timeit(eager)
        h = eager()
        t_q = timeit(lambda: pq.quantize(h))
        alg = M * H * 3 + 4 * M
        print(f"rmsnorm {M:5d} x {H:5d} bf16  fused {t_f:7.1f} us ({alg / t_f / 1e6:5.2f} TB/s of algorithmic bytes)   unfused: torch eager "
              f"chain {t_e:7.1f} us + K1 {t_q:6.1f} us   speed-up {(t_e + t_q) / t_f:4.2f}x")


if __name__ == "__main__":
    main()
    rms()
