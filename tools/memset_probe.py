"""Dev tool (GPU box): control experiment for profiles/r06_hipgraph_memset_hang.txt — torch.Tensor.zero_() captured in front of a consumer kernel, a later graph captured,
replays with the buffer dirtied in between: does the zeroing still happen in order?  (It does.)"""
import torch, sys
dev = torch.device("cuda:0")
def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g
for n in (192, 1 << 10, 1 << 20):
    t = torch.empty(n, device=dev); out = torch.empty(n, device=dev); u = torch.randn(1 << 20, device=dev); v = torch.empty_like(u)
    def fa(): t.zero_(); torch.add(t, 1.0, out=out)
    def fb(): torch.mul(u, 2.0, out=v)
    ga = capture(fa)
    gb = capture(fb)
    bad = 0
    for it in range(50):
        t.fill_(5.0); out.fill_(-1.0)
        ga.replay(); gb.replay(); torch.cuda.synchronize()
        bad += int((out != 1.0).sum().item())
    print(f"n={n}: torch zero_() node under capture, later graph captured: wrong elements over 50 replays = {bad}", flush=True)
# the same with a RAW hipMemsetAsync node (what the library used to issue for its tile tickets) instead of torch's zero_()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
for n in (192, 1 << 10, 1 << 20):
    t = torch.empty(n, device=dev); out = torch.empty(n, device=dev); u = torch.randn(1 << 20, device=dev); v = torch.empty_like(u)
    def fa():
        assert hip.hipMemsetAsync(t.data_ptr(), 0, 4 * n, torch.cuda.current_stream().cuda_stream) == 0
        torch.add(t, 1.0, out=out)
    def fb(): torch.mul(u, 2.0, out=v)
    for later in (False, True):
        ga = capture(fa)
        gb = capture(fb) if later else None
        bad = 0
        for it in range(50):
            t.fill_(5.0); out.fill_(-1.0)
            ga.replay()
            if gb is not None:
                gb.replay()
            torch.cuda.synchronize()
            bad += int((out != 1.0).sum().item())
        print(f"n={n}: RAW hipMemsetAsync node under capture, {'a later graph captured' if later else 'NO later graph'}: wrong elements over 50 replays = {bad} of {50 * n}", flush=True)
print("done")
