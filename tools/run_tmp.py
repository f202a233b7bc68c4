import torch, sys
sys.path.insert(0, "/root/repo")
import protoquant_amd as pq
from protoquant_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = torch.randn(4096, 4096, generator=g).to(torch.bfloat16)
xd = x.to(dev)
xf = xd.float()
amax = xf.abs().amax(dim=1, keepdim=True)
s = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
s_cpu = x.float().abs().amax(dim=1, keepdim=True) / 127.0
print("scale mismatches gpu-vs-cpu division:", int((s.cpu() != s_cpu).sum()), "of", s.numel())
q_gpu = torch.round(xf / s)
q_cpu = torch.round(x.float() / s_cpu)
print("code mismatches (gpu torch vs cpu torch):", int((q_gpu.cpu() != q_cpu).sum()), "of", q_cpu.numel())
qt = pq.quantize(xd)
print("library codes vs cpu torch:", int((qt.int_data.cpu().float() != q_cpu).sum()), " scales:", int((qt.scale.cpu() != s_cpu.flatten()).sum()))
print("library codes vs gpu torch:", int((qt.int_data.float() != q_gpu).sum()), " scales:", int((qt.scale != s.flatten()).sum()))
# selftest of the one-step encode domain
lib = L.lib()
for dt, name in ((0, "bf16"), (1, "fp16")):
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    L.check(lib.pq_selftest_half_encode(dt, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "selftest")
    torch.cuda.synchronize()
    print(name, "pairs", int(out[0]), "mismatches", int(out[1]))
