import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import protoquant_amd as pq
from oracle import c_oracle as C
z = np.load("tests/golden/cfg1_32x512x512_f32.npz")
x = torch.from_numpy(z["x"]).cuda()
q = pq.quantize(x)
print("scale equal:", np.array_equal(q.scale.cpu().numpy().view(np.uint32), z["xs"].view(np.uint32)))
print(q.scale.cpu().numpy()[:4], z["xs"][:4])
g = q.int_data.cpu().numpy()
print(g[0,:16]); print(z["xq"][0,:16])
print(z["x"][0,:8]/z["xs"][0])
for cols in (4, 8, 64, 256, 260, 512):
    xx = x[:, :cols].contiguous()
    qq = pq.quantize(xx)
    wq, ws = C.quant_rowwise(xx.cpu().numpy(), 2)
    print(cols, "codes bad:", (qq.int_data.cpu().numpy() != wq).sum(), "scale bad:", (qq.scale.cpu().numpy() != ws).sum())
