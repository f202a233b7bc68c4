"""Dev tool (GPU box): N launches of the 4096^3 GEMM through one library for a counter pass (tools/pmc_lds_conflicts.sh).
usage: python3 tools/pmc_lds_driver.py <lib.so> <bf16|fp16|f32|i32> [OPT=VAL ...]     (PQ_GEMM_DBG in the environment selects a dev-build ablation)"""
import ctypes
import os
import sys

import torch

i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
L.pq_qlinear_s8.restype = i32
L.pq_qlinear_s8.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
L.pq_gemm_s8s8s32.restype = i32
L.pq_gemm_s8s8s32.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, vp]
L.pq_set_option.restype = i32
L.pq_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    assert L.pq_set_option(k.encode(), v.encode()) == 0, kv
M = N = K = 4096
torch.manual_seed(0)
xq = (torch.randn(M, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
wq = (torch.randn(N, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
xs = torch.rand(M, device="cuda") * 0.01
ws = torch.rand(N, device="cuda") * 0.01
dt = sys.argv[2]
td, code = {"bf16": (torch.bfloat16, 0), "fp16": (torch.float16, 1), "f32": (torch.float32, 2), "i32": (torch.int32, 3)}[dt]
y = torch.empty((M, N), dtype=td, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    if dt == "i32":
        rc = L.pq_gemm_s8s8s32(xq.data_ptr(), K, wq.data_ptr(), K, y.data_ptr(), N, M, N, K, st)
    else:
        rc = L.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, code, M, N, K, None, 0, st)
    assert rc == 0
torch.cuda.synchronize()
