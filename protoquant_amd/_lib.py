"""ctypes binding of libpq_hip.so (include/pq_hip.h).  There is NO fallback: if the HIP library is
missing or fails to load, every entry point raises — the product never routes through oracle/ or a
CPU/eager path."""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must be imported first: libpq_hip.so shares torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpq_hip.so")
ABI_VERSION = 1

PQ_BF16, PQ_FP16, PQ_F32 = 0, 1, 2
_DT = {torch.bfloat16: PQ_BF16, torch.float16: PQ_FP16, torch.float32: PQ_F32}

EXPORTS = (
    "pq_version", "pq_last_error", "pq_quant_rowwise", "pq_quant_colwise", "pq_dequant",
    "pq_gemm_s8s8s32", "pq_qlinear_s8", "pq_qlinear_workspace_bytes", "pq_gemm_variant_name",
    "pq_selftest_fast_quotient", "pq_selftest_half_encode", "pq_selftest_silu_short", "pq_qlinear_dyn", "pq_qlinear_dyn_workspace_bytes", "pq_silu_mul_quant_rowwise",
    "pq_rmsnorm_quant_rowwise", "pq_set_option", "pq_qlinear_s8_t", "pq_qlinear_t_workspace_bytes",
    "pq_silu_mul_rowamax", "pq_silu_mul_quant_rowwise_amax", "pq_qlinear_s8_kslabs", "pq_qlinear_kslabs_workspace_bytes", "pq_qlinear_kslabs_workspace_bytes_for", "pq_kslabs_way_name",
    "pq_quant_rowamax", "pq_quant_rowwise_amax",
)

_lib = None
i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t


class PQError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libpq_hip.so once; raise loudly when it is absent (build it with __graft_entry__.build()
    or `make -C protoquant_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PQError(f"{LIB_PATH} not found: build it with `make -C protoquant_amd/csrc` "
                      "(protoquant_amd has no CPU/eager fallback)")
    L = ctypes.CDLL(LIB_PATH)
    L.pq_version.restype = i32
    L.pq_last_error.restype = ctypes.c_char_p
    L.pq_gemm_variant_name.restype = ctypes.c_char_p
    L.pq_gemm_variant_name.argtypes = [i64, i64, i64, i64, i64]
    L.pq_qlinear_workspace_bytes.restype = sz
    L.pq_qlinear_workspace_bytes.argtypes = [i64, i64, i64]
    L.pq_quant_rowwise.restype = i32
    L.pq_quant_rowwise.argtypes = [vp, i32, i64, i64, i64, vp, i64, vp, vp]
    L.pq_quant_colwise.restype = i32
    L.pq_quant_colwise.argtypes = [vp, i32, i64, i64, i64, vp, i64, vp, vp]
    L.pq_dequant.restype = i32
    L.pq_dequant.argtypes = [vp, i64, vp, i32, i64, i64, vp, i64, i32, vp]
    L.pq_gemm_s8s8s32.restype = i32
    L.pq_gemm_s8s8s32.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, vp]
    L.pq_qlinear_s8.restype = i32
    L.pq_qlinear_s8.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
    L.pq_qlinear_s8_t.restype = i32
    L.pq_qlinear_s8_t.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
    L.pq_qlinear_t_workspace_bytes.restype = sz
    L.pq_qlinear_t_workspace_bytes.argtypes = [i64, i64, i64]
    L.pq_qlinear_dyn_workspace_bytes.restype = sz
    L.pq_qlinear_dyn_workspace_bytes.argtypes = [i64, i64, i64]
    L.pq_qlinear_dyn.restype = i32
    L.pq_qlinear_dyn.argtypes = [vp, i32, i64, vp, i64, vp, vp, vp, i64, i64, i64, i64, vp, sz, vp]
    L.pq_silu_mul_rowamax.restype = i32
    L.pq_silu_mul_rowamax.argtypes = [vp, i64, vp, i64, i32, i64, i64, vp, vp]
    L.pq_silu_mul_quant_rowwise_amax.restype = i32
    L.pq_silu_mul_quant_rowwise_amax.argtypes = [vp, i64, vp, i64, i32, i64, i64, vp, vp, i64, vp, vp]
    L.pq_quant_rowamax.restype = i32
    L.pq_quant_rowamax.argtypes = [vp, i32, i64, i64, i64, vp, vp]
    L.pq_quant_rowwise_amax.restype = i32
    L.pq_quant_rowwise_amax.argtypes = [vp, i32, i64, i64, i64, vp, vp, i64, vp, vp]
    L.pq_qlinear_kslabs_workspace_bytes.restype = sz
    L.pq_qlinear_kslabs_workspace_bytes.argtypes = [i64, i64, i64, i64]
    L.pq_qlinear_kslabs_workspace_bytes_for.restype = sz
    L.pq_qlinear_kslabs_workspace_bytes_for.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, i64]
    L.pq_kslabs_way_name.restype = ctypes.c_char_p
    L.pq_kslabs_way_name.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, i64, sz]
    L.pq_qlinear_s8_kslabs.restype = i32
    L.pq_qlinear_s8_kslabs.argtypes = [vp, i64, i64, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
    L.pq_silu_mul_quant_rowwise.restype = i32
    L.pq_silu_mul_quant_rowwise.argtypes = [vp, i64, vp, i64, i32, i64, i64, vp, i64, vp, vp, i64, vp]
    L.pq_rmsnorm_quant_rowwise.restype = i32
    L.pq_rmsnorm_quant_rowwise.argtypes = [vp, i64, vp, ctypes.c_float, i32, i64, i64, vp, i64, vp, vp, i64, vp]
    L.pq_set_option.restype = i32
    L.pq_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.pq_selftest_fast_quotient.restype = i32
    L.pq_selftest_fast_quotient.argtypes = [vp, vp, i64, vp, vp]
    L.pq_selftest_half_encode.restype = i32
    L.pq_selftest_half_encode.argtypes = [i32, vp, vp]
    L.pq_selftest_silu_short.restype = i32
    L.pq_selftest_silu_short.argtypes = [i32, vp, vp]
    if L.pq_version() != ABI_VERSION:
        raise PQError(f"libpq_hip.so ABI {L.pq_version()} != expected {ABI_VERSION}")
    _lib = L
    return L


def check(status: int, what: str):
    if status != 0:
        raise PQError(f"{what} failed (status {status}): {lib().pq_last_error().decode()}")


def set_option(name: str, value) -> None:
    """Change a behaviour switch of the loaded library (pq_set_option): PQ_FORCE_VARIANT, PQ_NO_TAILSPLIT, PQ_NO_SPLITK,
    PQ_SKINNY_RB.  value None / "" restores the default.  The environment variables are read once at the first call."""
    v = "" if value is None else str(value)
    check(lib().pq_set_option(name.encode(), v.encode()), f"pq_set_option({name})")


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"protoquant_amd supports bf16/fp16/fp32 tensors, got {dt}") from None


def require_gpu(t: torch.Tensor, name: str):
    if t.device.type != "cuda":
        raise PQError(f"{name} is on {t.device}: protoquant_amd runs on MI355X (HIP) only and has no "
                      "CPU fallback — move the tensor to the GPU")


def stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def row_major_2d(t: torch.Tensor) -> torch.Tensor:
    """A 2-D view whose last dim is contiguous (stride(1) == 1); copies only if it must."""
    if t.dim() != 2:
        raise ValueError("expected a 2-D tensor")
    if t.shape[1] > 0 and t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
