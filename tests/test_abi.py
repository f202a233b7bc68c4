"""CPU: the C-ABI library loads (no GPU needed) and exports every symbol include/pq_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pq_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pq_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("pq_version", "pq_last_error", "pq_quant_rowwise", "pq_quant_colwise", "pq_dequant",
              "pq_gemm_s8s8s32", "pq_qlinear_s8", "pq_qlinear_workspace_bytes"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from protoquant_amd import _lib
    L = _lib.lib()
    for s in declared_symbols():
        assert hasattr(L, s), f"libpq_hip.so does not export {s}"
    assert set(_lib.EXPORTS) <= set(declared_symbols())
    assert L.pq_version() == _lib.ABI_VERSION
    assert L.pq_qlinear_workspace_bytes(4096, 4096, 4096) == 0                       # headline shape: single pass
    assert L.pq_qlinear_workspace_bytes(4096, 1024, 8192) == 0                        # 70B q/o shard: 128 x 128 ring tiles, no workspace
    assert L.pq_qlinear_workspace_bytes(1024, 1024, 8192) == 0                         # small grid: 64 x 64 ring tiles on every CU, single pass (round 4)
    assert L.pq_gemm_variant_name(1024, 1024, 8192, 8192, 8192) == b"ring64x64_16x16x64"
    assert L.pq_set_option(b"PQ_NO_MIDM", b"1") == 0
    assert L.pq_qlinear_workspace_bytes(1024, 1024, 8192) == 4 * 1024 * 1024 * 4      # ... the round-3 plan for it: quarter-filled grid, long K: split-K x4
    assert L.pq_set_option(b"PQ_NO_MIDM", b"") == 0
    assert L.pq_gemm_variant_name(4096, 1024, 8192, 8192, 8192) == b"ring128_16x16x64"
    assert L.pq_gemm_variant_name(16, 4096, 4096, 4096, 4096) == b"skinny_16x16x64"               # decode-like: weight streaming
    assert L.pq_gemm_variant_name(4096, 4096, 4096, 4096, 4096) == b"sp256_16x16x64"
    assert L.pq_gemm_variant_name(2048, 4096, 11008, 11008, 11008) == b"sp128x256_16x16x64"      # half a round of 256-row tiles
    assert L.pq_gemm_variant_name(5, 7, 3, 3, 3) == b"generic64"


def test_rccl_library_exports_every_declared_symbol():
    src = open(os.path.join(ROOT, "include", "pq_rccl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(pq_[a-z0-9_]+)\s*\(", src)))
    from protoquant_amd import _rccl
    L = _rccl.lib()
    for s in syms:
        assert hasattr(L, s), f"libpq_rccl.so does not export {s}"
    assert set(_rccl.EXPORTS) == set(syms)
    assert L.pq_allgather_cols_workspace_bytes(8, 4096, 512, 0) == 8 * 4096 * 512 * 2
    assert L.pq_allgather_cols(None, 8, None, None, 4, 4, 0, None, 0, None) == 1          # null communicator
    assert L.pq_unstack_cols(None, None, 0, 4, 4, 0, None) == 1


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected before any HIP call."""
    from protoquant_amd import _lib
    L = _lib.lib()
    assert L.pq_quant_rowwise(None, 7, 1, 1, 1, None, 1, None, None) == 1
    assert b"dtype" in L.pq_last_error()
    assert L.pq_dequant(None, 4, None, 3, 2, 2, None, 2, 0, None) == 1
    assert L.pq_gemm_s8s8s32(None, 1, None, 1, None, 1, -1, 1, 1, None) == 1
    assert L.pq_qlinear_s8(None, 8, None, None, 8, None, None, None, 8, 9, 1, 1, 8, None, 0, None) == 1
    assert L.pq_quant_rowwise(None, 0, 0, 16, 16, None, 16, None, None) == 0     # empty is a no-op


def test_product_has_no_cpu_fallback_and_never_imports_oracle():
    import torch
    import protoquant_amd as pq
    from protoquant_amd import _lib
    with pytest.raises(_lib.PQError):
        pq.quantize(torch.randn(2, 8))
    with pytest.raises(_lib.PQError):
        pq.qlinear(8, 8)(torch.randn(2, 8, dtype=torch.bfloat16))
    pkg = os.path.join(ROOT, "protoquant_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_no_memset_node_left_in_the_product():
    """round 6: a hipMemsetAsync node inside a captured graph hung the fused split-K on replay once another graph had been captured (profiles/r06_hipgraph_memset_hang.txt):
    the library initialises its counters and scratch with kernels of its own"""
    for f in os.listdir(os.path.join(ROOT, "protoquant_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            for ln in open(os.path.join(ROOT, "protoquant_amd", "csrc", f)):
                code = ln.split("//")[0]
                assert "hipMemset" not in code, (f, ln.strip()[:120])
