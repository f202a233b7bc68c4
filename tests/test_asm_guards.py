"""CPU: guards on the ISA that ships in libpq_hip.so (disassembled from the built library, no GPU needed).

The GEMM kernels keep hazards by hand that hipcc knows nothing about; a compiler upgrade or an innocent source edit must not break
them silently:
  * M0 — the LDS-DMA destination base — is written inside inline asm without a clobber (gemm_s8_fast.hip glds16_*; kloop_p3_asm.inc),
    relying on "no compiler-generated user of M0": every instruction that names m0 must be an SALU write of it that feeds the next
    global_load_lds_dwordx4, and every LDS-DMA must have such a write right in front of it.
  * v_fma_mixlo_f16 rounds a product ONCE where QSPEC rounds to binary32 first (the fp16 single-rounding defect): it must
    not appear in any kernel.
  * the split-ring GEMM (HIP and asm K-loops) must not spill: scratch reloads inside the K-loop share vmcnt with the DMA pieces.
  * kloop_p3_asm.inc must be what tools/gen_kloop_asm.py generates.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "protoquant_amd", "libpq_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    """[(disassembly text, kernel metadata text)] of every gfx950 code object bundled in libpq_hip.so"""
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not available")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "protoquant_amd", "csrc"), "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    d = tmp_path_factory.mktemp("co")
    fat = str(d / "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB, str(d / "unused.so")], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(MAGIC, data)]
    assert starts, "no offload bundle in libpq_hip.so"
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(data)
        b, co = str(d / f"b{k}.bin"), str(d / f"d{k}.co")
        open(b, "wb").write(data[s:e])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        if os.path.getsize(co) == 0:
            continue
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
        meta = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        out.append((dis, meta))
    assert len(out) >= 5, "expected the device code of five translation units"
    return out


def _instructions(dis):
    """(mnemonic, operand text) of every instruction line of an llvm-objdump listing, kernel by kernel"""
    kernels, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m and cur is not None:
            cur.append((m.group(1), m.group(2)))
    return kernels


def test_m0_only_feeds_lds_dma(code_objects):
    n_dma = 0
    for dis, _ in code_objects:
        for name, ins in _instructions(dis).items():
            for i, (op, args) in enumerate(ins):
                names_m0 = re.search(r"\bm0\b", args) is not None
                if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in args):
                    n_dma += 1
                    # the M0 write that feeds it: within the 4 preceding instructions, only s_nop / MFMA / ds_read in between
                    ok = False
                    for back in range(1, 5):
                        if i - back < 0:
                            break
                        pop, pargs = ins[i - back]
                        if pop in ("s_mov_b32", "s_add_u32") and pargs.startswith("m0,"):
                            ok = True
                            break
                        assert pop == "s_nop" or pop.startswith("v_mfma") or pop.startswith("ds_read"), f"{name}: {pop} {pargs} between the M0 write and its LDS-DMA"
                    assert ok, f"{name}: LDS-DMA #{i} without an M0 write right in front of it"
                elif names_m0:
                    assert op in ("s_mov_b32", "s_add_u32") and args.startswith("m0,"), f"{name}: unexpected user of M0: {op} {args}"
                    nxt = [o for o, _ in ins[i + 1:i + 5]]
                    assert any(o.startswith("global_load_lds") for o in nxt), f"{name}: M0 write #{i} does not feed an LDS-DMA"
    assert n_dma > 500, "the GEMM kernels' LDS-DMA instructions were not found"


def test_no_single_rounding_fp16_convert(code_objects):
    for dis, _ in code_objects:
        assert "v_fma_mixlo" not in dis and "v_fma_mixhi" not in dis and "v_mad_mixlo" not in dis


def test_split_ring_gemm_does_not_spill(code_objects):
    seen = 0
    for _, meta in code_objects:
        for blk in re.split(r"\n\s+- \.agpr_count", meta):
            m = re.search(r"\.name:\s+(\S*gemm_s8_(?:sp256|p3_persist)\S*)", blk)
            if not m:
                continue
            seen += 1
            assert re.search(r"\.private_segment_fixed_size:\s+0\b", blk), f"{m.group(1)} uses scratch"
            assert re.search(r"\.vgpr_spill_count:\s+0\b", blk), f"{m.group(1)} spills VGPRs"
    assert seen >= 14          # (incl. the six fused split-K instantiations: their pinned registers are hipcc's own assignment of the product kernel)


def test_generated_kloop_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_kloop_asm.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
