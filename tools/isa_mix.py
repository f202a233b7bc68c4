"""Dev tool: instruction mix of the MFMA loops from `make -C protoquant_amd/csrc asm` (build/gemm_s8_fast.s).
For every kernel the innermost backward-branch loop that holds the most v_mfma instructions is taken as the steady-state K-loop;
counts are printed per loop body and per 64 (or 32) MFMAs = one K-tile of one wave.   usage: python tools/isa_mix.py [file.s] [filter]"""
import collections
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "protoquant_amd/csrc/build/gemm_s8_fast.s"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
text = open(path).read()


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and "lds" in op): return "lds_dma"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "vmem_store"
    if op == "s_waitcnt": return "s_waitcnt"
    if op == "s_barrier": return "s_barrier"
    if op == "s_nop": return "s_nop"
    if op.startswith("s_cbranch") or op == "s_branch": return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"


for m in re.finditer(r"^(_ZN2pq\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    dn = demangle(name).split("(")[0]
    if "gemm_s8" not in dn or (flt and flt not in dn):
        continue
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        lm = re.match(r"^(\.LBB\d+_\d+):", l)
        if lm:
            labels[lm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        bm = re.match(r"\s+s_cbranch\S*\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if bm and bm.group(1) in labels and labels[bm.group(1)] < i:
            seg = lines[labels[bm.group(1)]:i + 1]
            nm = sum(1 for s in seg if re.match(r"\s+v_mfma", s))
            # innermost: prefer the loop with the most MFMAs, ties -> the shortest
            if nm and (best is None or nm > best[0] or (nm == best[0] and len(seg) < len(best[1]))):
                best = (nm, seg)
    if not best:
        continue
    nm, seg = best
    cnt = collections.Counter()
    for s in seg:
        im = re.match(r"\s+([a-z_0-9]+)", s)
        if im and not s.strip().startswith((".", ";")):
            cnt[klass(im.group(1))] += 1
    per = 64 if "sp256" in dn else 32
    tiles = nm / per
    order = ["mfma", "ds_read", "lds_dma", "valu", "salu", "branch", "s_waitcnt", "s_nop", "s_barrier", "vmem_load", "ds_write", "vmem_store", "other"]
    print(f"{dn}: loop of {len(seg)} lines = {tiles:g} K-tile(s) of one wave; per K-tile: " +
          ", ".join(f"{k} {cnt[k] / tiles:g}" for k in order if cnt[k]))
