#!/bin/bash
# Dev tool (GPU box), round 6: WHICH phase of gemm_s8_sp256 raises SQ_LDS_BANK_CONFLICT (2.62e5 per launch against 0 for the hipBLASLt kernel)?
# One counter pass (--kernel-trace + --pmc only) per form of the 4096^3 launch: the product (bf16 out: the epilogue stages through LDS with ds_write_b64), fp16, f32 and int32
# output (ds_write_b128 staging), and — dev build, tools/libpq_hip_abl.so — the same asm K-loop with NO epilogue (PQ_GEMM_DBG=1032), without fragment reads (asm variant 7) and
# without LDS-DMA (variant 6).  Usage: bash tools/pmc_lds_conflicts.sh <outdir under gpurun_out>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$R/protoquant_amd/libpq_hip.so
A=$R/tools/libpq_hip_abl.so
run() { # name env lib dtype opts...
  local name=$1 envs=$2 lib=$3 dt=$4; shift 4
  ( export $envs; timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/lds_$name -- python3 $R/tools/pmc_lds_driver.py $lib $dt "$@" > $OUT/lds_$name.log 2>&1 )
}
run product_bf16 PQ_X=0 $P bf16
run product_fp16 PQ_X=0 $P fp16
run product_f32 PQ_X=0 $P f32
run product_i32 PQ_X=0 $P i32
run hiploop_bf16 PQ_X=0 $P bf16 PQ_SP256_ASM=0
if [ -f $A ]; then
  run abl_stamps_only PQ_GEMM_DBG=1024 $A bf16
  run abl_no_epilogue PQ_GEMM_DBG=1032 $A bf16
  run abl_no_fragment_reads PQ_GEMM_DBG=1024 $A bf16 PQ_SP256_ASM=7
  run abl_no_lds_dma PQ_GEMM_DBG=1024 $A bf16 PQ_SP256_ASM=6
fi
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
lines = []
for d in sorted(glob.glob(out + "/lds_*/")):
    name = os.path.basename(d.rstrip("/"))[4:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "gemm_s8" in row["Kernel_Name"]:
                agg[row["Kernel_Name"].split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in agg.items():
        g = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else float("nan")
        lines.append(f"{name:24s} {k:64s} n={len(c.get('SQ_LDS_BANK_CONFLICT', [])):3d}  SQ_LDS_BANK_CONFLICT {g('SQ_LDS_BANK_CONFLICT'):10.4g}  SQ_LDS_IDX_ACTIVE {g('SQ_LDS_IDX_ACTIVE'):10.4g}  "
                     f"SQ_INSTS_LDS {g('SQ_INSTS_LDS'):10.4g}  SQ_ACTIVE_INST_LDS {g('SQ_ACTIVE_INST_LDS'):10.4g}  conflict / idx_active {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.4f}")
open(out + "/lds_conflicts.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/lds_*/
