#!/bin/bash
# Dev tool (GPU box), round 6: HBM / fabric traffic PER STEP of the non-headline workloads (BASELINE configs[2], [3], [4]-per-rank) — PMC FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes with --kernel-trace only (no other trace domain), summed over the product's kernels of every pass of the workload and divided by the number of passes
# (counted from a kernel that runs a known number of times per pass).  FETCH_SIZE x 2: the gfx950 correction of MI355X_MICROARCH.md (HBM section).
# Usage: bash tools/pmc_workloads.sh <outdir under gpurun_out> [workloads...]      -> <outdir>/pmc_workloads.txt + workload_traffic.json
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
WL=${@:-mlp llama8b llama70b-shard}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in $WL; do
  case $W in
    mlp) A="--workload mlp --steps 3 --warmup 1 --no-graph";;
    llama8b) A="--workload llama8b --steps 3";;
    llama8b-linears) A="--workload llama8b-linears --steps 3 --norms";;
    llama70b-shard) A="--workload llama70b-shard --steps 3 --no-extras";;
  esac
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 1500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -- python3 $R/bench.py $A --no-cpu-baseline > $OUT/pmc_${W}_$C.log 2>&1
  done
done
python3 - "$OUT" $WL <<'PY'
import csv, glob, json, sys, collections
out, wls = sys.argv[1], sys.argv[2:]
PRODUCT = ("gemm_s8", "quant_", "silu_mul", "rmsnorm", "splitk_reduce", "unstack_kslabs", "col_amax", "col_encode", "dequant")
MARK = {"mlp": ("silu_mul", 1), "llama8b": ("rmsnorm", 64), "llama8b-linears": ("rmsnorm", 64), "llama70b-shard": ("rmsnorm", 160)}
res, lines = {}, []
for W in wls:
    tot, cnt, per_kernel = {}, {}, collections.defaultdict(lambda: [0.0, 0.0, 0])
    for ci, C in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        s, marks = 0.0, 0
        for f in glob.glob(f"{out}/pmc_{W}_{C}/*/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if row["Counter_Name"] != C or not any(p in k for p in PRODUCT):
                    continue
                v = float(row["Counter_Value"])
                s += v
                kk = k.split("(")[0][:80]
                per_kernel[kk][ci] += v
                if ci == 0:
                    per_kernel[kk][2] += 1
                if MARK[W][0] in k:
                    marks += 1
        tot[C], cnt[C] = s, marks / MARK[W][1]
    if not cnt["FETCH_SIZE"] or cnt["FETCH_SIZE"] != cnt["WRITE_SIZE"]:
        lines.append(f"{W}: pass count mismatch {cnt}")
        continue
    passes = cnt["FETCH_SIZE"]
    traffic = (tot["FETCH_SIZE"] * 2 + tot["WRITE_SIZE"]) * 1024 / passes
    res[W] = int(traffic)
    lines.append(f"{W}: {passes:g} passes; FETCH_SIZE {tot['FETCH_SIZE'] / passes:.0f} KiB x 2 + WRITE_SIZE {tot['WRITE_SIZE'] / passes:.0f} KiB per pass -> traffic {traffic:.0f} bytes per step")
    for kk, (f_, w_, n_) in sorted(per_kernel.items(), key=lambda kv: -(kv[1][0] * 2 + kv[1][1])):
        lines.append(f"    {kk:80s} launches/pass {n_ / passes:7.1f}  fetch x2 {f_ * 2 / passes / 1024:9.1f} MiB  write {w_ / passes / 1024:9.1f} MiB per pass")
open(out + "/pmc_workloads.txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + "/workload_traffic.json", "w"), indent=1)
print("\n".join(lines))
PY
rm -rf $OUT/pmc_*_FETCH_SIZE $OUT/pmc_*_WRITE_SIZE
