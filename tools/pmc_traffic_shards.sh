#!/bin/bash
# Dev tool (GPU box): HBM/fabric traffic (PMC FETCH_SIZE, WRITE_SIZE: separate passes, no other trace domain) of the GEMM kernel at the shard
# widths a tp run of the headline uses (N / G for G = 2, 4, 8).  Usage: tools/pmc_traffic_shards.sh <outdir>
set -u
OUT=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
for N in 2048 1024 512; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/n${N}_$C -- python3 $R/bench.py --N $N --steps 20 --warmup 5 --repeats 3 --warmup-seconds 0.3 --no-cpu-baseline --no-gpu-context --no-graph > $R/$OUT/n${N}_$C.log 2>&1
  done
done
python3 - "$R/$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
for N in (2048, 1024, 512):
    vals = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{out}/n{N}_{C}/*/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if "gemm_s8" in row["Kernel_Name"] and row["Counter_Name"] == C:
                    agg[row["Kernel_Name"].split("(")[0][:70]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            vals.setdefault(k, {})[C] = (sum(v) / len(v), len(v))
    for k, d in vals.items():
        f, w = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
        print(f"4096x{N}x4096  {k}  FETCH_SIZE {f[0]:.0f} KiB (n={f[1]})  WRITE_SIZE {w[0]:.0f} KiB (n={w[1]})  -> traffic {(f[0] * 2 + w[0]) * 1024:.0f} bytes per launch (FETCH_SIZE x 2: gfx950 correction)")
PY
rm -rf $R/$OUT/n*_FETCH_SIZE $R/$OUT/n*_WRITE_SIZE
