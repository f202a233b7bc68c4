#!/bin/bash
# Dev tool (GPU box): PMC counter passes (own runs, one group per pass) for pq_qlinear_s8 on a list of shapes.  Usage: tools/pmc_shapes.sh <outdir> <shapes> [opts]
set -u
OUT=$1; SHAPES=$2; OPTS=${3:-}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/run_shapes.py --shapes $SHAPES --reps 100 --opts "$OPTS" > $R/$OUT/$name.log 2>&1
}
mkdir -p $R/$OUT
run sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE
run grbm GRBM_GUI_ACTIVE
python3 - "$R/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "pq::gemm" not in row["Kernel_Name"]: continue
        k = row["Kernel_Name"].split("(")[0][:70] + "  grid=" + row["Grid_Size"]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c, v in sorted(m.items()):
        print(f"   {c:28s} n={len(d[c]):4d} mean={v:.4g}")
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8          # shader cycles of the launch (the counter sums the 8 XCDs)
        # SQ_LDS_IDX_ACTIVE: cycles the LDS index (address / data) pipeline is busy, summed over CUs -> busy fraction per CU
        cus = min(256, int(k.split("grid=")[1]) // 512 if "ringt" in k or "ring128" in k else 256)
        print(f"   -> launch {cyc:.0f} cycles; MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):.3f} of 1024 SIMDs; "
              f"LDS pipeline busy {m.get('SQ_LDS_IDX_ACTIVE', 0) / (cyc * 256):.3f} of 256 CUs (over {cus} busy CUs: {m.get('SQ_LDS_IDX_ACTIVE', 0) / (cyc * max(cus, 1)):.3f}); "
              f"bank-conflict cycles / LDS-busy cycles {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
