#!/bin/bash
# Dev tool (GPU box): the round's final measurement set on the final build.  Usage: bash tools/r06_measure.sh <outdir under gpurun_out> [quick]
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_line_under_rocprof.err
f=$(ls $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
rm -rf $OUT/prof
python3 $R/bench.py --workload mlp --steps 200 --warmup 20 > $OUT/mlp_block.json 2> $OUT/mlp_block.err
python3 $R/bench.py --workload llama8b --steps 3 > $OUT/llama8b_model.json 2> $OUT/llama8b_model.err
python3 $R/bench.py --workload llama70b-shard --steps 5 > $OUT/llama70b_shard.json 2> $OUT/llama70b_shard.err
[ "${2:-}" = quick ] && { ls -la $OUT; exit 0; }
python3 $R/bench.py --mode tp --steps 20 --warmup 5 > $OUT/bench_tp_world1.json 2> $OUT/bench_tp_world1.err
python3 $R/bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 > $OUT/bench_tp2_gloo.json 2> $OUT/bench_tp2_gloo.err
for W in mlp llama8b; do
  A="--workload $W --steps 200 --warmup 20"; [ $W = llama8b ] && A="--workload llama8b --steps 3"
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -- python3 $R/bench.py $A --no-cpu-baseline > $OUT/${W}_under_rocprof.json 2> $OUT/${W}_under_rocprof.err
  f=$(ls $OUT/prof_$W/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${W}_kernel_stats.csv
  rm -rf $OUT/prof_$W
done
bash $R/tools/pmc_gemm.sh gpurun_out/$1/pmc > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/pmc/*/
cd $R
( echo "# tools/race_screen.py"; timeout 1500 python3 tools/race_screen.py; echo "# tools/fuzz_variants.py (FUZZ_SECONDS=180)"; FUZZ_SECONDS=180 timeout 600 python3 tools/fuzz_variants.py; echo "# tools/fsk_stress.py"; timeout 900 python3 tools/fsk_stress.py ) > $OUT/race_screen_fuzz.txt 2>&1
timeout 1500 python3 tools/dispatch_audit.py --quick > $OUT/dispatch_audit_quick.txt 2>&1
ls -la $OUT
