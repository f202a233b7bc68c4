#!/bin/bash
# Dev tool (GPU box): the round's final measurement set.  Usage: bash tools/r05_measure.sh <outdir under gpurun_out>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_line_under_rocprof.err
f=$(ls $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv
rm -rf $OUT/prof
python3 $R/bench.py --mode tp --steps 20 --warmup 5 > $OUT/bench_tp_world1.json 2> $OUT/bench_tp_world1.err
python3 $R/bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 > $OUT/bench_tp2_gloo.json 2> $OUT/bench_tp2_gloo.err
python3 $R/bench.py --workload mlp --steps 200 --warmup 20 > $OUT/mlp_block.json 2> $OUT/mlp_block.err
python3 $R/bench.py --workload llama8b --steps 3 > $OUT/llama8b_model.json 2> $OUT/llama8b_model.err
ls -la $OUT
