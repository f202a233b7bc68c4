#!/bin/bash
# round 6, GPU call 3: LDS bank-conflict phases (PMC) + fresh rocprofv3 kernel stats of configs[2] / [3] + the workload lines with cpu_baseline and traffic
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
bash $R/tools/pmc_lds_conflicts.sh $1
cd /tmp && export TMPDIR=/tmp
for W in mlp llama8b; do
  A="--workload $W --steps 200 --warmup 20"; [ $W = llama8b ] && A="--workload llama8b --steps 3"
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -- python3 $R/bench.py $A --no-cpu-baseline > $OUT/${W}_under_rocprof.json 2> $OUT/${W}_under_rocprof.err
  f=$(ls $OUT/prof_$W/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${W}_kernel_stats.csv
  rm -rf $OUT/prof_$W
done
python3 $R/bench.py --workload mlp --steps 200 --warmup 20 > $OUT/mlp_block.json 2> $OUT/mlp_block.err
python3 $R/bench.py --workload llama8b --steps 3 > $OUT/llama8b_model.json 2> $OUT/llama8b_model.err
python3 $R/bench.py --workload llama70b-shard --steps 5 > $OUT/llama70b_shard.json 2> $OUT/llama70b_shard.err
ls -la $OUT; tail -2 $OUT/*.err
