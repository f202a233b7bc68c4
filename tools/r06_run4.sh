#!/bin/bash
# round 6, GPU call 4: the whole -m gpu suite on the half-swap epilogue + smoke + the LDS conflict counter pass + the driver's bench command
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests/ -x -q -m gpu --durations=30 > $OUT/pytest_full.txt 2>&1
tail -45 $OUT/pytest_full.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/pmc_lds_conflicts.sh $1 2>&1 | grep -v "^$" | cut -c1-260
cd $R; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err; python3 -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_us'], d['verified'])"
