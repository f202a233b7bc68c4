#!/bin/bash
# Dev tool (GPU box): the whole -m gpu suite with the 30 slowest tests, then smoke().  usage: bash tools/gpu_suite.sh <outdir under gpurun_out>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests/ -x -q -m gpu --durations=30 > $OUT/pytest_full.txt 2>&1
tail -42 $OUT/pytest_full.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
