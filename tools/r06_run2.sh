#!/bin/bash
# round 6, GPU call 2: the split bench (contract, multirank, survivor line), the new GPU tests, the driver's command, the workload lines + their PMC traffic
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests/test_bench_contract.py tests/test_gpu_multirank.py tests/test_gpu_rmsnorm_vs_eager.py tests/test_gpu_llama.py tests/test_gpu_int8_exchange.py tests/test_gpu_fuzz.py -x -q -m gpu -s > $OUT/pytest_a.txt 2>&1
tail -15 $OUT/pytest_a.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err; tail -3 $OUT/bench_line.err; head -c 600 $OUT/bench_line.json; echo
bash tools/pmc_workloads.sh $1 mlp llama8b llama70b-shard
