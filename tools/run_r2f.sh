cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_parity.py -m gpu -q -k "llama or quant" > gpurun_out/r2f/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2f/pytest.log
tail -5 gpurun_out/r2f/pytest.log
timeout 600 python tools/k1_ab.py > gpurun_out/r2f/k1_ab.log 2>&1; cat gpurun_out/r2f/k1_ab.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2f/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2f/prof_bench.log 2>&1
cd $R
find gpurun_out/r2f/prof -name "*kernel_stats.csv" | head -3
head -8 $(find gpurun_out/r2f/prof -name "*kernel_stats.csv" | head -1)
tail -2 gpurun_out/r2f/prof_bench.log | cut -c1-600
