cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 900 python tools/shapes_bench.py > gpurun_out/r2h/shapes.txt 2>&1; cat gpurun_out/r2h/shapes.txt
bash tools/pmc_gemm.sh gpurun_out/r2h/pmc > gpurun_out/r2h/pmc_summary.txt 2>&1; tail -60 gpurun_out/r2h/pmc_summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2h/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2h/prof_bench.log 2>&1
cd $R
cat $(find gpurun_out/r2h/prof -name "*kernel_stats.csv" | head -1) | head -6
tail -1 gpurun_out/r2h/prof_bench.log | cut -c1-1500
