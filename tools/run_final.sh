cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r2z/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2z/pytest.log
tail -3 gpurun_out/r2z/pytest.log
timeout 900 python tools/shapes_bench.py > gpurun_out/r2z/shapes.txt 2>&1; cat gpurun_out/r2z/shapes.txt
timeout 600 python bench.py --workload mlp --steps 200 > gpurun_out/r2z/mlp.json 2>/dev/null; cut -c1-200 gpurun_out/r2z/mlp.json
timeout 900 python bench.py --workload llama8b --steps 5 > gpurun_out/r2z/llama8b.json 2>/dev/null; cut -c1-900 gpurun_out/r2z/llama8b.json
timeout 900 python bench.py --workload llama70b-shard --steps 5 > gpurun_out/r2z/llama70b.json 2>/dev/null; cut -c1-300 gpurun_out/r2z/llama70b.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2z/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2z/prof_bench.log 2>&1
cd $R
cp $(find gpurun_out/r2z/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r2z/kernel_stats.csv; rm -rf gpurun_out/r2z/prof
head -4 gpurun_out/r2z/kernel_stats.csv | cut -c1-300
