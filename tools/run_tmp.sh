cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r2t/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2t/pytest.log
tail -3 gpurun_out/r2t/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2t/bench.json 2>/dev/null; cut -c1-2100 gpurun_out/r2t/bench.json
timeout 900 python tools/shapes_bench.py > gpurun_out/r2t/shapes.txt 2>&1; cat gpurun_out/r2t/shapes.txt
timeout 600 python bench.py --workload mlp --steps 200 > gpurun_out/r2t/mlp.json 2>/dev/null; cut -c1-200 gpurun_out/r2t/mlp.json
timeout 900 python bench.py --workload llama8b --steps 5 > gpurun_out/r2t/llama8b.json 2>/dev/null; cut -c1-900 gpurun_out/r2t/llama8b.json
timeout 900 python bench.py --workload llama70b-shard --steps 5 > gpurun_out/r2t/llama70b.json 2>/dev/null; cut -c1-300 gpurun_out/r2t/llama70b.json
