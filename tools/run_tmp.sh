cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split_rings or tail_split or half_encode" > gpurun_out/r3o/pytest.log 2>&1; tail -3 gpurun_out/r3o/pytest.log
timeout 600 python bench.py --workload mlp --steps 200 > gpurun_out/r3o/mlp.json 2>/dev/null; cut -c1-200 gpurun_out/r3o/mlp.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-context > gpurun_out/r3o/bench.json 2>/dev/null; cut -c1-250 gpurun_out/r3o/bench.json
