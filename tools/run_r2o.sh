cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
timeout 600 python tools/k1s_ab.py > gpurun_out/r2o/k1s_ab.log 2>&1; cat gpurun_out/r2o/k1s_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "silu or mlp or fuzz" > gpurun_out/r2o/pytest.log 2>&1; tail -3 gpurun_out/r2o/pytest.log
timeout 300 python bench.py --workload mlp --steps 200 2>/dev/null | tail -1 | cut -c1-250
