cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q > gpurun_out/r2e/pytest_llama.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/pytest_llama.log
tail -12 gpurun_out/r2e/pytest_llama.log
timeout 900 python bench.py --workload llama8b --steps 5 > gpurun_out/r2e/llama8b_model.json 2> gpurun_out/r2e/llama8b_model.err; cat gpurun_out/r2e/llama8b_model.json; grep -v amdgpu.ids gpurun_out/r2e/llama8b_model.err | tail -15
timeout 900 python bench.py --workload llama8b --steps 5 --no-layer-fusion > gpurun_out/r2e/llama8b_model_nofuse.json 2> gpurun_out/r2e/llama8b_model_nofuse.err; cat gpurun_out/r2e/llama8b_model_nofuse.json; grep -v amdgpu.ids gpurun_out/r2e/llama8b_model_nofuse.err | tail -5
timeout 600 python bench.py --workload llama8b-linears --steps 5 --norms > gpurun_out/r2e/llama8b_linears.json 2>&1; tail -2 gpurun_out/r2e/llama8b_linears.json
timeout 600 python bench.py --workload mlp --steps 200 > gpurun_out/r2e/mlp.json 2>&1; tail -1 gpurun_out/r2e/mlp.json
