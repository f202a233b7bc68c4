cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 1200 python -m pytest tests/test_gpu_sharded_forms.py tests/test_gpu_serialize.py -m gpu -q > gpurun_out/r2d/pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest_new.log
tail -25 gpurun_out/r2d/pytest_new.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu > gpurun_out/r2d/bench_tp2_dry.json 2> gpurun_out/r2d/bench_tp2_dry.err; cat gpurun_out/r2d/bench_tp2_dry.json; grep -v "Gloo\|^$\|amdgpu.ids\|socket.cpp" gpurun_out/r2d/bench_tp2_dry.err | tail -12
