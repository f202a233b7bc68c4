cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 1200 python -m pytest tests/test_gpu_sharded_forms.py tests/test_gpu_model_shapes.py -m gpu -q -x > gpurun_out/r2c/pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/pytest_new.log
tail -15 gpurun_out/r2c/pytest_new.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/bench_n1.json 2> gpurun_out/r2c/bench_n1.err; cat gpurun_out/r2c/bench_n1.json; tail -3 gpurun_out/r2c/bench_n1.err
timeout 600 python bench.py > gpurun_out/r2c/bench_default.json 2> gpurun_out/r2c/bench_default.err; cat gpurun_out/r2c/bench_default.json; tail -3 gpurun_out/r2c/bench_default.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu > gpurun_out/r2c/bench_tp2_dry.json 2> gpurun_out/r2c/bench_tp2_dry.err; cat gpurun_out/r2c/bench_tp2_dry.json; tail -5 gpurun_out/r2c/bench_tp2_dry.err
