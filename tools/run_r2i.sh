cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 900 python tools/ab_gemm.py base=protoquant_amd/libpq_hip.so order1=tools/ab/libpq_order1.so order2=tools/ab/libpq_order2.so --shapes 4096x4096x4096,4096x4096x14336,8192x8192x8192 --rounds 21 > gpurun_out/r2i/ab_order.log 2>&1
cat gpurun_out/r2i/ab_order.log
timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -x > gpurun_out/r2i/pytest_bench.log 2>&1; tail -5 gpurun_out/r2i/pytest_bench.log
