set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 600 python tools/ab_gemm.py r1=tools/ab/libpq_r1.so new=protoquant_amd/libpq_hip.so --shapes 4096x4096x4096,4096x4096x14336,2048x11008x4096,2048x4096x11008,4096x1024x4096,4096x1024x8192,8192x8192x8192 > gpurun_out/r2a/ab.log 2>&1
cat gpurun_out/r2a/ab.log
timeout 300 python tools/ab_gemm.py r1=tools/ab/libpq_r1.so new=protoquant_amd/libpq_hip.so --shapes 4096x4096x4096 --dtype f32 >> gpurun_out/r2a/ab.log 2>&1
timeout 300 python tools/ab_gemm.py r1=tools/ab/libpq_r1.so new=protoquant_amd/libpq_hip.so --shapes 4096x4096x4096 --bias >> gpurun_out/r2a/ab.log 2>&1
timeout 300 python tools/ab_gemm.py r1=tools/ab/libpq_r1.so new=protoquant_amd/libpq_hip.so --shapes 4096x4096x4096 --eager >> gpurun_out/r2a/ab.log 2>&1
tail -8 gpurun_out/r2a/ab.log
