cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2g/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2g/pytest.log
tail -4 gpurun_out/r2g/pytest.log
timeout 900 python tools/ab_gemm.py r1=tools/ab/libpq_r1.so ring4=tools/ab/libpq_cur_copy.so@PQ_RING_LC=0 lc=protoquant_amd/libpq_hip.so --shapes 4096x1024x4096,4096x1024x8192,4096x1024x28672,512x4096x4096,1024x1024x8192,4096x128x8192,2048x4096x11008 > gpurun_out/r2g/ab_lc.log 2>&1
cat gpurun_out/r2g/ab_lc.log
