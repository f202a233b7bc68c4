cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2j/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2j/pytest.log
tail -6 gpurun_out/r2j/pytest.log
timeout 900 python tools/ab_gemm.py lc=protoquant_amd/libpq_hip.so w8=tools/ab/libpq_cur_copy.so@PQ_SP128_LC=0 --shapes 2048x4096x11008,2048x4096x4096,2048x6144x4096,512x8192x4096,2048x11008x4096,4096x6144x4096,1024x8192x8192 > gpurun_out/r2j/ab_lc128.log 2>&1
cat gpurun_out/r2j/ab_lc128.log
timeout 300 python bench.py --workload mlp --steps 200 2>/dev/null | tail -1 | cut -c1-300
(python tools/ab_gemm.py lc=protoquant_amd/libpq_hip.so --shapes 4096x4096x4096 --rounds 400 > /dev/null 2>&1 &)
sleep 6
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" | head -6; sleep 1; done > gpurun_out/r2j/smi.log 2>&1
cat gpurun_out/r2j/smi.log | head -20
wait
