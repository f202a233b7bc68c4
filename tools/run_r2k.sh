cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 900 python tools/ab_gemm.py base=protoquant_amd/libpq_hip.so sc1=tools/ab/libpq_dma1.so sc0=tools/ab/libpq_dma2.so --shapes 4096x4096x4096,4096x4096x14336,4096x1024x8192,2048x4096x11008 --rounds 21 > gpurun_out/r2k/ab_dma_policy.log 2>&1
cat gpurun_out/r2k/ab_dma_policy.log
(python tools/ab_gemm.py lc=protoquant_amd/libpq_hip.so --shapes 4096x4096x4096 --rounds 3000 > gpurun_out/r2k/bg.log 2>&1 &)
sleep 30
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | head -6; sleep 0.5; done > gpurun_out/r2k/smi.log 2>&1
rocm-smi --showmaxpower 2>&1 | grep -i "max" | head -3 >> gpurun_out/r2k/smi.log
cat gpurun_out/r2k/smi.log | head -40
wait
