cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
for i in 6 10 16 24; do cp protoquant_amd/libpq_hip.so /tmp/libpq_pf$i.so; done
for rot in 24 1; do
echo "== rotate $rot"
timeout 600 python3 tools/ab_gemm.py pf0=protoquant_amd/libpq_hip.so pf6=/tmp/libpq_pf6.so@PQ_RING_PF=6 pf10=/tmp/libpq_pf10.so@PQ_RING_PF=10 pf16=/tmp/libpq_pf16.so@PQ_RING_PF=16 pf24=/tmp/libpq_pf24.so@PQ_RING_PF=24 --shapes 4096x1024x28672,512x4096x4096,1024x4096x4096,4096x1024x8192 --rotate-weights $rot --per-graph 24 --rounds 9 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3m/ring_pf.txt
