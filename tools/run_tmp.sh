cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout 900 python3 tools/race_screen.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3i/race.txt | tail -25
timeout 900 python3 tools/fuzz_variants.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r3i/fuzz.txt
cp protoquant_amd/libpq_hip.so /tmp/libpq_p2.so
for rot in 1 24; do
timeout 600 python3 tools/ab_gemm.py p3=protoquant_amd/libpq_hip.so p2=/tmp/libpq_p2.so@PQ_SP256_P3=0 --shapes 4096x4096x4096,4096x6144x4096,4096x4096x14336,4096x28672x4096 --rotate-weights $rot --per-graph 24 --rounds 11 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3i/ab_p3.txt
done
