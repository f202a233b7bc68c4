cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python3 tools/k1_feed_probe.py - PQ_K1_RPW=2 PQ_K1_LDS=65536 PQ_K1_LDS=49152 PQ_K1_LDS=40000 PQ_K1_RPW=2,PQ_K1_LDS=65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/k1_feed.txt
python3 tools/k1_feed_probe.py --rows 16384 - PQ_K1_RPW=2 PQ_K1_LDS=65536 PQ_K1_LDS=40000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3a/k1_feed.txt
