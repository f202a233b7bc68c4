cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 300 tools/ubench/k1_feed 2>&1 | tee gpurun_out/r3b/k1_feed_ubench.txt
