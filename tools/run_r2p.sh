cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
timeout 600 python tools/k1n_ab.py > gpurun_out/r2p/k1n_ab.log 2>&1; cat gpurun_out/r2p/k1n_ab.log
