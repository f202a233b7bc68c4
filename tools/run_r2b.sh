cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -5 gpurun_out/r2b/pytest.log
(cd protoquant_amd/csrc && make ABLATION=1 -j16 > /dev/null 2>&1)
timeout 600 python tools/ablate.py > gpurun_out/r2b/ablate.log 2>&1
cat gpurun_out/r2b/ablate.log
