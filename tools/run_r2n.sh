cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r2n/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2n/pytest.log
tail -4 gpurun_out/r2n/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2n/smoke.log 2>&1; tail -2 gpurun_out/r2n/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2n/bench.json 2> gpurun_out/r2n/bench.err; cat gpurun_out/r2n/bench.json | cut -c1-2500
