cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 900 python tools/race_screen.py > gpurun_out/r2m/race_screen.log 2>&1; tail -22 gpurun_out/r2m/race_screen.log
FUZZ_SECONDS=240 timeout 600 python tools/fuzz_variants.py > gpurun_out/r2m/fuzz_variants.log 2>&1; tail -5 gpurun_out/r2m/fuzz_variants.log
FUZZ_SECONDS=60 FUZZ_SEED=7 timeout 300 python tests/fuzz_quant.py > gpurun_out/r2m/fuzz_quant.log 2>&1; tail -3 gpurun_out/r2m/fuzz_quant.log
