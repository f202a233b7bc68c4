cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2v/tr -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2v/bench_prof.json 2> $R/gpurun_out/r2v/bench_prof.err
cd $R
f=$(find gpurun_out/r2v/tr -name '*kernel_trace.csv' | head -1)
python3 tools/trace_instep.py $f > gpurun_out/r2v/instep.txt 2>&1; cat gpurun_out/r2v/instep.txt
rm -rf gpurun_out/r2v/tr
timeout 600 python3 tools/vendor_gemm.py > gpurun_out/r2v/vendor.txt 2>&1; cat gpurun_out/r2v/vendor.txt
