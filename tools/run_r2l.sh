cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 900 python bench.py --workload llama70b-shard --steps 5 > gpurun_out/r2l/llama70b_shard.json 2> gpurun_out/r2l/llama70b_shard.err; cat gpurun_out/r2l/llama70b_shard.json; tail -3 gpurun_out/r2l/llama70b_shard.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2l/prof_l8 -- python3 $R/bench.py --workload llama8b --steps 3 > $R/gpurun_out/r2l/prof_l8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2l/prof_mlp -- python3 $R/bench.py --workload mlp --steps 200 > $R/gpurun_out/r2l/prof_mlp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2l/prof_70b -- python3 $R/bench.py --workload llama70b-shard --steps 3 > $R/gpurun_out/r2l/prof_70b.log 2>&1
cd $R
for d in prof_l8 prof_mlp prof_70b; do f=$(find gpurun_out/r2l/$d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r2l/${d}_kernel_stats.csv; rm -rf gpurun_out/r2l/$d; done
head -12 gpurun_out/r2l/prof_l8_kernel_stats.csv | cut -c1-220
tail -1 gpurun_out/r2l/prof_l8.log | cut -c1-400
