#!/bin/bash
# round 6, GPU call: the planned 128 x 160 tile against the round-5 plan (PQ_NO_RING160=1, workspace as planned: fused split-K where it was) over the class incl. long K; full -m gpu suite
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
S=protoquant_amd/libpq_hip.so
for n in a b; do cp $S /tmp/pq_$n.so; done
SH=4096x1280x8192,4096x1280x4096,4096x1280x16384,4096x1280x28672,2048x2560x8192,2048x2560x14336,1024x5120x8192,768x6144x4096,512x10240x8192,1536x3200x11008
echo "# HBM-fed (40 rotating weight matrices)" > $OUT/ab_plan160.txt
timeout 900 python3 tools/ab_gemm.py round5_plan=/tmp/pq_a.so@PQ_NO_RING160=1 round6_plan=/tmp/pq_b.so --shapes $SH --rotate-weights 40 --rounds 15 >> $OUT/ab_plan160.txt 2>&1
echo "# warm (one weight matrix)" >> $OUT/ab_plan160.txt
timeout 900 python3 tools/ab_gemm.py round5_plan=/tmp/pq_a.so@PQ_NO_RING160=1 round6_plan=/tmp/pq_b.so --shapes $SH --rounds 15 >> $OUT/ab_plan160.txt 2>&1
grep -v amdgpu.ids $OUT/ab_plan160.txt
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > $OUT/pytest_full.txt 2>&1
tail -5 $OUT/pytest_full.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
