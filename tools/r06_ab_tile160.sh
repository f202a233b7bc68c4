#!/bin/bash
# round 6: the 128 x 160 ring tile (forced) against the shipped dispatch, warm and HBM-fed; + the parity tests that sweep every forced variant
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_large_index.py tests/test_gpu_fuzz.py -x -q -m gpu -k "variant or fuzz or large" > $OUT/pytest_variants.txt 2>&1; tail -4 $OUT/pytest_variants.txt
S=protoquant_amd/libpq_hip.so
for n in a b c d; do cp $S /tmp/pq_$n.so; done
SH=4096x1280x8192,4096x2560x8192,4096x1280x4096,2048x1280x8192,8192x1280x8192,4096x640x8192,4096x1024x8192,4096x1920x8192
echo "# HBM-fed (40 rotating weight matrices)" > $OUT/ab_tile160.txt
timeout 900 python3 tools/ab_gemm.py default=/tmp/pq_a.so t128x160=/tmp/pq_b.so@PQ_FORCE_VARIANT=ring128x160 ring128=/tmp/pq_c.so@PQ_FORCE_VARIANT=ring128 sp128x256=/tmp/pq_d.so@PQ_FORCE_VARIANT=sp128_16 \
    --shapes $SH --rotate-weights 40 --rounds 15 >> $OUT/ab_tile160.txt 2>&1
echo "# warm (one weight matrix)" >> $OUT/ab_tile160.txt
timeout 900 python3 tools/ab_gemm.py default=/tmp/pq_a.so t128x160=/tmp/pq_b.so@PQ_FORCE_VARIANT=ring128x160 ring128=/tmp/pq_c.so@PQ_FORCE_VARIANT=ring128 sp128x256=/tmp/pq_d.so@PQ_FORCE_VARIANT=sp128_16 \
    --shapes $SH --rounds 15 >> $OUT/ab_tile160.txt 2>&1
grep -v amdgpu.ids $OUT/ab_tile160.txt
