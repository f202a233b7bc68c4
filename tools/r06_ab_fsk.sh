#!/bin/bash
# Dev tool (GPU box), round 6: (a) the new in-place tests, (b) the 70B `down` shard on STACKED blocks — fused split-K walking the slabs in place against the ring tile in place and the
# contiguous forms, (c) the 70B fused-qkv shard 4096 x 1280 x 8192 under 2 / 3 / 4 ticket slices of the 256 x 256 tile, the 128 x 128 ring tile and today's 128 x 256 LC tile, warm and HBM-fed.
# usage: bash tools/r06_ab_fsk.sh <outdir under gpurun_out>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_int8_exchange.py -x -q -m gpu > $OUT/pytest_int8_exchange.txt 2>&1
tail -5 $OUT/pytest_int8_exchange.txt
S=protoquant_amd/libpq_hip.so
for n in a b c d e f g; do cp $S /tmp/pq_$n.so; done
echo "# down shard, HBM-fed (40 rotating weight matrices)" > $OUT/ab_down.txt
timeout 600 python3 tools/ab_gemm.py contig=/tmp/pq_a.so stk8_fsk4=/tmp/pq_b.so@STACKED=8 stk8_ring=/tmp/pq_c.so@STACKED=8,PQ_FSK=0 contig_ring=/tmp/pq_d.so@PQ_FSK=0 stk8_fsk8=/tmp/pq_e.so@STACKED=8,PQ_FSK=8 stk4_fsk4=/tmp/pq_f.so@STACKED=4 stk2_fsk4=/tmp/pq_g.so@STACKED=2 \
    --shapes 4096x1024x28672 --rotate-weights 40 --rounds 15 >> $OUT/ab_down.txt 2>&1
echo "# down shard, warm (one weight matrix)" >> $OUT/ab_down.txt
timeout 600 python3 tools/ab_gemm.py contig=/tmp/pq_a.so stk8_fsk4=/tmp/pq_b.so@STACKED=8 stk8_ring=/tmp/pq_c.so@STACKED=8,PQ_FSK=0 contig_ring=/tmp/pq_d.so@PQ_FSK=0 \
    --shapes 4096x1024x28672 --rounds 15 >> $OUT/ab_down.txt 2>&1
cat $OUT/ab_down.txt
echo "# fused-qkv shard and neighbours, HBM-fed (40 rotating weight matrices)" > $OUT/ab_qkv.txt
timeout 900 python3 tools/ab_gemm.py default=/tmp/pq_a.so fsk2=/tmp/pq_b.so@PQ_FSK=2 fsk3=/tmp/pq_c.so@PQ_FSK=3 fsk4=/tmp/pq_d.so@PQ_FSK=4 ring128=/tmp/pq_e.so@PQ_FORCE_VARIANT=ring128 \
    --shapes 4096x1280x8192,4096x1024x8192,4096x7168x8192,2048x4096x11008 --rotate-weights 40 --rounds 15 >> $OUT/ab_qkv.txt 2>&1
echo "# warm (one weight matrix)" >> $OUT/ab_qkv.txt
timeout 900 python3 tools/ab_gemm.py default=/tmp/pq_a.so fsk2=/tmp/pq_b.so@PQ_FSK=2 fsk3=/tmp/pq_c.so@PQ_FSK=3 fsk4=/tmp/pq_d.so@PQ_FSK=4 ring128=/tmp/pq_e.so@PQ_FORCE_VARIANT=ring128 \
    --shapes 4096x1280x8192,4096x1024x8192 --rounds 15 >> $OUT/ab_qkv.txt 2>&1
cat $OUT/ab_qkv.txt
