#!/bin/bash
# round 5, first call: trace + kernel stats of learn() on the 128-node shard at i8w
export TMPDIR=/tmp
o=gpurun_out/r5_first
rm -rf $o; mkdir -p $o
python3 scripts/gpu_shard_trace.py 128 i8w 1 5 > $o/shard128_i8w_trace.txt 2>&1
python3 scripts/gpu_shard_trace.py 128 i8x 1 5 > $o/shard128_i8x_trace.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 scripts/gpu_shard_trace.py 128 i8w 0 6 > $o/prof.log 2>&1
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); cp "$f" $o/shard128_i8w_kernel_stats.csv
t=$(find $o/prof -name "*kernel_trace.csv" | head -1); cp "$t" $o/shard128_i8w_kernel_trace.csv
rm -rf $o/prof
tail -8 $o/shard128_i8w_trace.txt
