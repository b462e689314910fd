#!/bin/bash
# kernel stats of learn() on the 128-node shard and on config 2 (where the iteration, not the pass kernels, is what costs)
export TMPDIR=/tmp
o=gpurun_out/prof_small
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/shard128 -- python3 scripts/gpu_shard128.py 128 > $o/shard128.log 2>&1
f=$(find $o/shard128 -name "*kernel_stats.csv" | head -1); cp "$f" $o/shard128_kernel_stats.csv
t=$(find $o/shard128 -name "*kernel_trace.csv" | head -1); cp "$t" $o/shard128_kernel_trace.csv
tail -3 $o/shard128.log | head -2
cut -d, -f1-4 $o/shard128_kernel_stats.csv | cut -c1-70,200- | head -16
