#!/bin/bash
# kernel stats of learn() on the 128-node shard with a variant library: gpu_prof_variant.sh TAG [args of gpu_shard_trace.py]
export TMPDIR=/tmp
tag=$1; shift
o=gpurun_out/prof_$tag
rm -rf $o; mkdir -p $o
[ "$tag" = tree ] || export GML_LIB_OVERRIDE=gpurun_ab/libgml_$tag.so
rocprofv3 --kernel-trace --stats --output-format csv -d $o/p -- python3 scripts/gpu_shard_trace.py ${@:-128 i8w 0 5} > $o/log.txt 2>&1
f=$(find $o/p -name "*kernel_stats.csv" | head -1); cp "$f" $o/kernel_stats.csv
rm -rf $o/p
tail -2 $o/log.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$o/kernel_stats.csv")))[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:8.1f} us")
PY
