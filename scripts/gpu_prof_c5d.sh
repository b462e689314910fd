#!/bin/bash
# kernel-time breakdown of config 5 at the reference's default regulariser (n=512, order 3, 131k columns, 1e6 samples)
export TMPDIR=/tmp
o=gpurun_out/prof_c5d
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python3 scripts/gpu_c5d_trace.py verbose=0 "$@" > $o/log.txt 2>&1
f=$(find $o -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/c5d_kernel_stats_r4.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print("%-70s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
grep -v "^\[gml\]" $o/log.txt | tail -2 | cut -c1-500
find $o -name "*.csv" -size +1M -delete; find $o -name "*.db" -delete
