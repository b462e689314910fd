#!/bin/bash
# kernel-time breakdown of the order-3 config (n=512, 131k columns, 1e6 samples)
export TMPDIR=/tmp
o=gpurun_out/prof_c5
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python3 scripts/gpu_c5.py 512 1000000 1.2 > $o/log.txt 2>&1
f=$(find $o -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-60s calls %5s total %9.2f ms avg %9.3f ms max %9.3f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
grep -v "^\[gml\]" $o/log.txt | tail -3 | cut -c1-500
find $o -name "*.csv" -size +1M -delete; find $o -name "*.db" -delete
