#!/bin/bash
# kernel-time breakdown of one of the scripts/gpu_configs.py cases: bash scripts/gpu_prof_cfg.sh c4shard
export TMPDIR=/tmp
o=gpurun_out/prof_cfg_$1
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python3 scripts/gpu_configs.py $1 > $o/log.txt 2>&1
f=$(find $o -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-60s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
tail -2 $o/log.txt | cut -c1-400
find $o -name "*.csv" -size +1M -delete; find $o -name "*.db" -delete
