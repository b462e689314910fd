#!/bin/bash
# timeline of the last learn() on the 128-node shard (kernel trace): gpu_shard_timeline.sh [nl] [prec] -> gpurun_out/timeline_<nl>_<prec>.txt
export TMPDIR=/tmp
nl=${1:-128}; prec=${2:-i8w}
o=gpurun_out/tl_$$; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/p -- python3 scripts/gpu_shard_trace.py $nl $prec 0 3 > $o/log.txt 2>&1
t=$(find $o/p -name "*kernel_trace.csv" | head -1)
python3 - "$t" > gpurun_out/timeline_${nl}_${prec}.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_kind' in r['Kernel_Name']]
seg = rows[idx[-1]:]
t0 = int(seg[0]['Start_Timestamp']); prev = t0
tot = {}
for r in seg:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('gml::', '')[:44]
    print(f"{s/1e3:9.1f} +{(e-s)/1e3:7.1f} gap {(s-prev)/1e3:6.1f}  {name}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    tot[name] = tot.get(name, 0) + (e - s); tot['(gaps)'] = tot.get('(gaps)', 0) + max(0, s - prev)
    prev = e
print("---- totals of this learn(), us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"{v/1e3:9.1f}  {k}")
PY
tail -3 $o/log.txt; rm -rf $o
sed -n '/---- totals/,$p' gpurun_out/timeline_${nl}_${prec}.txt
