#!/bin/bash
# Where the waves of the pass kernels spend their cycles (raw SQ counters, separate passes).
export TMPDIR=/tmp
o=gpurun_out/prof_stall
rm -rf $o; mkdir -p $o
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_IFETCH SQ_ACTIVE_INST_MISC"; do
  d=$o/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-learn --no-host-learn --no-f64 > $d.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for f in glob.glob('gpurun_out/prof_stall/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        per[(row['Kernel_Name'], row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
    for (k, _, c), v in per.items():
        name = k.split('(')[0].replace('void ', '')
        if 'fwd_i8' in name or 'bwd_i8' in name:
            res[name][c] = max(res[name].get(c, 0.0), v)
json.dump(res, open('gpurun_out/prof_stall/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
tail -3 $o/*.log | grep -i -E "error|invalid|not" | head
find $o -name "*.csv" -size +1M -delete; find $o -name "*.db" -delete
