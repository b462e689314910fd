#!/bin/bash
# MFMA utilisation and LDS bank conflicts of the pass kernels (derived PMC metrics, separate passes).
export TMPDIR=/tmp
o=gpurun_out/prof_util
rm -rf $o; mkdir -p $o
for c in MfmaUtil "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  d=$o/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-learn --no-host-learn --no-weighted --no-sparse-theta > $d.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for f in glob.glob('gpurun_out/prof_util/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        per[(row['Kernel_Name'], row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
    for (k, _, c), v in per.items():
        name = k.split('(')[0].replace('void ', '')
        if 'fwd_i8' in name or 'bwd_i8' in name or 'hess_bits' in name or '_f64' in name:
            res[name][c] = max(res[name].get(c, 0.0), v)
json.dump(res, open('gpurun_out/prof_util/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $o -name "*.csv" -size +1M -delete; find $o -name "*.db" -delete
