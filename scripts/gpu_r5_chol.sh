#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r5_chol; rm -rf $o; mkdir -p $o
python3 -m pytest tests/test_gpu_newton_solve.py -x -q -m gpu 2>&1 | tail -5
rocprofv3 --kernel-trace --stats --output-format csv -d $o/p -- python3 scripts/gpu_newton_ubench.py 128 30 64 100 128 190 > $o/ubench.log 2>&1
f=$(find $o/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'newton_chol' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows: print(r['Kernel_Name'][:30], r['Grid_Size_X'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us')
PY
cat $o/ubench.log | grep "max rel"
rm -rf $o/p
python3 scripts/gpu_tune_shard.py ";0=65536" i8w 128
python3 scripts/gpu_tune_shard.py "" i8w 1024
