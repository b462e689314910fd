#!/bin/bash
# per-launch durations of the Hessian-vector GEMM kernels of config 5 at the default regulariser, grouped by grid size
export TMPDIR=/tmp
o=gpurun_out/prof_c5d_trace
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o -- python3 scripts/gpu_c5d_trace.py verbose=0 "$@" > $o/log.txt 2>&1
f=$(find $o -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_fwd_i8<2" in n or "k_bwd_i8<1, 2>" in n:
        key = ("fwd" if "k_fwd" in n else "bwd", int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        a = acc[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
print("kernel grid(x,y,z) calls total_ms avg_ms ms_per_kblock")
for k, (c, t) in rows[:40]:
    nb = k[1] * k[2] * k[3]
    print(k[0], k[1:], c, round(t, 1), round(t / c, 3), round(t / c / nb * 1000, 3))
PY
find $o -name "*.csv" -size +1M -delete; find $o -name "*.db" -delete
