"""Summarise rocprofv3 --pmc output: per kernel, the per-dispatch total of each counter for the largest
dispatch (the full-size launch).  usage: pmc_summarize.py OUT.json DIR [DIR ...]"""
import csv, glob, json, os, sys
from collections import defaultdict
out, dirs = sys.argv[1], sys.argv[2:]
res = defaultdict(dict)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)  # (kernel, dispatch, counter) -> sum over dimensions
        with open(f) as fh:
            for row in csv.DictReader(fh):
                per[(row["Kernel_Name"], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (k, _, c), v in per.items():
            name = k.split("(")[0].replace("void ", "")
            res[name][c + "_KB"] = max(res[name].get(c + "_KB", 0.0), v)
# the kernels the counters belong to: bench.py compares this with the source it runs and says so when they differ
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # (the hash of every source file a pass's kernels come from: bench.pass_kernels_sha256)
summary = {k: v for k, v in res.items() if k.startswith("gml::")}
summary["_pass_kernels_sha256"] = bench.pass_kernels_sha256()
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if "fwd_i8" in k or "bwd_i8" in k or "_f64" in k}, indent=1))
