"""Interleaved A/B of two builds of libgml_hip on the same GPU (separate processes, alternating)."""
import subprocess, sys, json, os
res = {"new": [], "old": []}
for rnd in range(3):
    for tag in ("new", "old"):
        env = dict(os.environ)
        if tag == "old": env["GML_LIB_OVERRIDE"] = "graphicalmodellearning.jl_amd/libgml_hip_old.so"
        out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--no-cpu", "--no-learn"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[tag].append((d["ms_per_step"], d["roofline"]["fwd_ms"], d["roofline"]["bwd_ms"]))
        print(tag, res[tag][-1], flush=True)
