"""Interleaved timing of builds of libgml_hip on the FP64 pass (argv: tag=path ...): prints (pass, fwd, bwd) ms."""
import subprocess, sys, json, os
libs = [a.split("=", 1) for a in sys.argv[1:]]
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for tag, path in libs:
        env = dict(os.environ)
        if path: env["GML_LIB_OVERRIDE"] = path
        out = subprocess.run([sys.executable, "bench.py", "--precision", "f64", "--steps", "10", "--warmup", "2", "--no-cpu", "--no-learn", "--no-i8x", "--no-weighted"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(tag, (round(d["ms_per_step"], 3), round(d["roofline"]["fwd_ms"], 3), round(d["roofline"]["bwd_ms"], 3)), flush=True)
        except Exception:
            print(tag, "FAILED", out.stderr[-500:], flush=True)
