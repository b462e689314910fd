"""Interleaved timing of one build of libgml_hip under different environments (argv: tag=VAR=value[,VAR=value] ...)."""
import subprocess, sys, json, os
cases = []
for a in sys.argv[1:]:
    tag, rest = a.split("=", 1)
    cases.append((tag, dict(kv.split("=", 1) for kv in rest.split(",") if kv)))
for rnd in range(3):
    for tag, extra in cases:
        env = dict(os.environ)
        env.update(extra)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "3", "--no-cpu", "--no-learn", "--no-f64"], env=env,
                             capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(tag, (round(d["ms_per_step"], 3), round(d["roofline"]["fwd_ms"], 3), round(d["roofline"]["bwd_ms"], 3)), flush=True)
        except Exception:
            print(tag, "FAILED", out.stdout[-500:], out.stderr[-1500:], flush=True)
