"""Interleaved timing of several builds of libgml_hip (argv: tag=path ...) on one GPU; prints (step, fwd, bwd) ms."""
import subprocess, sys, json, os
libs = [a.split("=", 1) for a in sys.argv[1:]]
res = {t: [] for t, _ in libs}
for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for tag, path in libs:
        env = dict(os.environ)
        if path: env["GML_LIB_OVERRIDE"] = path
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "3", "--no-cpu", "--no-learn", "--no-f64", "--no-i8x", "--no-weighted"] + os.environ.get("AB_ARGS", "").split(), env=env,
                             capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(tag, "FAILED", out.stdout[-300:], out.stderr[-800:], flush=True)
            continue
        res[tag].append((round(d["ms_per_step"], 3), round(d["roofline"]["fwd_ms"], 3), round(d["roofline"]["bwd_ms"], 3)))
        print(tag, res[tag][-1], flush=True)
