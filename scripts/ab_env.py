"""Interleaved A/B of an environment switch on the same GPU: python scripts/ab_env.py VAR valA valB [rounds]"""
import subprocess, sys, json, os
var, a, b = sys.argv[1:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for rnd in range(rounds):
    for val in (a, b):
        env = dict(os.environ); env[var] = val
        out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--no-cpu", "--no-learn"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(var, val, "ms/step %.3f fwd %.3f bwd %.3f" % (d["ms_per_step"], d["roofline"]["fwd_ms"], d["roofline"]["bwd_ms"]), flush=True)
