"""Interleaved runs of several builds of libgml_hip (argv: tag=path ..., '' = the tree's build) on the 128-node shard: learn() ms, t_hess."""
import subprocess, sys, os
libs = [a.split("=", 1) for a in sys.argv[1:]]
args = os.environ.get("AB_ARGS", "128 i8w 0 5").split()
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for tag, path in libs:
        env = dict(os.environ)
        if path: env["GML_LIB_OVERRIDE"] = path
        out = subprocess.run([sys.executable, "scripts/gpu_shard_trace.py"] + args, env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if " ms " in l]
        print(tag, lines[-1] if lines else ("FAILED " + out.stderr[-500:]), flush=True)
