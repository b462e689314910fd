"""How the CPU oracle's blocked objective/gradient scales with host threads on this box (cpu_baseline context)."""
import os, sys, time, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    K, n, nn = 200000, 1024, 512
    spins = np.where(rng.random((K, n)) < 0.5, 1, -1).astype(np.int8)
    th = rng.normal(scale=0.01, size=(nn, n))
    nodes = np.arange(nn) * 2
    O.objgrad_nodes("RISE", None, spins[:2000], nodes[:32], th[:32])
    t = time.time(); O.objgrad_nodes("RISE", None, spins, nodes, th); dt = time.time() - t
    print(json.dumps({"threads": int(os.environ.get("OMP_NUM_THREADS", "0")), "s": dt, "gflops": 4.0 * K * n * nn / dt / 1e9}))
else:
    try:
        print(open("/sys/fs/cgroup/cpu.max").read().strip(), "| affinity", len(os.sched_getaffinity(0)))
    except Exception as e:
        print("cgroup:", e)
    for t in (8, 32, 64, 128, 256):
        env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND="spread", OMP_PLACES="cores" if t <= 128 else "threads")
        print(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
