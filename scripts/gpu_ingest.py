"""Ingest timings on the GPU box: learn() from a host Matrix{Int64}-shaped (column-major) histogram at the headline
config, both ingest routes, and gml_multi_create for C4 with 8 parts on device 0 (pack once, replicate).
Writes gpurun_out/ingest.json."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gml_amd as gml  # noqa: E402
from gml_amd import _lib, synthetic  # noqa: E402


def host_hist(spins, dtype=np.int64, order="F"):
    K, n = spins.shape
    h = np.empty((K, n + 1), dtype=dtype, order=order)
    h[:, 0] = 1
    for j0 in range(0, n, 64):
        h[:, 1 + j0:1 + j0 + 64] = spins[:, j0:j0 + 64]
    return h


def meminfo():
    d = {}
    for ln in open("/proc/meminfo"):
        k, v = ln.split(":")
        d[k] = int(v.split()[0]) / 1e6
    lim = None
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
    except Exception:
        pass
    return {"MemTotal_GB": d["MemTotal"], "MemAvailable_GB": d["MemAvailable"], "cgroup_memory_max": lim, "nproc": os.cpu_count()}


def main():
    res = {"host": meminfo()}
    print(res, flush=True)
    n, K = 1024, 1000000
    J = synthetic.block_ising_model(n, block=16, seed=0)
    with _lib.Problem(model=J, num_samples=K, seed=0) as p:
        spins = p.spins()
        t0 = time.perf_counter()
        p.learn("RISE", 0.4, tol=1e-9)
        t0 = time.perf_counter()
        _, _, st = p.learn("RISE", 0.4, tol=1e-9)
        res["solve_resident_s"] = time.perf_counter() - t0
    for dtype, order in ((np.int64, "F"), (np.float64, "F"), (np.int64, "C"), (np.int8, "C")):
        h = host_hist(spins, dtype, order)
        for ingest in ("host", "device"):
            if ingest == "device" and (dtype, order) != (np.int64, "F"):
                continue
            runs = []
            for rep in range(3):
                t0 = time.perf_counter()
                with _lib.Problem(h, ingest=ingest) as p:
                    t1 = time.perf_counter()
                    out, kkt, st = p.learn("RISE", 0.4, tol=1e-9)
                    t2 = time.perf_counter()
                    it = p.ingest_times()
                runs.append({"create_s": t1 - t0, "solve_s": t2 - t1, "total_s": t2 - t0, "ingest": it, "max_kkt": st["max_kkt"]})
            key = f"headline_{np.dtype(dtype).name}_{order}_{ingest}"
            res[key] = {"bytes": h.nbytes, "runs": runs, "best_total_s": min(r["total_s"] for r in runs)}
            print(key, res[key], flush=True)
        del h
    del spins
    # C4: n = 4096, K = 1e6, 8 parts on device 0 (one box): Matrix{Int64} = 32.8 GB on the host
    n4 = 4096
    avail = meminfo()["MemAvailable_GB"]
    K4 = 1000000 if avail > 48 else 250000
    J4 = synthetic.block_ising_model(n4, block=8, seed=0)
    with _lib.Problem(model=J4, num_samples=K4, seed=0, node_range=(0, 32)) as p:
        spins4 = p.spins()
    h4 = host_hist(spins4, np.int64, "F")
    del spins4
    runs = []
    for rep in range(2):
        t0 = time.perf_counter()
        m = _lib.MultiProblem(h4, [0] * 8)
        t1 = time.perf_counter()
        m.close()
        runs.append(t1 - t0)
    res["c4_multi_create_8_parts_device0"] = {"K": K4, "n": n4, "bytes": h4.nbytes, "create_s": runs}
    print(res["c4_multi_create_8_parts_device0"], flush=True)
    t0 = time.perf_counter()
    with _lib.Problem(h4, node_range=(0, 512)) as p:
        res["c4_single_part_create_s"] = time.perf_counter() - t0
        res["c4_single_part_ingest"] = p.ingest_times()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/ingest.json", "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
