"""Evidence files for the five BASELINE.json configs (last row): one JSON per config under
profiles/ with shapes, seed, lambda, iterations, evaluations, wall-clock split, evals/sec, algorithmic flops and
bytes, t_roof, roofline fraction, parity against the CPU oracle (max-abs / rel-Frobenius on objective+gradient of
sampled nodes, oracle-evaluated KKT residual of the learned rows), and the CPU side (core count, CPU model, CPU learn()
wall-clock with the same method and tolerance: measured where it finishes in the budget, else extrapolated from the
measured CPU objective/gradient rate and labelled so).

Run on the GPU box:  python scripts/gpu_configs.py [c1 c2 c3 c4 c5] [--round r2]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gml_amd as gml  # noqa: E402
from oracle import oracle as O  # noqa: E402

syn = __import__("importlib").import_module("gml_amd.synthetic")
PEAK = {"i8x": 5.0e15, "i8w": 5.0e15, "f64": 78.6e12}  # dense MFMA op/s (bench.py)
PREC = "i8x"   # --precision i8w: every device leg at the FP64-grade limbs (round 4: the arithmetic of the bench headline)
VPLANES = {"i8x": 4, "i8w": 6}
HBM = 8.0e12


def cpu_info():
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        model = "unknown"
    hc = O.host_cpus()
    return {"nproc": os.cpu_count(), "cpu_model": model, "cgroup_cpu_quota": hc["quota"], "oracle_threads": hc["threads"]}


def roof(K, P, nloc, pass_ms, prec=None, n_spins=None):
    """t_roof = max(bytes_alg / BW_peak, flops_alg / F_peak) with the ALGORITHMIC figures -- 4 K P flops per node
    evaluation; spins read once (bit-packed: K n / 8 B) + 8 K B of weights + 16 n_loc P B of Theta in / G out.  What THIS
    implementation moves on top of that (the int8 limb planes of V: written by the forward, read by the backward kernel) is
    reported apart as traffic_impl, never folded into the roofline."""
    prec = prec or PREC
    n_spins = n_spins or P
    flops = 4.0 * K * P * nloc
    bytes_alg = K * n_spins / 8.0 + 8.0 * K + 16.0 * nloc * P
    traffic_impl = 2 * K * P / 8.0 + 2.0 * VPLANES.get(prec, 4) * K * nloc  # both operand bit images + the V planes out and back in
    t_roof = max(bytes_alg / HBM, flops / PEAK[prec])
    return {"flops_alg_per_pass": flops, "bytes_alg_per_pass": bytes_alg, "traffic_impl_per_pass": traffic_impl, "t_roof_ms": t_roof * 1e3,
            "pass_ms": pass_ms, "roofline_frac": t_roof * 1e3 / pass_ms, "bound": "mfma" if flops / PEAK[prec] >= bytes_alg / HBM else "hbm",
            "achieved_TFLOPs": flops / (pass_ms * 1e-3) / 1e12, "alg_HBM_GBps": bytes_alg / (pass_ms * 1e-3) / 1e9,
            "impl_HBM_GBps": traffic_impl / (pass_ms * 1e-3) / 1e9}


def kkt_from_oracle(form, spins, rows, nodes, lam, counts=None):
    f, g = O.objgrad_nodes(form, counts, spins, np.asarray(nodes), rows)
    return max(O.kkt_residual(rows[a], g[a], lam, int(u)) for a, u in enumerate(nodes))


def pairwise(name, desc, J, K, form, c, seed, node_range=None, sample_nodes=4, cpu_learn="measure", tol=1e-9, hist=None):
    n = J.shape[0] if J is not None else hist.shape[1] - 1
    n0, n1 = node_range or (0, n)
    rec = {"config": desc, "n": n, "K": K, "formulation": f"{form}({c})", "seed": seed,
           # learn() at the library default: "auto" = the 38/31-bit int8 limbs, the FP64-grade ones for small problems (config 1)
           "precision": ("auto -> i8w" if K * n * n <= 2 ** 28 else ("auto -> i8x" if PREC == "i8x" else PREC)), "tol": tol,
           "node_range": [n0, n1], "n_gpus": 1, **cpu_info()}
    t0 = time.time()
    prob = gml.Problem(hist, node_range=node_range) if hist is not None else \
        gml.Problem(model=J, num_samples=K, seed=seed, node_range=node_range)
    rec["create_s"] = time.time() - t0
    with prob as p:
        K = p.K
        t0 = time.time()
        small = K * n * n <= 2 ** 28
        out, kkt, st = p.learn(form, c, tol=tol, raise_on_fail=False, precision="auto" if (small or PREC == "i8x") else PREC)
        rec["learn_s"] = time.time() - t0
        # a second solve on the warm handle (the first one of a handle also pays for the workspace allocation)
        t0 = time.time()
        p.learn(form, c, tol=tol, raise_on_fail=False, precision="auto" if (small or PREC == "i8x") else PREC)
        rec["learn_warm_s"] = time.time() - t0
        rec.update({"lambda": st["lambda_"], "iterations": st["iterations"], "passes": st["passes"],
                    "forward_passes": st["forward_passes"], "hessian_passes": st["hessian_passes"], "node_evals": st["node_evals"],
                    "max_kkt": st["max_kkt"], "not_converged": st["not_converged"], "polished": st["polished"],
                    "t_pass": st["t_pass"], "t_hess": st["t_hess"], "t_host": st["t_host"]})
        km = p.bench_pass_resident(form, out, steps=10, warmup=2, precision=PREC)
        km = {k: v for k, v in km.items() if k != "step_ms"}
        rec["pass"] = {**km, **roof(K, n, n1 - n0, km["device_ms_per_pass"])}
        rec["ingest"] = p.ingest_times()
        rec["node_evals_per_s"] = (n1 - n0) / (km["device_ms_per_pass"] * 1e-3)
        some = np.unique(np.linspace(n0, n1 - 1, sample_nodes).astype(np.int64))
        rng = np.random.default_rng(0)
        th = out[some - n0] + rng.normal(scale=0.02, size=(len(some), n)) * (rng.random((len(some), n)) < 0.05)
        f8, g8 = p.objgrad(form, some, th, precision=PREC)
        spins = p.spins()
    counts = None if hist is None else np.ascontiguousarray(hist[:, 0], dtype=np.float64)
    t0 = time.time()
    fo, go = O.objgrad_nodes(form, counts, spins, some, th)
    t_or = time.time() - t0
    rec["parity"] = {"nodes_checked": some.tolist(), "objgrad_max_abs_f": float(np.abs(f8 - fo).max()),
                     "objgrad_max_abs_g": float(np.abs(g8 - go).max()),
                     "objgrad_rel_frobenius_g": float(np.linalg.norm(g8 - go) / np.linalg.norm(go)),
                     "kkt_of_learned_rows_by_oracle": kkt_from_oracle(form, spins, out[some - n0], some, rec["lambda"], counts)}
    if J is not None:
        blk = out[:, n0:n1] if node_range else 0.5 * (out + out.T)
        rec["max_err_vs_generating_model"] = float(np.abs(blk - J[n0:n1, n0:n1]).max())
    # CPU side: objective/gradient rate of the blocked oracle, learn() with the same method and tolerance
    nn = min(n1 - n0, 256)
    cn = np.unique(np.linspace(n0, n1 - 1, nn).astype(np.int64))
    t0 = time.time()
    O.objgrad_nodes(form, counts, spins, cn, out[cn - n0])
    t_cpu = time.time() - t0
    rec["cpu"] = {"objgrad_node_evals_per_s": len(cn) / t_cpu, "objgrad_sample": f"{len(cn)} nodes at full K", "oracle_check_s": t_or}
    if cpu_learn == "measure":
        t0 = time.time()
        co, ck, cs = O.learn_pair_fast(counts, spins, form, c=c, node_range=(n0, n1), tol=tol)
        rec["cpu"].update({"learn_s": time.time() - t0, "learn_kind": "measured", "learn_passes": cs["passes"], "learn_node_evals": cs["node_evals"],
                           "learn_max_kkt": float(ck.max()), "max_abs_diff_cpu_vs_gpu": float(np.abs(co - out).max()),
                           "rel_frobenius_cpu_vs_gpu": float(np.linalg.norm(co - out) / np.linalg.norm(co))})
    else:
        rec["cpu"].update({"learn_s": rec["node_evals"] / rec["cpu"]["objgrad_node_evals_per_s"],
                           "learn_kind": "extrapolated: GPU node evaluations / measured CPU objective+gradient rate (Hessians, solves not counted)"})
    rec["speedup_learn_vs_cpu"] = rec["cpu"]["learn_s"] / rec["learn_s"]
    return name, rec


def c5(name, K=1000000, n=512, c=1.2, seed=5, tol=1e-8, max_iter=100):
    terms = syn.block_multibody_terms(n, block=16, seed=0)
    rec = {"config": "multi-body (3-spin) model, multiRISE/ISODUS order 3", "n": n, "K": K, "formulation": f"multiRISE({c}, true, 3)",
           "seed": seed, "precision": PREC, "tol": tol, "n_gpus": 1, **cpu_info()}
    t0 = time.time()
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=seed, order=3) as p:
        rec["create_s"] = time.time() - t0
        rec["P_per_node"] = P = p.P
        t0 = time.time()
        out, kkt, st = p.learn("RISE", c, tol=tol, precision=PREC, max_iter=max_iter, raise_on_fail=False)
        rec["learn_s"] = time.time() - t0
        rec.update({"lambda": st["lambda_"], "iterations": st["iterations"], "passes": st["passes"], "forward_passes": st["forward_passes"],
                    "hessian_and_hv_passes": st["hessian_passes"], "hv_node_evals": st["hv_evals"],
                    "node_evals": st["node_evals"], "max_kkt": st["max_kkt"], "not_converged": st["not_converged"],
                    "t_pass": st["t_pass"], "t_hess": st["t_hess"], "t_host": st["t_host"],
                    "nnz_per_node_max": int((out != 0).sum(1).max()), "nnz_per_node_mean": float((out != 0).sum(1).mean())})
        try:
            km = p.bench_pass_resident("RISE", out, steps=2, warmup=1, precision=PREC)
        except gml.GMLError:  # dense theta: some rows need the rescaled re-run, which the host-pointer pass performs
            km = p.bench_pass("RISE", out, steps=2, warmup=1, precision=PREC)
            km["device_ms_per_pass"] = km["pass_ms"]
            km["note"] = "kernel times of the first (bound-scaled) pass of gml_bench_pass; the rescaled re-run of some rows is extra"
        km = {k: v for k, v in km.items() if k != "step_ms"}
        rec["pass"] = {**km, **roof(K, P, n, km["device_ms_per_pass"], n_spins=n)}
        rec["node_evals_per_s"] = n / (km["device_ms_per_pass"] * 1e-3)
        some = np.array([0, n - 1])
        f8, g8 = p.objgrad("RISE", some, out[some], precision=PREC)
        keys0 = p.multi_keys(0)
        spins = p.spins()
    if "--front-door" in sys.argv:
        # learn(samples, ISODUS(c, true, 3), HIP()) as a user calls it: the K x (1+n) matrix in, a FactorGraph out (round 6: the
        # assembly of the 67 M solved parameters runs on the device, gml_learn_terms; the terms stay an array, factor_graph.TermArray)
        import threading

        import psutil
        hist = np.empty((K, n + 1), dtype=np.int8)
        hist[:, 0] = 1
        hist[:, 1:] = spins
        proc, stop = psutil.Process(), threading.Event()
        peak = [proc.memory_info().rss]
        base = peak[0]

        def watch():
            while not stop.wait(0.01):
                peak[0] = max(peak[0], proc.memory_info().rss)
        th = threading.Thread(target=watch, daemon=True)
        th.start()
        m = gml.HIP(precision=PREC, tol=tol, max_iter=max_iter)
        t0 = time.time()
        fg = gml.learn(hist, gml.ISODUS(c, True, 3), m)
        rec["learn_front_door_s"] = time.time() - t0
        stop.set()
        th.join()
        rec["front_door"] = {"solve_s": m.stats["t_total"] - m.stats["t_assemble"], "handle_from_matrix_s": m.stats["t_pack"],
                             "assemble_s": m.stats["t_assemble"], "terms": len(fg), "container": type(fg.terms).__name__,
                             "overhead_over_solve_s": rec["learn_front_door_s"] - (m.stats["t_total"] - m.stats["t_assemble"]),
                             "peak_host_rss_over_entry_GB": (peak[0] - base) / 1e9,
                             "max_err_vs_generating_terms": max(abs(fg[k] - v) for k, v in terms.items())}
        del hist, fg
    t0 = time.time()
    fo, go = O.objgrad_multi3_nodes(None, spins, some, out[some])
    t_cpu = time.time() - t0
    worst = 0.0
    for a in range(2):
        x, g = out[some[a]], go[a]
        pg = np.where(x > 0, g + rec["lambda"], np.where(x < 0, g - rec["lambda"], np.sign(g) * np.maximum(np.abs(g) - rec["lambda"], 0)))
        pg[0] = g[0]
        worst = max(worst, float(np.abs(pg).max()))
    rec["parity"] = {"nodes_checked": some.tolist(), "objgrad_max_rel_f": float(np.abs(f8 / fo - 1).max()),
                     "objgrad_max_abs_g": float(np.abs(g8 - go).max()),
                     "objgrad_rel_frobenius_g": float(np.linalg.norm(g8 - go) / np.linalg.norm(go)), "kkt_of_learned_rows_by_oracle": worst}
    rec["max_err_node0_vs_generating_terms"] = max(abs(v - terms.get(tuple(sorted(i + 1 for i in key)), 0.0)) for key, v in zip(keys0, out[0]))
    rate = 2 / t_cpu
    rec["cpu"] = {"objgrad_node_evals_per_s": rate, "objgrad_sample": "2 nodes at full K (OpenMP over samples)",
                  "learn_s": rec["node_evals"] / rate, "learn_kind": "extrapolated: GPU node evaluations / measured CPU objective+gradient rate"}
    rec["speedup_learn_vs_cpu"] = rec["cpu"]["learn_s"] / rec["learn_s"]
    return name, rec


def main():
    global PREC
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rnd = "r4"
    if "--round" in sys.argv:
        rnd = sys.argv[sys.argv.index("--round") + 1]
        args = [a for a in args if a != rnd]
    if "--precision" in sys.argv:
        PREC = sys.argv[sys.argv.index("--precision") + 1]
        args = [a for a in args if a != PREC]
    which = args or ["c1", "c2", "c3", "c4", "c5", "c5d"]
    out_dir = os.path.join(ROOT, "gpurun_out", "configs")
    os.makedirs(out_dir, exist_ok=True)
    jobs = []
    if "c1" in which:  # README 3-spin model (= test model a), the reference's own 1e6-sample histogram
        hist = np.loadtxt(os.path.join(ROOT, "tests", "golden", "a_samples.csv"), delimiter=",")
        jobs.append(lambda: pairwise("C1", "README 3-spin FactorGraph (test/data/a_samples.csv, M=1e6), RISE(0.4)", None, 8, "RISE", 0.4, 0,
                                     sample_nodes=3, hist=hist, tol=1e-11))
    if "c2" in which:
        jobs.append(lambda: pairwise("C2", "n=256 random (16-spin block) Ising, 1e5 samples, RISE(0.4)", syn.block_ising_model(256, 16, 0),
                                     100000, "RISE", 0.4, 0))
    if "c3" in which:
        jobs.append(lambda: pairwise("C3", "n=1024 random (16-spin block) Ising, 1e6 samples, logRISE(0.8)", syn.block_ising_model(1024, 16, 0),
                                     1000000, "logRISE", 0.8, 3, cpu_learn="measure" if "--cpu-full" in sys.argv else "extrapolate"))
        jobs.append(lambda: pairwise("C3_RISE", "n=1024 random (16-spin block) Ising, 1e6 samples, RISE(0.4) (the headline metric's problem)",
                                     syn.block_ising_model(1024, 16, 0), 1000000, "RISE", 0.4, 0,
                                     cpu_learn="measure" if "--cpu-full" in sys.argv else "extrapolate"))
    if "c4" in which:
        jobs.append(lambda: pairwise("C4_rank0of8", "n=4096 sparse (8-spin block) Ising, 1e6 samples, RISE(0.4): the shard of rank 0 of 8",
                                     syn.block_ising_model(4096, 8, 1), 1000000, "RISE", 0.4, 4, node_range=(0, 512),
                                     cpu_learn="measure" if "--cpu-full" in sys.argv else "extrapolate"))
    if "c4full" in which:  # the whole of config 4 on ONE GPU (what an 8-GPU run shards): all 4096 nodes, 1e6 samples
        jobs.append(lambda: pairwise("C4_full_1gpu", "n=4096 sparse (8-spin block) Ising, 1e6 samples, RISE(0.4): all nodes on one GPU",
                                     syn.block_ising_model(4096, 8, 1), 1000000, "RISE", 0.4, 4, cpu_learn="extrapolate"))
    if "c5" in which:
        jobs.append(lambda: c5("C5"))
    if "c5d" in which:  # the reference's default regulariser (multiRISE(0.4, true, 3) / ISODUS()): dense optimum, matrix-free Newton-CG
        jobs.append(lambda: c5("C5_default_c0.4", c=0.4))
    for job in jobs:
        name, rec = job()
        if name == "C1":
            G = np.loadtxt(os.path.join(ROOT, "tests", "golden", "a_RISE_learned.csv"), delimiter=",")
            rec["note"] = "golden a_RISE_learned.csv is reproduced by tests/test_gpu_parity.py::test_learn_abc_goldens (<= 5e-8)"
            rec["golden_shape"] = list(G.shape)
        path = os.path.join(out_dir, f"{rnd}_{name}" + ("" if PREC == "i8x" else "_" + PREC) + ".json")
        json.dump(rec, open(path, "w"), indent=1)
        print(name, json.dumps({k: rec[k] for k in ("learn_s", "node_evals_per_s", "speedup_learn_vs_cpu", "parity") if k in rec}), flush=True)


if __name__ == "__main__":
    main()
