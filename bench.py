#!/usr/bin/env python3
"""Benchmark of the learn() hot path on MI355X (BASELINE.json metric).

A "step" is one objective+gradient pass of the RISE operator over all n node-wise problems
(= n node evaluations: each node's f and grad over all K configurations) on the synthetic
BASELINE workload: n=1024 spins, K=1e6 samples (block-Ising, 64 blocks x 16 spins, seed 0).
Inputs (packed spins, weights, and the parameters Theta) are resident in HBM before the timed region;
the PCIe-inclusive rate through the host-pointer boundary is reported next to it.  With --gpus N the
nodes are sharded over N ranks (one process per GPU); no collective on the data path.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, algorithmic flops /
HIP-event time vs the MFMA peak of the arithmetic type), "cpu_baseline" (the CPU oracle timed on
this box's host cores on a bounded sample of the same workload), and the learn() wall-clock.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks (flop/s or op/s).  i8 / bf16: /opt/skills/guides/MI355X_MICROARCH.md (bf16
# ~2.5 PF dense, i8 = 2x bf16 per clock).  FP64: AMD datasheet (78.6 TF matrix = vector); the local
# guide lists no FP64 figure.
PEAKS = {"f64": 78.6e12, "i8x": 5.0e15}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--precision", default=os.environ.get("GML_BENCH_PRECISION", "i8x"), choices=["f64", "i8x"],
                    help="i8x: exact int8-limb MFMA pass (default, fastest); f64: FP64 MFMA pass")
    ap.add_argument("--no-learn", action="store_true", help="skip the full learn() wall-clock leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    device = local_rank

    import gml_amd as gml
    synthetic = __import__("importlib").import_module("gml_amd.synthetic")

    n, K = args.n, args.samples
    t0 = time.time()
    spins, J = synthetic.block_ising(n, K, block=args.block, seed=0)
    t_gen = time.time() - t0
    node0, node1 = (rank * n) // world, ((rank + 1) * n) // world
    nloc = node1 - node0

    t0 = time.time()
    prob = gml.Problem(spins=spins, node_range=(node0, node1), device=device)
    torch.cuda.synchronize()
    t_pack = time.time() - t0

    # evaluation point: the generating couplings (rows of the true model; diagonal = fields)
    theta = np.ascontiguousarray(J[node0:node1])

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Timed region: Theta resident in HBM (uploaded before the clock starts), K passes back to back on the library's
    # stream, no host round trip inside; f and the gradient stay in HBM until the last pass has finished.
    if args.warmup > 0:
        prob.bench_pass_resident("RISE", theta, steps=args.warmup, warmup=0, precision=args.precision)
    sync()
    t0 = time.perf_counter()
    km, f_res, g_res = prob.bench_pass_resident("RISE", theta, steps=args.steps, warmup=0, precision=args.precision, want_output=True)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n * args.steps / elapsed  # node evaluations per second, whole job
    # the same pass through the host-pointer operator boundary (Theta up and gradient down over PCIe every pass)
    t0 = time.perf_counter()
    prob.bench_pass("RISE", theta, steps=3, warmup=1, precision=args.precision)
    pcie_ms = (time.perf_counter() - t0) / 4 * 1e3

    # roofline of the dominant kernel (rank 0's shard): algorithmic flops = 2*K*P*n_loc per
    # kernel (forward energies or gradient accumulation; SURVEY.md 8(d): 4*K*P per node-eval)
    P = n
    flops_kernel = 2.0 * K * P * nloc
    dom = "bwd" if km["bwd_ms"] >= km["fwd_ms"] else "fwd"
    dom_ms = km[dom + "_ms"]
    achieved = flops_kernel / (dom_ms * 1e-3)
    peak = PEAKS[args.precision]
    roofline = {"bound": "mfma", "kernel": {"f64": {"fwd": "k_fwd_f64", "bwd": "k_bwd_f64"},
                                            "i8x": {"fwd": "k_fwd_i8", "bwd": "k_bwd_i8"}}[args.precision][dom],
                "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": None, "fwd_ms": km["fwd_ms"], "bwd_ms": km["bwd_ms"],
                "pass_tflops": 2 * flops_kernel / (km["pass_ms"] * 1e-3) / 1e12,
                # `value`: wall clock of K passes with Theta resident in HBM (one upload before, one download of f and
                # the gradient after the K passes, both inside the timed call).  For reference: the two GEMM kernels
                # alone, and one pass through the host-pointer boundary (Theta up, gradient down over PCIe per pass)
                "kernels_ms_per_step": km["pass_ms"], "device_ms_per_step": km["device_ms_per_pass"],
                "pcie_inclusive_ms_per_step": pcie_ms, "pcie_inclusive_node_evals_per_s_per_gpu": nloc / (pcie_ms * 1e-3)}
    if args.precision == "i8x":
        # the int8-limb pass issues LF forward + 4 backward digit-plane products per algorithmic one
        LF = int(os.environ.get("GML_I8_LF", "5"))
        limbs = {"fwd": LF, "bwd": 4}[dom]
        roofline["limb_products"] = limbs
        roofline["mfma_issue_frac"] = limbs * achieved / peak
        roofline["note"] = ("achieved counts algorithmic flops once; the kernel issues limb_products int8 MFMA products per "
                            "algorithmic product (exact fixed point), so frac <= 1/limb_products")

    # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process, so
    # the value measured with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on this same command
    # (profiles/r1_i8x_pmc_traffic.json, separate passes) is reported.  gfx950 correction per
    # MI355X_MICROARCH.md: FETCH_SIZE counts half the bytes of 16-B/lane loads -> doubled.
    try:
        if args.precision == "i8x" and (n, K, world) == (1024, 1000000, 1):
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_i8x_pmc_traffic.json")))
            kn = [k for k in pm if ("k_fwd_i8" in k if dom == "fwd" else "k_bwd_i8" in k)][0]
            roofline["traffic"] = (2.0 * pm[kn]["FETCH_SIZE_KB"] + pm[kn]["WRITE_SIZE_KB"]) * 1024.0
            roofline["traffic_note"] = ("bytes per launch from profiles/r1_i8x_pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, KB); "
                                        "algorithmic bytes per launch: %.3g (bit image K*n/8 + the 4 int8 limb planes of V, 4*K*n_loc)"
                                        % (K * n / 8.0 + 4.0 * K * nloc))
    except Exception:
        pass

    extra = {}
    if not args.no_learn:
        sync()
        t0 = time.perf_counter()
        out, kkt, st = prob.learn("RISE", 0.4, tol=1e-9, precision=args.precision, raise_on_fail=False)
        sync()
        t_learn = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([t_learn], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_learn = float(t.item())
            # final gather of the row blocks over RCCL/xGMI (the only collective of the path)
            tg = time.perf_counter()
            buf = torch.from_numpy(out).cuda()
            allb = torch.empty((world,) + tuple(buf.shape), dtype=buf.dtype, device="cuda")
            dist.all_gather_into_tensor(allb, buf)
            torch.cuda.synchronize()
            extra["gather_s"] = time.perf_counter() - tg
        sym_err = None
        if world == 1:
            sym_err = float(np.abs(0.5 * (out + out.T) - J).max())
        extra.update({"learn_wall_s": t_learn, "learn_pack_s": t_pack, "learn_iterations": st["iterations"],
                      "learn_passes": st["passes"], "learn_forward_passes": st["forward_passes"],
                      "learn_node_evals": st["node_evals"], "learn_max_kkt": st["max_kkt"],
                      "learn_not_converged": st["not_converged"], "learn_t_pass": st["t_pass"],
                      "learn_t_hess": st["t_hess"], "learn_t_host": st["t_host"], "max_err_vs_true_model": sym_err})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        # CPU baseline: the oracle's RISE objgrad (same math, FP64, C -O3, OpenMP over nodes) on this
        # box's host cores, on a bounded sample: `cores` node evaluations at full K and n per batch.
        from oracle import oracle as O
        cores = os.cpu_count() or 1
        counts = np.ones(K)
        nodes = np.arange(cores, dtype=np.int64) * (n // cores)
        O.objgrad_rise_nodes(counts[:1000], spins[:1000], nodes[:1], theta[nodes[:1]])  # load/compile
        done, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds and done < 4 * cores:
            t0 = time.perf_counter()
            f_cpu, g_cpu = O.objgrad_rise_nodes(counts, spins, nodes, theta[nodes])
            t_cpu += time.perf_counter() - t0
            done += len(nodes)
        # parity spot check of the timed GPU operator against the oracle on the same rows
        f_gpu, g_gpu = prob.objgrad("RISE", nodes, theta[nodes], precision=args.precision)
        assert np.array_equal(g_gpu, g_res[nodes]) and np.array_equal(f_gpu, f_res[nodes]), "timed passes != operator output"
        cpu = {"value": done / t_cpu, "unit": "node-evals/s", "cores": cores, "kind": "port",
               "sample": f"{done} node evaluations ({len(nodes)} nodes spread over 0..{n - 1}, one per core) at full K={K}, n={n}; oracle/gml_oracle.c "
                         f"gml_oracle_objgrad_rise_nodes, OpenMP over nodes",
               "seconds": t_cpu,
               "parity_max_abs_grad_diff": float(np.abs(g_gpu - g_cpu).max()),
               "parity_max_rel_f_diff": float(np.abs(f_gpu / f_cpu - 1).max())}

    if rank == 0:
        line = {"metric": "obj/grad evals/sec (RISE, n=%d spins, %d samples)" % (n, K), "value": value,
                "unit": "node-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": args.precision, "data": "synthetic",
                "config": {"workload": "BASELINE configs[2]-shaped RISE pass: n=%d random block-Ising (%d-spin blocks), "
                                       "%d samples, all n node-wise objective+gradient evaluations per step" % (n, args.block, K),
                           "n": n, "samples": K, "nodes_per_gpu": nloc, "parallelism": "node-shard x%d" % world,
                           "gen_s": t_gen, "pack_upload_s": t_pack},
                "roofline": roofline, "cpu_baseline": cpu}
        line.update(extra)
        print(json.dumps(line))
    prob.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
