"""Worker for the world_size-2 gloo tests of the node-sharded learn(): `oracle` mode injects the CPU oracle as the
per-rank solver (runs without a GPU); `hip` mode uses the product's HIP solver, every rank on a GPU (ranks share
device 0 on a 1-GPU box)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gml_amd as gml  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    out_dir, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "oracle")
    if mode == "oracle":  # the CPU oracle stands in for the device solver of every rank
        from test_host_api import inject_oracle
        inject_oracle()
        method = lambda: gml.HIP(distributed=True)  # noqa: E731
    else:
        import torch
        dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        method = lambda: gml.HIP(distributed=True, device=dev, tol=1e-11)  # noqa: E731
    # ONLY rank 0 holds sample matrices: it packs each once and broadcasts the bits (learn._packed_from_rank0); the other ranks
    # pass None and must never need more
    s = c = None
    if rank == 0:
        s = np.loadtxt(os.path.join(ROOT, "tests", "golden", "mvt_samples.csv"), delimiter=",")
        c = np.loadtxt(os.path.join(ROOT, "tests", "golden", "c_samples.csv"), delimiter=",")
    res = {}
    m = method()
    res["rise_sym"] = gml.learn(s, gml.RISE(0.2, True), m)
    res["start"] = np.array([m.stats["pack_s"], m.stats["bcast_s"], m.stats["bcast_bytes"]])
    res["rise"] = gml.learn(s, gml.RISE(0.2, False), method())
    fg = gml.learn(c, gml.multiRISE(0.2, True, 3), method())
    keys = sorted(fg.keys(), key=lambda k: (len(k), k))
    res["multi_vals"] = np.array([fg[k] for k in keys])
    if mode == "hip":
        # a problem wide enough that both ranks own several 32-node tiles, sampled identically by every rank
        synthetic = __import__("importlib").import_module("gml_amd.synthetic")
        hist = None
        if rank == 0:
            spins, _ = synthetic.block_ising(160, 20000, block=16, seed=12)
            hist = np.concatenate([np.ones((len(spins), 1)), spins.astype(np.float64)], axis=1)
        m = gml.HIP(distributed=True, device=dev, tol=1e-9, precision="i8x")
        res["wide"] = gml.learn(hist, gml.logRISE(0.8, False), m)
    # a matrix that cannot be packed fails on every rank (nobody is left waiting in a broadcast)
    bad = np.array([[1.0, 1.0, 2.0], [1.0, -1.0, 1.0]]) if rank == 0 else None
    try:
        gml.learn(bad, gml.RISE(), method())
        res["bad_raised"] = 0
    except gml.GMLError:
        res["bad_raised"] = 1
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), world=world, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
