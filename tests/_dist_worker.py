"""Worker for test_distributed_gloo.py: world_size-2 gloo run of the node-sharded learn()."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gml_amd as gml  # noqa: E402
from test_host_api import oracle_local_solve  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    out_dir = sys.argv[1]
    s = np.loadtxt(os.path.join(ROOT, "tests", "golden", "mvt_samples.csv"), delimiter=",")
    res = {}
    res["rise_sym"] = gml.learn(s, gml.RISE(0.2, True), gml.HIP(distributed=True), _local_solve=oracle_local_solve)
    res["rise"] = gml.learn(s, gml.RISE(0.2, False), gml.HIP(distributed=True), _local_solve=oracle_local_solve)
    c = np.loadtxt(os.path.join(ROOT, "tests", "golden", "c_samples.csv"), delimiter=",")
    fg = gml.learn(c, gml.multiRISE(0.2, True, 3), gml.HIP(distributed=True), _local_solve=oracle_local_solve)
    keys = sorted(fg.keys(), key=lambda k: (len(k), k))
    res["multi_vals"] = np.array([fg[k] for k in keys])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), world=world, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
