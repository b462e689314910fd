"""N>1 path on CPU: two gloo ranks shard the nodes, solve their ranges (oracle injected as the
per-rank solver -- no GPU here), all-gather the row blocks and assemble the result.  Checks the
partition + gather + symmetrise logic that runs over RCCL on the GPU box."""
import os
import socket
import subprocess
import sys

import numpy as np

import gml_amd as gml
from conftest import ROOT, load_csv
from test_host_api import oracle_local_solve


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_matches_single_process(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"),
           str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["world"]) == 2
    s = load_csv("mvt_samples.csv")
    single = gml.learn(s, gml.RISE(0.2, False), _local_solve=oracle_local_solve)
    single_sym = gml.learn(s, gml.RISE(0.2, True), _local_solve=oracle_local_solve)
    for rr in (r0, r1):  # every rank holds the full gathered result
        assert np.array_equal(rr["rise"], single)
        assert np.array_equal(rr["rise_sym"], single_sym)
    fg = gml.learn(load_csv("c_samples.csv"), gml.multiRISE(0.2, True, 3), _local_solve=oracle_local_solve)
    keys = sorted(fg.keys(), key=lambda k: (len(k), k))
    assert np.allclose(r0["multi_vals"], [fg[k] for k in keys], atol=1e-15)
    assert np.array_equal(r0["multi_vals"], r1["multi_vals"])
    assert np.abs(r0["rise"] - load_csv("mvt_RISE_learned.csv")).max() <= 3e-4
