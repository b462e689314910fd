"""N>1 path on CPU: two gloo ranks shard the nodes, solve their ranges (oracle injected as the
per-rank solver -- no GPU here), all-gather the row blocks and assemble the result.  Checks the
partition + gather + symmetrise logic that runs over RCCL on the GPU box."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import gml_amd as gml
from conftest import ROOT, load_csv
from test_host_api import oracle_learn


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
    single = oracle_learn(s, gml.RISE(0.2, False))
    single_sym = oracle_learn(s, gml.RISE(0.2, True))
    for rr in (r0, r1):  # every rank holds the full gathered result
        assert np.array_equal(rr["rise"], single)
        assert np.array_equal(rr["rise_sym"], single_sym)
    fg = oracle_learn(load_csv("c_samples.csv"), gml.multiRISE(0.2, True, 3))
    keys = sorted(fg.keys(), key=lambda k: (len(k), k))
    assert np.allclose(r0["multi_vals"], [fg[k] for k in keys], atol=1e-15)
    assert np.array_equal(r0["multi_vals"], r1["multi_vals"])
    assert np.abs(r0["rise"] - load_csv("mvt_RISE_learned.csv")).max() <= 3e-4
    # the start of the run: rank 0 packed the matrix (rank 1 was handed None), both took part in one broadcast of the bits --
    # 9 spins x 32 words of 32 configurations + the 512 counts -- and a matrix that cannot be packed raised on both
    wpr = 1024 // 32
    for rr in (r0, r1):
        assert rr["start"][2] == 9 * wpr * 4 + 8 * 512 and rr["start"][1] > 0 and int(rr["bad_raised"]) == 1
    assert r1["start"][0] < r0["start"][0]  # (rank 1's "packing" is the empty branch)


def test_bench_launches_its_own_ranks(tmp_path):
    # `python bench.py --gpus N` with no WORLD_SIZE starts N ranks itself (torch.distributed.run as a child, before
    # anything touches a GPU) and relays rank 0's line; --dry-run keeps the rendezvous and the node partition only
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["partition"] == [[0, 512], [512, 1024]]


@pytest.mark.gpu
def test_two_rank_hip_solver_matches_single_process(tmp_path):
    # the sharded HIP path under real ranks: two processes, disjoint node ranges, the product's solver on the GPU
    # (both ranks on device 0 of a 1-GPU box), rows gathered through torch.distributed
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"),
           str(tmp_path), "hip"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["world"]) == 2
    s = load_csv("mvt_samples.csv")
    single = gml.learn(s, gml.RISE(0.2, False), gml.HIP(tol=1e-11))
    single_sym = gml.learn(s, gml.RISE(0.2, True), gml.HIP(tol=1e-11))
    for rr in (r0, r1):
        assert np.abs(rr["rise"] - single).max() <= 1e-9
        assert np.abs(rr["rise_sym"] - single_sym).max() <= 1e-9
    assert np.array_equal(r0["rise"], r1["rise"]) and np.array_equal(r0["multi_vals"], r1["multi_vals"])
    fg = gml.learn(load_csv("c_samples.csv"), gml.multiRISE(0.2, True, 3), gml.HIP(tol=1e-11))
    keys = sorted(fg.keys(), key=lambda k: (len(k), k))
    assert np.abs(r0["multi_vals"] - np.array([fg[k] for k in keys])).max() <= 1e-9
    synthetic = __import__("importlib").import_module("gml_amd.synthetic")
    spins, _ = synthetic.block_ising(160, 20000, block=16, seed=12)
    hist = np.concatenate([np.ones((len(spins), 1)), spins.astype(np.float64)], axis=1)
    wide = gml.learn(hist, gml.logRISE(0.8, False), gml.HIP(tol=1e-9, precision="i8x"))
    assert r0["wide"].shape == (160, 160) and np.array_equal(r0["wide"], r1["wide"])
    assert np.abs(r0["wide"] - wide).max() <= 2e-8  # same optimum from rows [0,80) + [80,160) as from one handle


@pytest.mark.gpu
def test_bench_two_ranks_on_the_gpu_box():
    # bench.py --gpus 2 end to end on the 1-GPU box: its own launcher, both ranks on device 0, node-sharded passes,
    # learn() and the gather; a small problem so that it takes seconds
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--spins", "256", "--samples", "50000", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["nodes_per_gpu"] == 128 and len(line["per_rank_ms_per_step"]) == 2
    assert line["value"] > 0 and line["learn_not_converged"] == 0 and line["max_err_vs_true_model"] < 0.1
    assert line["collective_ranks"] == 2 and "gather_s" in line
    # the front door of the process-per-GPU route from a host matrix only rank 0 holds: packed once, the bits broadcast
    fd = line["learn_distributed_front_door"]
    assert fd["same_rows_as_device_sampled_handles"] and len(fd["pack_s"]) == 2 and fd["pack_s"][1] < fd["pack_s"][0]
    assert fd["bcast_bytes"] == 256 * (50176 // 32) * 4 and all(v > 0 for v in fd["bcast_s"])
    # the headline is measured in arithmetic as wide as the reference's Float64 (the int8 limbs at 54 / 47 bits), the other two
    # arithmetics ride along, and a first multi-GPU run is diagnosable from the line alone
    assert line["dtype"] == "i8w" and line["learn_precision"] == "i8w"
    assert line["f64"]["max_abs_grad_diff_vs_headline"] <= 1e-12 and line["f64"]["max_rel_f_diff_vs_headline"] <= 1e-12
    assert line["value_f64"] > 0 and line["value_i8x"] > 0 and line["learn_wall_s_f64"] > 0 and line["learn_wall_s_i8x"] > 0
    for key in ("collective_backend", "devices_visible", "learn_per_rank_s", "learn_per_rank_iterations", "learn_per_rank_passes",
                "learn_per_rank_node_evals", "learn_per_rank_t_pass_s", "learn_per_rank_t_hess_s"):
        assert key in line, key
    assert len(line["learn_per_rank_iterations"]) == 2
