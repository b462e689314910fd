"""gml_multi_*: one problem over several GPUs from one process (the boundary a Julia caller binds).  The test box has one
GPU, so the device list repeats device 0 -- the code path (one handle + one host thread per part, disjoint node ranges,
rows gathered into the caller's matrix, device-resident gather) is the same; with a single part the device-resident
gather goes through RCCL's all-gather (one rank), with a repeated device through the peer-copy fallback."""
import numpy as np
import pytest

import gml_amd as gml
from conftest import load_csv

pytestmark = pytest.mark.gpu
synthetic = __import__("importlib").import_module("gml_amd.synthetic")


def _hist(spins):
    return np.concatenate([np.ones((spins.shape[0], 1)), spins.astype(np.float64)], axis=1)


def test_multi_matches_single_handle():
    spins, J = synthetic.block_ising(96, 20000, block=16, seed=21)
    hist = _hist(spins)
    with gml.Problem(hist) as p:
        ref, kref, sref = p.learn("RISE", 0.4, tol=1e-10)
    with gml.MultiProblem(hist, [0, 0, 0]) as m:
        assert (m.n, m.K, m.P, m.ndev) == (96, 20000, 96, 3)
        out, kkt, st = m.learn("RISE", 0.4, tol=1e-10)
        parts = m.part_stats()  # per node shard: what a scaling run reads to see stragglers
    assert len(parts) == 3 and sum(q["node_evals"] for q in parts) == st["node_evals"]
    assert max(q["t_total"] for q in parts) == st["t_total"] and all(q["iterations"] > 0 and q["t_pack"] > 0 for q in parts)
    assert st["not_converged"] == 0 and kkt.max() <= 1e-10
    assert np.abs(out - ref).max() <= 2e-9  # same optimum (the parts' Newton trajectories differ: adaptive Hessian budget)
    assert st["node_evals"] > 0 and st["passes"] >= sref["passes"]


def test_multi_through_learn_and_goldens():
    s = load_csv("mvt_samples.csv")
    R = gml.learn(s, gml.RISE(0.2, False), gml.HIP(tol=1e-11, devices=[0, 0]))
    R1 = gml.learn(s, gml.RISE(0.2, False), gml.HIP(tol=1e-11))
    assert np.abs(R - R1).max() <= 1e-9 and np.abs(R - load_csv("mvt_RISE_learned.csv")).max() <= 3e-4
    c = load_csv("c_samples.csv")
    fg = gml.learn(c, gml.multiRISE(0.2, True, 3), gml.HIP(tol=1e-11, devices=[0, 0]))
    fg1 = gml.learn(c, gml.multiRISE(0.2, True, 3), gml.HIP(tol=1e-11))
    assert set(fg.keys()) == set(fg1.keys()) and max(abs(fg[k] - fg1[k]) for k in fg1.keys()) <= 1e-9


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_multi_device_resident_gather(devices):
    import torch
    spins, J = synthetic.block_ising(64, 8000, block=16, seed=22)
    hist = _hist(spins)
    bufs = [torch.full((64, 64), float("nan"), dtype=torch.float64, device="cuda:0") for _ in devices]
    with gml.MultiProblem(hist, devices) as m:
        out, kkt, st = m.learn("logRISE", 0.8, tol=1e-9, dev_out=[b.data_ptr() for b in bufs])
        kind = m.gather_kind()
    torch.cuda.synchronize()
    assert kind == ("rccl-allgather" if len(devices) == 1 else "peer-copy")
    for b in bufs:  # every part holds the full matrix
        assert np.array_equal(b.cpu().numpy(), out)
    assert st["not_converged"] == 0


def test_multi_rejects_bad_device_lists():
    hist = _hist(synthetic.block_ising(16, 500, block=16, seed=1)[0])
    with pytest.raises(gml.GMLError):
        gml.MultiProblem(hist, [])
    with pytest.raises(gml.GMLError):
        gml.MultiProblem(hist, [0, 99])


def test_multi_collective_setup_is_diagnosable_and_fails_cleanly():
    # What can be checked of the RCCL path on one GPU: (1) a single part builds a real communicator and says so; (2) a device
    # list that repeats a GPU is told apart and falls back to peer copies; (3) forcing ncclCommInitAll on the repeated list --
    # which RCCL refuses -- comes back as GML_EHIP carrying ncclGetErrorString's text, not as a crash or a hang, and the handle
    # keeps working afterwards.  (Several DISTINCT devices run for the first time on the driver's 8-GPU box.)
    from gml_amd import _lib
    import ctypes as C
    spins, J = synthetic.block_ising(32, 4000, block=16, seed=23)
    hist = _hist(spins)
    with gml.MultiProblem(hist, [0]) as m:
        assert "RCCL communicators ready: 1 ranks" in m.diag()
    with gml.MultiProblem(hist, [0, 0]) as m:
        assert "repeats a GPU" in m.diag() and "peer copies" in m.diag()
        L = _lib.lib()
        L.gml_test_multi_force_rccl.argtypes = [C.c_void_p]
        rc = L.gml_test_multi_force_rccl(m._h)
        if rc != 0:
            assert rc == _lib.GML_EHIP
            msg = L.gml_last_error().decode()
            assert "ncclCommInitAll" in msg and len(msg) > len("ncclCommInitAll over 2 ranks failed: ")
            assert "ncclCommInitAll failed" in m.diag()
        out, kkt, st = m.learn("RISE", 0.4, tol=1e-9)  # the handle still solves (host gather)
        assert st["not_converged"] == 0
