"""Ingest on the GPU box: the host packer route of gml_problem_create (pack on the host, upload the bits), the
device-convert route (raw upload, convert on the device) and the packed form give the SAME resident sign bits, bit for
bit, for every element type and layout the reference can hand over (Matrix{Int64} from sample(), sampling.jl:52-54;
Float64 from readdlm, test/runtests.jl:71; column-major as Julia stores them, row-major as numpy does)."""
import numpy as np
import pytest

import gml_amd as gml
from gml_amd import _lib
from test_host_pack import make_hist, numpy_pack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.int64, np.float64])
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("K,n", [(8, 3), (1000, 70), (70001, 37)])
def test_host_pack_and_device_convert_give_the_same_bits(dtype, order, K, n):
    h, spins, counts = make_hist(K, n, dtype, order, seed=K)
    want = numpy_pack(spins)
    with _lib.Problem(h) as p, _lib.Problem(h, ingest="device") as q:
        a, b = p.sign_bits(), q.sign_bits()
        assert np.array_equal(a, want)
        assert np.array_equal(b, want)
        assert p.M == q.M == counts.sum()
        assert np.array_equal(p.spins(), spins)
        t = p.ingest_times()
        assert t["total_s"] > 0 and t["pack_s"] > 0


def test_large_column_major_int64_histogram_bits_and_learn():
    """a Matrix{Int64}-shaped input big enough for several pinned stages (n * Kp / 8 = 40 MB of bits)"""
    K, n = 2_500_000, 128
    rng = np.random.default_rng(1)
    h = np.empty((K, n + 1), dtype=np.int64, order="F")
    h[:, 0] = 1
    h[:, 1:] = rng.integers(0, 2, size=(K, n), dtype=np.int8) * 2 - 1
    with _lib.Problem(h, node_range=(0, 32)) as p, _lib.Problem(h, ingest="device", node_range=(0, 32)) as q:
        a = p.sign_bits()
        assert np.array_equal(a, q.sign_bits())
        assert np.array_equal(a, numpy_pack(h[:, 1:]))
        x, kkt, _ = p.learn("RISE", 0.4, tol=1e-9)
        y, _, _ = q.learn("RISE", 0.4, tol=1e-9)
        assert np.array_equal(x, y)  # same bits, same arithmetic: identical learned rows


def test_packed_form_round_trip(golden):
    s = golden("mvt_samples.csv")
    bits, counts, M = _lib.pack_histogram(s)
    with _lib.Problem(s) as p, _lib.Problem(packed=(bits, counts, s.shape[0])) as q:
        assert q.M == M == p.M
        assert np.array_equal(q.sign_bits(), p.sign_bits())
        a, _, _ = p.learn("RISE", 0.2, tol=1e-10, precision="f64")
        b, _, _ = q.learn("RISE", 0.2, tol=1e-10, precision="f64")
        assert np.array_equal(a, b)
    # garbage beyond K in the caller's last word must not leak into the padding configurations
    dirty = bits.copy()
    K = s.shape[0]
    if K % 32:
        dirty[:, K // 32] |= np.uint32(0xFFFFFFFF) << np.uint32(K % 32)
    with _lib.Problem(packed=(dirty, counts, K)) as q:
        assert np.array_equal(q.sign_bits(), bits)


def test_sampled_handle_to_packed_handles_on_other_node_ranges():
    """pack once, replicate: the bits of a handle sampled on the device build the handles of other node ranges"""
    from gml_amd import synthetic
    J = synthetic.block_ising_model(64, block=8, seed=3)
    with _lib.Problem(model=J, num_samples=50000, seed=1) as p:
        bits = p.sign_bits()
        full, _, _ = p.learn("RISE", 0.4, tol=1e-9)
    parts = []
    for n0, n1 in ((0, 32), (32, 64)):
        with _lib.Problem(packed=(bits, None, 50000), node_range=(n0, n1)) as q:
            parts.append(q.learn("RISE", 0.4, tol=1e-9)[0])
    assert np.abs(np.concatenate(parts) - full).max() < 1e-7


def test_bad_entries_are_reported_by_both_routes():
    h, _, _ = make_hist(3000, 20, np.int64, "F")
    h[1234, 7] = 0
    for ingest in ("host", "device"):
        with pytest.raises(gml.GMLError, match="configuration 1234 holds a spin that is not"):
            _lib.Problem(h, ingest=ingest)
    h, _, _ = make_hist(3000, 20, np.float64, "C")
    h[77, 0] = -3
    for ingest in ("host", "device"):
        with pytest.raises(gml.GMLError, match="count of configuration 77"):
            _lib.Problem(h, ingest=ingest)


def test_multi_create_packs_once_and_matches_single_device(golden):
    s = golden("c_samples.csv")
    want = golden("c_RISE_learned.csv")
    with _lib.MultiProblem(np.asfortranarray(s.astype(np.int64)), [0, 0]) as m:
        out, kkt, st = m.learn("RISE", 0.4, tol=1e-10)
    got = 0.5 * (out + out.T)
    assert np.abs(got - want).max() < 5e-8
