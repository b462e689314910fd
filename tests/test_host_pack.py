"""The host-side histogram packer (csrc/gml_pack.cpp, gml_pack_histogram): the ingest step of gml_problem_create.
Pure host code -- no GPU needed.  Checked against a numpy statement of the packed form: sign word w of spin i holds
configuration 32 w + j at bit j, set <=> the spin is -1 (the K x (1+n) layout of sampling.jl:52-54, column 0 = counts)."""
import numpy as np
import pytest

import gml_amd as gml
from gml_amd import _lib


def numpy_pack(spins):
    """spins K x n in {-1,+1} -> [n][round_up(K,1024)/32] uint32"""
    K, n = spins.shape
    Kp = (K + 1023) // 1024 * 1024
    neg = np.zeros((n, Kp), dtype=np.uint8)
    neg[:, :K] = (spins.T < 0)
    return np.packbits(neg.reshape(n, Kp // 32, 32), axis=2, bitorder="little").view(np.uint32).reshape(n, Kp // 32)


def make_hist(K, n, dtype, order, seed=0):
    rng = np.random.default_rng(seed)
    spins = rng.choice(np.array([-1, 1]), size=(K, n))
    counts = rng.integers(0, 50, size=K)
    counts[0] = 7
    h = np.concatenate([counts[:, None], spins], axis=1).astype(dtype)
    return np.asarray(h, order=order), spins, counts


@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.int64, np.float64])
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("K,n", [(1, 1), (31, 3), (32, 33), (1000, 70), (4097, 64), (70001, 37)])
def test_pack_matches_numpy(dtype, order, K, n):
    h, spins, counts = make_hist(K, n, dtype, order, seed=K + n)
    bits, cnt, M = _lib.pack_histogram(h)
    assert bits.shape == (n, (K + 1023) // 1024 * 32)
    assert np.array_equal(bits, numpy_pack(spins))
    assert np.array_equal(cnt, counts.astype(np.float64))
    assert M == counts.sum()


def test_pack_with_leading_dimension_and_views():
    """a sub-block of a larger matrix (ld > K / ld > n+1), as a Julia `view` or a numpy slice would hand over"""
    import ctypes as C
    L = _lib.lib()
    K, n, ldc, ldr = 777, 45, 900, 60
    rng = np.random.default_rng(5)
    big_c = np.asfortranarray(rng.choice(np.array([-1, 1], dtype=np.int64), size=(ldc, n + 1)))
    big_c[:, 0] = rng.integers(1, 9, size=ldc)
    big_r = np.ascontiguousarray(rng.choice(np.array([-1.0, 1.0]), size=(K, ldr)))
    big_r[:, 0] = rng.integers(1, 9, size=K)
    for arr, dt, ld, cm, spins, counts in ((big_c, 2, ldc, 1, big_c[:K, 1:], big_c[:K, 0]), (big_r, 3, ldr, 0, big_r[:, 1:n + 1], big_r[:, 0])):
        wpr = L.gml_packed_words(K)
        bits = np.empty((n, wpr), dtype=np.uint32)
        cnt = np.empty(K)
        M = C.c_double()
        _lib.check(L.gml_pack_histogram(_lib._ptr(arr), dt, K, n, ld, cm, _lib._ptr(bits), wpr, _lib._ptr(cnt), C.byref(M)))
        assert np.array_equal(bits, numpy_pack(spins))
        assert np.array_equal(cnt, counts.astype(np.float64))


@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.int64, np.float64])
@pytest.mark.parametrize("order", ["C", "F"])
def test_pack_rejects_what_is_not_plus_minus_one(dtype, order):
    K, n = 5000, 40
    for bad_value in ([0, 2, -2, 3] if dtype != np.float64 else [0.0, 0.5, -1.0000000000000002, np.nan, np.inf, 2.0]):
        h, _, _ = make_hist(K, n, dtype, order, seed=3)
        h[4321, 17] = bad_value
        h[4800, 2] = bad_value  # the message names the FIRST offending configuration
        with pytest.raises(gml.GMLError, match="configuration 4321 holds a spin that is not"):
            _lib.pack_histogram(h)


def test_pack_rejects_bad_counts():
    h, _, _ = make_hist(300, 5, np.float64, "F")
    for v in (-1.0, np.nan, np.inf):
        g = h.copy(order="F")
        g[123, 0] = v
        with pytest.raises(gml.GMLError, match="count of configuration 123"):
            _lib.pack_histogram(g)
    g = h.copy(order="F")
    g[:, 0] = 0
    with pytest.raises(gml.GMLError, match="sum of counts is zero"):
        _lib.pack_histogram(g)


def test_pack_golden_fixture():
    """the reference's own fixture (test/data/mvt_samples.csv: Float64 via readdlm): 512 configurations of 9 spins"""
    from conftest import load_csv
    s = load_csv("mvt_samples.csv")
    bits, cnt, M = _lib.pack_histogram(s)
    assert np.array_equal(bits, numpy_pack(s[:, 1:].astype(np.int64)))
    assert M == s[:, 0].sum()
