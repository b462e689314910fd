"""The library keeps released device blocks for the next handle of the same shape (csrc/gml_alloc.cpp): a solve on recycled, dirty
memory must give the bits of a solve on fresh memory, and gml_trim_cache must hand everything back."""
import numpy as np
import pytest

import gml_amd as gml
from gml_amd import _lib

pytestmark = pytest.mark.gpu


def _hist(seed):
    rng = np.random.default_rng(seed)
    n, K = 40, 6000
    s = rng.choice([-1.0, 1.0], size=(K, n))
    s[:, 1] = np.where(rng.random(K) < 0.8, s[:, 0], -s[:, 0])
    return np.concatenate([rng.integers(1, 5, size=(K, 1)).astype(float), s], axis=1)


def test_recycled_blocks_give_the_same_solution_and_trim_releases_them():
    _lib.trim_cache()
    h1, h2 = _hist(1), _hist(2)

    def solve(h):
        with gml.Problem(h) as p:
            out, kkt, st = p.learn("logRISE", 0.3, tol=1e-10, precision="i8x")
            f, g = p.objgrad("RISE", np.arange(4), out[:4], precision="f64")
        return out, f, g

    fresh = solve(h2)            # every block new
    assert _lib.trim_cache() > 0  # ... and kept after the handle is gone, until trimmed
    assert _lib.trim_cache() == 0
    solve(h1)                    # leaves blocks full of another problem's data
    again = solve(h2)            # ... which this handle recycles
    assert np.array_equal(fresh[0], again[0])  # (the int8-limb passes are integer arithmetic: reproducible to the bit)
    for a, b in zip(fresh[1:], again[1:]):     # (the FP64 pass sums with atomics: to rounding)
        assert np.abs(a - b).max() <= 1e-13
    assert _lib.trim_cache() > 0


def test_cache_limit_zero_keeps_nothing_and_eviction_makes_room():
    _lib.trim_cache()
    try:
        _lib.set_cache_limit(0)
        with gml.Problem(_hist(3)) as p:
            p.learn("RISE", 0.4, tol=1e-9, precision="i8x")
        assert _lib.trim_cache() == 0  # every release went straight back to the driver
        _lib.set_cache_limit(3 << 20)  # room for one or two of the megabyte-sized blocks: the oldest are evicted, never more than the cap
        with gml.Problem(_hist(3)) as p:
            p.learn("RISE", 0.4, tol=1e-9, precision="i8x")
        assert 0 < _lib.trim_cache() <= (3 << 20)
    finally:
        _lib.set_cache_limit(-1)


def test_histogram_flag_validates_like_the_c_path():
    m = np.array([[0.1, 0.5, 0.0], [0.5, 0.0, -0.3], [0.0, -0.3, 0.2]])
    bad = m.copy()
    bad[0, 1] = 0.4
    with pytest.raises(gml.GMLError):
        gml.Problem(model=bad, num_samples=1000, histogram=True)
    with pytest.raises(gml.GMLError):
        gml.Problem(model=bad, num_samples=1000)
    with pytest.raises(gml.GMLError):
        gml.Problem(_hist(1), histogram=True)
    # same seed, same draws: the histogram of the device equals the histogram of the rows
    with gml.Problem(model=m, num_samples=5000, seed=3) as p:
        rows = p.spins()
    with gml.Problem(model=m, num_samples=5000, seed=3, histogram=True) as p:
        cfg, cnt = p.spins(), p.counts()
    u, c = np.unique(rows, axis=0, return_counts=True)
    got = {tuple(r): int(k) for r, k in zip(cfg, cnt)}
    assert got == {tuple(r): int(k) for r, k in zip(u, c)}
