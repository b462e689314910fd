"""The closed-form key order of the array-backed multi-body result (gml_terms_count / _keys / _rank: host only) against the
reference's own construction -- `permutations` (models.jl:228-246), the per-node key lists (:94-104) and the listing order of
show / jsondata (models.jl:61,72) -- restated in the oracle; the TermArray / FactorGraph container protocol on top of it
(models.jl:79-85); and the ABI identity check of the bindings.  No GPU needed."""
import math

import numpy as np
import pytest

import gml_amd as gml
from oracle import oracle as O

_lib = gml._lib
TermArray = __import__("importlib").import_module("gml_amd.factor_graph").TermArray


def reference_keys(n, order, symmetrize):
    """the keys of learn(samples, multiRISE(., symmetrize, order)) in listing order, 1-based (oracle restatement of :94-104,:135-149)"""
    rec = {}
    for u in range(n):
        for k in O.multi_keys(n, order, u):
            rec[tuple(i + 1 for i in k)] = 0.0
    if symmetrize:
        rec = {tuple(sorted(k)): 0.0 for k in rec}
    return O.listing_order(rec)


@pytest.mark.parametrize("n,order", [(1, 1), (3, 1), (2, 2), (5, 2), (3, 3), (7, 3), (9, 4), (6, 6), (8, 5)])
@pytest.mark.parametrize("sym", [True, False])
def test_term_positions_follow_the_reference_listing_order(n, order, sym):
    want = reference_keys(n, order, sym)
    T = _lib.terms_count(n, order, sym)
    assert T == len(want)
    if sym:
        assert T == sum(math.comb(n, s) for s in range(1, order + 1))
    else:
        assert T == n * sum(math.comb(n - 1, s - 1) for s in range(1, order + 1))
    keys = _lib.terms_keys(n, order, sym)
    got = [tuple(int(v) + 1 for v in row if v >= 0) for row in keys]
    assert got == want
    for t, k in enumerate(want):  # rank is the inverse of keys
        assert _lib.terms_rank(n, order, sym, [i - 1 for i in k]) == t
    # any window of the key table is the same as the slice of the whole (chunked generation restarts by unranking)
    rng = np.random.default_rng(n * 10 + order)
    for _ in range(5):
        a = int(rng.integers(0, T))
        c = int(rng.integers(0, T - a + 1))
        assert np.array_equal(_lib.terms_keys(n, order, sym, a, c), keys[a:a + c])


def test_term_positions_at_config5_scale():
    # n = 512, order 3: 22.5 M symmetrised terms, 67.0 M unsymmetrised; spot checks by closed form, windows across size borders
    n, order = 512, 3
    T = _lib.terms_count(n, order, True)
    assert T == 512 + 512 * 511 // 2 + 512 * 511 * 510 // 6 == 22370048
    assert _lib.terms_count(n, order, False) == 512 * (1 + 511 + 511 * 510 // 2)
    k = _lib.terms_keys(n, order, True, n + 512 * 511 // 2 - 2, 4)
    assert k.tolist() == [[509, 511, -1], [510, 511, -1], [0, 1, 2], [0, 1, 3]]
    assert _lib.terms_keys(n, order, True, T - 1, 1).tolist() == [[509, 510, 511]]
    rng = np.random.default_rng(0)
    for t in rng.integers(0, T, 200):
        key = _lib.terms_keys(n, order, True, int(t), 1)[0]
        assert _lib.terms_rank(n, order, True, key[key >= 0]) == t
    Tu = _lib.terms_count(n, order, False)
    for t in rng.integers(0, Tu, 200):
        key = _lib.terms_keys(n, order, False, int(t), 1)[0]
        assert _lib.terms_rank(n, order, False, key[key >= 0]) == t
    # a 3 M window crossing many chunk borders is strictly increasing in (length, key)
    w = _lib.terms_keys(n, order, True, 130000, 3000000).astype(np.int64)
    code = ((w >= 0).sum(1) << 40) + ((w[:, 0] + 1) << 24) + ((w[:, 1] + 1) << 12) + (w[:, 2] + 1)
    assert (np.diff(code) > 0).all()


def test_rank_rejects_what_is_not_a_key():
    assert _lib.terms_rank(5, 3, True, [1, 1]) == -1       # not strictly ascending
    assert _lib.terms_rank(5, 3, True, [2, 1]) == -1
    assert _lib.terms_rank(5, 3, True, [0, 5]) == -1       # spin out of range
    assert _lib.terms_rank(5, 3, True, [0, 1, 2, 3]) == -1  # longer than the order
    assert _lib.terms_rank(5, 3, False, [2, 2]) == -1      # (u, u)
    assert _lib.terms_rank(5, 3, False, [2, 4, 1]) == -1   # others not ascending
    assert _lib.terms_rank(5, 3, False, [2, 1, 4]) >= 0
    with pytest.raises(gml.GMLError, match="orders above 8"):  # refused with a message, never a wrong number
        _lib.terms_count(5, 9, True)
    with pytest.raises(gml.GMLError, match="1e12"):
        _lib.terms_count(100000, 8, True)                  # more terms than any memory holds


def test_term_array_is_the_reference_container():
    n, order = 6, 3
    for sym in (True, False):
        want = reference_keys(n, order, sym)
        w = np.random.default_rng(3).normal(size=len(want))
        ta = TermArray(n, order, sym, w)
        ref = dict(zip(want, w.tolist()))
        assert len(ta) == len(ref) and list(ta) == want and list(ta.keys()) == want
        assert dict(ta.items()) == ref and ta.to_dict() == ref and ta == ref
        assert ta.values() is ta.weights
        for k in want[::7]:
            assert ta[k] == ref[k] and k in ta
        for bad in [(0,), (7,), (1, 1), (1, 2, 3, 4), "x"]:
            assert bad not in ta
        with pytest.raises(KeyError):
            ta[(3, 3)]
        fg = gml.FactorGraph(order, n, "spin", ta)
        assert len(fg) == len(ref) and fg[want[5]] == ref[want[5]] and list(fg.keys()) == want
        assert dict(fg) == ref                                         # iteration yields (key, weight) pairs: models.jl:79
        assert fg.jsondata() == gml.FactorGraph(order, n, "spin", ref).jsondata()
        assert str(fg) == str(gml.FactorGraph(order, n, "spin", ref))
        ka = ta.keys_array(3, 4)
        assert ka.dtype == np.int32 and [tuple(int(v) for v in r if v > 0) for r in ka] == want[3:7]
    with pytest.raises(ValueError):
        TermArray(n, order, True, np.zeros(5))
    with pytest.raises(ValueError):
        gml.FactorGraph(2, n, "spin", TermArray(n, 3, True, np.zeros(_lib.terms_count(n, 3, True))))


def test_term_array_pairwise_to_matrix():
    n = 7
    want = reference_keys(n, 2, True)
    w = np.random.default_rng(4).normal(size=len(want))
    fg = gml.FactorGraph(2, n, "spin", TermArray(n, 2, True, w))
    assert np.array_equal(fg.to_matrix(), gml.FactorGraph(2, n, "spin", dict(zip(want, w.tolist()))).to_matrix())


def test_bindings_refuse_a_library_of_another_abi(monkeypatch):
    L = _lib.lib()
    assert L.gml_abi_version() == _lib.GML_ABI_VERSION
    _lib._check_abi(L)  # the loaded one is accepted
    monkeypatch.setattr(_lib, "GML_ABI_VERSION", _lib.GML_ABI_VERSION + 1)
    with pytest.raises(gml.GMLError, match="different revisions"):
        _lib._check_abi(L)
    monkeypatch.undo()

    class Grown(_lib.Stats):  # a binding whose mirror has a field the library does not know
        _fields_ = [("extra", _lib.C.c_double)]
    monkeypatch.setattr(_lib, "Stats", Grown)
    with pytest.raises(gml.GMLError, match="sizeof"):
        _lib._check_abi(L)
