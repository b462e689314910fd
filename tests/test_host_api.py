"""Host-side mirror of the reference API: types/defaults (GraphicalModelLearning.jl:20-65), the
learn() front door (:69-73) and result assembly (:181-188, :129-151), exercised without a GPU by
injecting the CPU oracle as the per-rank solver (test infrastructure only)."""
import numpy as np
import pytest

import gml_amd as gml
from conftest import DEFAULT_C, MODELS, load_csv
from oracle import oracle as O

synthetic = __import__("importlib").import_module("gml_amd.synthetic")
learn_module = __import__("importlib").import_module("gml_amd.learn")


def unpack_histogram(packed):
    """(sign_bits [n][words] uint32, counts or None, K) -> K x (1+n) float64 histogram (bit set <=> spin -1; include/gml.h)"""
    bits, counts, K = packed
    n = bits.shape[0]
    sp = ((bits[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)[:, :K].T
    return np.concatenate([(np.ones(K) if counts is None else counts)[:, None], 1.0 - 2.0 * sp], axis=1)


def oracle_local_solve(samples, formulation, method, order, node_range, device, terms=None, packed=None, matrix=None):
    name = type(formulation).__name__
    assert terms is None and matrix is None  # (the fused solve + assembly calls are the product solver's own)
    if packed is not None:  # a rank of a distributed run: the bits rank 0 broadcast, no sample matrix
        assert samples is None
        samples = unpack_histogram(packed)
    n = samples.shape[1] - 1
    if name == "multiRISE":
        counts, spins = O.split_histogram(samples)
        P = O.lib().gml_oracle_multi_nparams(n, order)
        out = np.zeros((n, P))
        kkt = np.zeros(n)
        O.lib().gml_oracle_learn_multi(spins.shape[0], n, order, O._ptr(counts), O._ptr(spins),
                                       float(formulation.regularizer), 1e-12, O._ptr(out), O._ptr(kkt))
        return out[node_range[0]:node_range[1]], kkt, {}
    form = {"RISEA": "RISE"}.get(name, name)
    R, kkt, _ = O.learn_pair(samples, form, c=formulation.regularizer, symmetrize=False)
    return R[node_range[0]:node_range[1]], kkt[node_range[0]:node_range[1]], {}


def oracle_assemble_terms(rows, n, order, symmetrize, device):
    """the reference's dict assembly (:129-151, restated in the oracle) in place of the device kernel: the model's weights in
    (length, key) listing order -- checks the closed-form key order of the product's TermArray on the way"""
    rec = O.assemble_multi_dict(rows, [O.multi_keys(n, order, u) for u in range(n)], symmetrize)
    return np.array([rec[k] for k in O.listing_order(rec)])


def host_symmetrize(R, device):
    return 0.5 * (R + R.T)  # GraphicalModelLearning.jl:184-186 as the reference writes it


def inject_oracle():
    learn_module._local_solve_hip = oracle_local_solve
    learn_module._assemble_terms_hip = oracle_assemble_terms
    learn_module._symmetrize_hip = host_symmetrize


def oracle_learn(*args):
    """gml.learn with the CPU oracle standing in for the per-rank device solver and the device assembly (host-layer tests
    without a GPU)"""
    old = learn_module._local_solve_hip, learn_module._assemble_terms_hip, learn_module._symmetrize_hip
    inject_oracle()
    try:
        return gml.learn(*args)
    finally:
        learn_module._local_solve_hip, learn_module._assemble_terms_hip, learn_module._symmetrize_hip = old


def test_type_defaults_match_reference():
    assert (gml.RISE().regularizer, gml.RISE().symmetrization) == (0.4, True)          # :35
    assert (gml.RISEA().regularizer, gml.RISEA().symmetrization) == (0.4, True)        # :42
    assert (gml.logRISE().regularizer, gml.logRISE().symmetrization) == (0.8, True)    # :49
    assert (gml.RPLE().regularizer, gml.RPLE().symmetrization) == (0.2, True)          # :56
    m = gml.multiRISE()
    assert (m.regularizer, m.symmetrization, m.interaction_order) == (0.4, True, 2)    # :28
    assert gml.ISODUS is gml.multiRISE
    assert gml.RISE(0.2, False) == gml.RISE(regularizer=0.2, symmetrization=False)
    for T in (gml.RISE, gml.RISEA, gml.logRISE, gml.RPLE, gml.multiRISE):
        assert issubclass(T, gml.GMLFormulation)
    assert issubclass(gml.NLP, gml.GMLMethod) and issubclass(gml.HIP, gml.GMLMethod)
    # the device method's own defaults (include/gml.h: gml_default_opts)
    assert (gml.HIP().tol, gml.HIP().precision, gml.HIP().max_iter) == (1e-9, "auto", 100)


def test_learn_rejects_wrong_types():
    s = load_csv("a_samples.csv")
    with pytest.raises(TypeError):
        gml.learn(s, "RISE")
    with pytest.raises(TypeError):
        gml.learn(s, gml.RISE(), "ipopt")
    with pytest.raises(ValueError):
        gml.learn(np.zeros(5))


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("form", ["RISE", "logRISE", "RPLE"])
def test_learn_assembly_against_goldens(name, form):
    # runtests.jl:68-80 through the host layer (oracle injected as the node solver)
    F = getattr(gml, form)
    R = oracle_learn(load_csv(f"{name}_samples.csv"), F())
    G = load_csv(f"{name}_{form}_learned.csv")
    assert np.abs(R - G).max() <= 5e-8
    assert np.allclose(R, R.T)


def test_learn_default_arguments_are_rise():
    s = load_csv("a_samples.csv")
    R1 = oracle_learn(s)
    R2 = oracle_learn(s, gml.RISE(), gml.NLP())
    with pytest.warns(UserWarning, match="configured optimizer is not used"):  # a configured Ipopt is not silently replaced
        R3 = oracle_learn(s, gml.RISE(), gml.NLP(solver="Ipopt.Optimizer"))
    assert np.array_equal(R2, R3)
    assert np.array_equal(R1, R2)


def test_unsymmetrised_and_fortran_and_transposed_inputs():
    s = load_csv("mvt_samples.csv")
    G = load_csv("mvt_RISE_learned.csv")
    for arr in (s, np.asfortranarray(s), np.ascontiguousarray(s.T).T, s.astype(np.int64)):
        R = oracle_learn(arr, gml.RISE(0.2, False))
        assert np.abs(R - G).max() <= 3e-4
        assert not np.allclose(R, R.T)


def test_multirise_returns_factor_graph_and_matches_rise():
    # runtests.jl:132-146
    for name in "abc":
        s = load_csv(f"{name}_samples.csv")
        ising = oracle_learn(s, gml.RISE(0.2, False))
        two = oracle_learn(s, gml.multiRISE(0.2, False, 2))
        assert isinstance(two, gml.FactorGraph) and two.order == 2
        d = gml.matrix_to_terms(ising)
        assert len(d) == len(two)
        for k, v in d.items():
            assert two[k] == pytest.approx(v, abs=1e-7)


def test_multirise_symmetrisation_groups_sorted_keys():
    s = load_csv("c_samples.csv")
    fg = oracle_learn(s, gml.multiRISE(0.2, True, 3))
    raw = oracle_learn(s, gml.multiRISE(0.2, False, 3))
    assert all(tuple(sorted(k)) == k for k in fg.keys())
    assert fg[(1, 2, 3)] == pytest.approx(np.mean([raw[(1, 2, 3)], raw[(2, 1, 3)], raw[(3, 1, 2)]]))
    assert fg[(2,)] == pytest.approx(raw[(2,)])


def test_factor_graph_matrix_round_trip():
    # runtests.jl:17-30
    for name, m in MODELS.items():
        gm = gml.FactorGraph(m)
        m2 = gm.to_matrix()
        gm2 = gml.FactorGraph(m2)
        for key in gm.keys():
            assert gm[key] == pytest.approx(gm2[key])
            v = m2[key[0] - 1, key[0] - 1] if len(key) == 1 else m2[key[0] - 1, key[1] - 1]
            assert gm[key] == pytest.approx(v)
    assert [d["term"] for d in gml.FactorGraph(MODELS["a"]).jsondata()] == [[1, 2], [1, 3], [2, 3]]


def test_learned_model_accuracy_thresholds():
    # runtests.jl:105-127 with this repo's enumeration sampler (different RNG: thresholds transfer)
    for name, m in MODELS.items():
        for N, thr in ((1000, 0.15), (10000, 0.05)):
            hist = synthetic.enumerate_sample(m, N, seed=0)
            assert hist[:, 0].sum() == N
            for F in (gml.RISE, gml.logRISE, gml.RPLE):
                R = oracle_learn(hist, F())
                assert np.abs(R - m).max() <= thr


def test_docs_example():
    # runtests.jl:188-196 / README quick start
    model = np.array([[0.0, 0.1, 0.2], [0.1, 0.0, 0.3], [0.2, 0.3, 0.0]])
    hist = synthetic.enumerate_sample(model, 100000, seed=0)
    learned = oracle_learn(hist)
    assert np.abs(learned - model).max() <= 0.01


def test_synthetic_block_ising_is_deterministic_and_pm1():
    a, J = synthetic.block_ising(32, 1000, block=16, seed=3)
    b, _ = synthetic.block_ising(32, 1000, block=16, seed=3)
    assert a.dtype == np.int8 and a.shape == (1000, 32)
    assert np.array_equal(a, b) and set(np.unique(a)) == {-1, 1}
    assert np.allclose(J, J.T) and np.count_nonzero(J[:16, 16:]) == 0


def test_node_partition_covers_everything():
    from importlib import import_module
    part = import_module("gml_amd.learn")._node_partition
    for n in (3, 9, 1024, 4097):
        for world in (1, 2, 3, 8):
            edges = [part(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_factor_graph_io_and_validation():
    # models.jl:23-54 (validation), :56-76 (show, jsondata), :185-225 (term list / dict constructors),
    # :228-246 (permutations); runtests.jl:17-30 for the matrix round trip of every fixture model
    for name, m in MODELS.items():
        gm = gml.FactorGraph(m)
        gm2 = gml.FactorGraph(gm.to_matrix())
        for key in gm.keys():
            assert gm[key] == gm2[key]
            assert gm[key] == (m[key[0] - 1, key[0] - 1] if len(key) == 1 else m[key[0] - 1, key[1] - 1])
        back = gml.FactorGraph(gm.jsondata())  # JSON term list round trip
        assert back.terms == gm.terms and back.order == gm.order and back.varible_count <= gm.varible_count
    fg = gml.FactorGraph({(1,): 0.5, (2, 3): -0.25, (1, 2, 3): 0.125})
    assert (fg.order, fg.varible_count, fg.alphabet, len(fg)) == (3, 3, "spin", 3)
    assert [d["term"] for d in fg.jsondata()] == [[1], [2, 3], [1, 2, 3]]  # sorted by (length, key)
    assert str(fg).splitlines()[:3] == ["alphabet: spin", "vars: 3", "terms: 3"]
    assert gml.FactorGraph(2, 3, "spin", {(1, 1): 0.1, (3, 3): 0.2, (1, 2): 0.3}).diag_keys() == [(1, 1), (3, 3)]
    assert gml.permutations(range(1, 4), 2) == [(1, 2), (1, 3), (2, 3)]
    assert len(gml.permutations(range(1, 4), 2, asymmetric=True)) == 9
    with pytest.raises(ValueError):
        gml.FactorGraph(2, 3, "spin", {(1, 2, 3): 1.0})  # more indices than the order
    with pytest.raises(ValueError):
        gml.FactorGraph(2, 3, "spin", {(1, 4): 1.0})  # index out of range
    with pytest.raises(ValueError):
        gml.FactorGraph(2, 3, "letters", {(1, 2): 1.0})  # unsupported alphabet
    with pytest.raises(ValueError):
        gml.FactorGraph(2, 3, "spin", {(1, 2): 1.0}, variable_names=["a", "b"])
