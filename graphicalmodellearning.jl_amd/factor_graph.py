"""FactorGraph container -- the return type of learn(..., multiRISE) (models.jl:8-20,
GraphicalModelLearning.jl:151) and the matrix <-> dict <-> term-list glue around the learn() path
(models.jl:23-54 validation, :56-76 show / jsondata, :79-98 container protocol, :104-225 conversions,
:228-246 permutations).  Host-side glue, no arithmetic; keys are 1-based tuples like the reference's."""
import warnings
from collections.abc import Mapping

import numpy as np

ALPHABETS = ("spin", "boolean", "integer", "integer_pos", "real", "real_pos")  # models.jl:5


def permutations(items, order, asymmetric=False):
    """permutations(items, order; asymmetric) (models.jl:228-246): sorted tuples of `order` items, strictly
    ascending unless asymmetric (then every ordered tuple, repeats included, like the reference)."""
    items = list(items)

    def rec(partial, left):
        if left == 0:
            return [tuple(partial)]
        out = []
        for it in items:
            if not asymmetric and partial and partial[-1] >= it:
                continue
            out.extend(rec(partial + [it], left - 1))
        return out

    return sorted(rec([], int(order)))


def check_model_data(order, varible_count, alphabet, terms, variable_names=None):
    """models.jl:23-54: raises ValueError where the reference calls error()."""
    if alphabet not in ALPHABETS:
        raise ValueError(f"alphabet {alphabet} is not supported")
    if variable_names is not None and len(variable_names) != varible_count:
        raise ValueError(f"expected {varible_count} but only given {len(variable_names)}")
    for k in terms:
        if len(k) > order:
            raise ValueError(f"a term has {len(k)} indices but should have {order} indices")
        for index in k:
            if index < 1 or index > varible_count:
                raise ValueError(f"a term has an index of {index} but it should be in the range of 1:{varible_count}")
    return True


class TermArray(Mapping):
    """The `terms` of a learned multi-body model without a hash table: ONE float64 array of weights in the order the reference
    lists a model's terms in (models.jl:61,72: sorted by (length(key), key)), as gml_learn_terms / gml_terms_assemble leave it
    (include/gml.h).  A key <-> its position is closed-form (gml_terms_rank / gml_terms_keys), so the 22.5 M terms of config 5
    (n = 512, order 3) cost their 179 MB of weights and nothing else; a Python dict of them would take ~25 GB and minutes.

    Read-only Mapping with the reference's 1-based tuple keys: len, `ta[(1, 2, 3)]`, `in`, iteration in (length, key) order, items().
      symmetrized   True: keys are the ascending tuples (:135-149); False: (u, ascending others) for every node u (:129-132)
      weights       the array itself (values() returns it, no copy)
      keys_array(first, count)   int32 [count, order], 1-based, 0 = unused slot, for vectorised consumers
    to_dict() materialises the reference's Dict (small models)."""

    CHUNK = 1 << 16

    def __init__(self, varible_count, order, symmetrized, weights):
        from . import _lib
        self._lib = _lib
        self.varible_count, self.order, self.symmetrized = int(varible_count), int(order), bool(symmetrized)
        self.weights = np.asarray(weights, dtype=np.float64)
        if self.weights.shape != (_lib.terms_count(self.varible_count, self.order, self.symmetrized),):
            raise ValueError(f"{self.weights.shape[0]} weights for a model of {self.varible_count} spins at order {self.order}")

    def __len__(self):
        return self.weights.shape[0]

    def rank(self, key):
        """position of the 1-based key, -1 if the model has no such term"""
        try:
            k0 = [int(i) - 1 for i in key]
        except (TypeError, ValueError):
            return -1
        if not 1 <= len(k0) <= self.order:
            return -1
        return self._lib.terms_rank(self.varible_count, self.order, self.symmetrized, k0)

    def __getitem__(self, key):
        t = self.rank(key)
        if t < 0:
            raise KeyError(key)
        return float(self.weights[t])

    def __contains__(self, key):
        return self.rank(key) >= 0

    def keys_array(self, first=0, count=None):
        k = self._lib.terms_keys(self.varible_count, self.order, self.symmetrized, first, count)
        k += 1  # 1-based like the reference's tuples; the unused slots (-1) become 0
        return k

    def _chunks(self):
        T = len(self)
        for a in range(0, T, self.CHUNK):
            k = self.keys_array(a, min(self.CHUNK, T - a))
            lens = (k > 0).sum(axis=1)
            yield a, k, lens

    def __iter__(self):
        for _, k, lens in self._chunks():
            for row, ln in zip(k.tolist(), lens.tolist()):
                yield tuple(row[:ln])

    def items(self):
        for a, k, lens in self._chunks():
            for row, ln, v in zip(k.tolist(), lens.tolist(), self.weights[a:a + len(k)].tolist()):
                yield tuple(row[:ln]), v

    def values(self):
        return self.weights

    def to_dict(self):
        return dict(self.items())

    def __eq__(self, other):
        if isinstance(other, TermArray):
            return (self.varible_count, self.order, self.symmetrized) == (other.varible_count, other.order, other.symmetrized) \
                and np.array_equal(self.weights, other.weights)
        if isinstance(other, Mapping):
            return len(other) == len(self) and all(k in other and other[k] == v for k, v in self.items())
        return NotImplemented

    __hash__ = None

    def __repr__(self):
        return f"TermArray(vars={self.varible_count}, order={self.order}, symmetrized={self.symmetrized}, terms={len(self)})"


class FactorGraph:
    def __init__(self, order_or_data, varible_count=None, alphabet="spin", terms=None, variable_names=None):
        if varible_count is None:  # FactorGraph(matrix) / FactorGraph(dict)  (models.jl:18-19)
            data = order_or_data
            if isinstance(data, dict):  # models.jl:209-225
                terms = {tuple(k): v for k, v in data.items()}
                assert all(min(k) > 0 for k in terms)
                order = max(len(k) for k in terms)
                varible_count = max(max(k) for k in terms)
            elif isinstance(data, (list, tuple)):  # the jsondata() term list, models.jl:185-206
                terms = {tuple(item["term"]): item["weight"] for item in data}
                assert all(min(k) > 0 for k in terms)
                order = max(len(k) for k in terms)
                varible_count = max(max(k) for k in terms)
            else:
                m = np.asarray(data, dtype=float)
                assert m.ndim == 2 and m.shape[0] == m.shape[1]
                iu = np.triu_indices(m.shape[0], 1)
                bad = (m.T[iu] != 0) & (m[iu] != m.T[iu])
                if bad.any():  # models.jl:126-130: only the (i<j) entry is used
                    warnings.warn(f"{int(bad.sum())} matrix entries differ from their transposes; the upper triangle is used")
                terms = matrix_to_terms(m, asymmetric=False)
                order, varible_count = 2, m.shape[0]
            self.order, self.varible_count, self.alphabet, self.terms = order, varible_count, "spin", terms
        elif isinstance(terms, TermArray):  # array-backed terms of a learned model: valid keys by construction
            self.order, self.varible_count, self.alphabet, self.terms = int(order_or_data), int(varible_count), alphabet, terms
            if (terms.order, terms.varible_count) != (self.order, self.varible_count):
                raise ValueError(f"the term array is of order {terms.order} over {terms.varible_count} variables")
        else:
            self.order, self.varible_count, self.alphabet, self.terms = int(order_or_data), int(varible_count), alphabet, dict(terms)
        self.variable_names = variable_names
        check_model_data(self.order, self.varible_count, self.alphabet, {} if isinstance(self.terms, TermArray) else self.terms,
                         variable_names)  # models.jl:12

    # models.jl:79-85
    def __iter__(self):
        return iter(self.terms.items())

    def __len__(self):
        return len(self.terms)

    def __getitem__(self, key):
        return self.terms[tuple(key)]

    def keys(self):
        return self.terms.keys()

    def _sorted_items(self):
        """(key, weight) in the reference's listing order (models.jl:61,72); an array-backed store is in that order already"""
        if isinstance(self.terms, TermArray):
            return self.terms.items()
        return ((k, self.terms[k]) for k in sorted(self.terms, key=lambda x: (len(x), x)))

    def to_matrix(self):  # convert(Array{T,2}, gm)  models.jl:137-154
        if self.order != 2:
            raise ValueError(f"cannot convert a FactorGraph of order {self.order} to a matrix")
        m = np.zeros((self.varible_count, self.varible_count))
        if isinstance(self.terms, TermArray) and self.terms.symmetrized:
            n, w = self.varible_count, self.terms.weights
            m[np.arange(n), np.arange(n)] = w[:n]
            iu = np.triu_indices(n, 1)  # (i < j) row by row = the lexicographic order of the pair keys
            m[iu] = w[n:]
            m.T[iu] = w[n:]
            return m
        for k, v in self.terms.items():
            if len(k) == 1:
                m[k[0] - 1, k[0] - 1] = v
            else:
                m[k[0] - 1, k[1] - 1] = v
                m[k[1] - 1, k[0] - 1] = v
        return m

    def iter_jsondata(self):
        """jsondata() as a stream (a model of tens of millions of terms is written out without holding the list)"""
        for k, v in self._sorted_items():
            yield {"term": list(k), "weight": v}

    def jsondata(self):  # models.jl:70-76
        return list(self.iter_jsondata())

    def diag_key(self, i):  # models.jl:98
        return (i,) * self.order

    def diag_keys(self):  # models.jl:87-96
        return sorted(self.diag_key(i) for i in range(1, self.varible_count + 1) if self.diag_key(i) in self.terms)

    def __repr__(self):
        return f"FactorGraph(order={self.order}, vars={self.varible_count}, terms={len(self.terms)})"

    def __str__(self):  # Base.show, models.jl:56-68
        lines = [f"alphabet: {self.alphabet}", f"vars: {self.varible_count}"]
        if self.variable_names is not None:
            lines += ["variable names: ", f"  {self.variable_names}"]
        lines.append(f"terms: {len(self.terms)}")
        lines += [f"  {k} => {v}" for k, v in self._sorted_items()]
        return "\n".join(lines)


def matrix_to_terms(m, asymmetric=True):
    """convert(Dict, matrix) (models.jl:157-182; asymmetric=True keeps (i,j) and (j,i)) and the
    symmetric variant used by FactorGraph(matrix) (models.jl:105-134).  Entries ~0 are dropped
    (`isapprox(weight, 0.0)` is an exact-zero test in Julia)."""
    n = m.shape[0]
    terms = {}
    for i in range(n):
        if m[i, i] != 0.0:
            terms[(i + 1,)] = float(m[i, i])
    for i in range(n):
        for j in range(n):
            if i == j or (not asymmetric and i > j):
                continue
            if m[i, j] != 0.0:
                terms[(i + 1, j + 1)] = float(m[i, j])
    return terms
