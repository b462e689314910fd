"""Minimal FactorGraph container -- the return type of learn(..., multiRISE) (models.jl:8-20,
GraphicalModelLearning.jl:151) and the matrix <-> dict glue the reference's tests use around the
learn() path (models.jl:104-182).  Host-side glue, no arithmetic; keys are 1-based tuples like
the reference's."""
import numpy as np


class FactorGraph:
    def __init__(self, order_or_data, varible_count=None, alphabet="spin", terms=None, variable_names=None):
        if varible_count is None:  # FactorGraph(matrix) / FactorGraph(dict)  (models.jl:18-19)
            data = order_or_data
            if isinstance(data, dict):
                terms = dict(data)
                order = max(len(k) for k in terms)
                varible_count = max(max(k) for k in terms)
            else:
                m = np.asarray(data, dtype=float)
                assert m.ndim == 2 and m.shape[0] == m.shape[1]
                terms = matrix_to_terms(m, asymmetric=False)
                order, varible_count = 2, m.shape[0]
            self.order, self.varible_count, self.alphabet, self.terms = order, varible_count, "spin", terms
        else:
            self.order, self.varible_count, self.alphabet, self.terms = int(order_or_data), int(varible_count), alphabet, dict(terms)
        self.variable_names = variable_names

    # models.jl:79-85
    def __iter__(self):
        return iter(self.terms.items())

    def __len__(self):
        return len(self.terms)

    def __getitem__(self, key):
        return self.terms[tuple(key)]

    def keys(self):
        return self.terms.keys()

    def to_matrix(self):  # convert(Array{T,2}, gm)  models.jl:137-154
        if self.order != 2:
            raise ValueError(f"cannot convert a FactorGraph of order {self.order} to a matrix")
        m = np.zeros((self.varible_count, self.varible_count))
        for k, v in self.terms.items():
            if len(k) == 1:
                m[k[0] - 1, k[0] - 1] = v
            else:
                m[k[0] - 1, k[1] - 1] = v
                m[k[1] - 1, k[0] - 1] = v
        return m

    def jsondata(self):  # models.jl:70-76
        return [{"term": list(k), "weight": self.terms[k]} for k in sorted(self.terms, key=lambda x: (len(x), x))]

    def __repr__(self):
        return f"FactorGraph(order={self.order}, vars={self.varible_count}, terms={len(self.terms)})"


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
