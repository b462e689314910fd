"""FactorGraph container -- the return type of learn(..., multiRISE) (models.jl:8-20,
GraphicalModelLearning.jl:151) and the matrix <-> dict <-> term-list glue around the learn() path
(models.jl:23-54 validation, :56-76 show / jsondata, :79-98 container protocol, :104-225 conversions,
:228-246 permutations).  Host-side glue, no arithmetic; keys are 1-based tuples like the reference's."""
import warnings

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
        else:
            self.order, self.varible_count, self.alphabet, self.terms = int(order_or_data), int(varible_count), alphabet, dict(terms)
        self.variable_names = variable_names
        check_model_data(self.order, self.varible_count, self.alphabet, self.terms, variable_names)  # models.jl:12

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
        lines += [f"  {k} => {self.terms[k]}" for k in sorted(self.terms, key=lambda x: (len(x), x))]
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
