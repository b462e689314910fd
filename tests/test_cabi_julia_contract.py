"""Contract between include/gml.h and the Julia binding (julia/GraphicalModelLearningHIP.jl), checked through a C program
that calls the ABI exactly as the `ccall`s do (tests/native/cabi_julia_contract.c): struct layouts and constants on the
CPU box, the column-major Int64 call sequence against the reference's golden vectors on the GPU box."""
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

SRC = os.path.join(ROOT, "tests", "native", "cabi_julia_contract.c")
PKG = os.path.join(ROOT, "graphicalmodellearning.jl_amd")
JL = os.path.join(PKG, "julia", "GraphicalModelLearningHIP.jl")
JL_SIZES = {"Cdouble": 8, "Float64": 8, "Int64": 8, "Int32": 4, "Cint": 4, "UInt32": 4}


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if not os.path.exists(os.path.join(PKG, "libgml_hip.so")):
        import __graft_entry__ as ge
        ge.build()
    out = str(tmp_path_factory.mktemp("cabi") / "cabi_julia_contract")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", out, "-L", PKG, "-lgml_hip",
                           f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return out


def julia_struct(name):
    """[(field, type)] of `struct name ... end` in the .jl file"""
    text = open(JL).read()
    body = re.search(r"struct %s\b(.*?)\nend" % name, text, re.S).group(1)
    body = re.sub(r"#.*", "", body)
    return re.findall(r"(\w+)::(\w+)", body)


def c_layout(fields):
    """offsets and size of a C struct with natural alignment (what Julia gives an isbits struct of these types)"""
    off, out, align = 0, {}, 1
    for f, t in fields:
        sz = JL_SIZES[t]
        off = (off + sz - 1) // sz * sz
        out[f] = off
        off += sz
        align = max(align, sz)
    return out, (off + align - 1) // align * align


def test_struct_layouts_and_constants_match_the_julia_file(exe):
    txt = subprocess.run([exe, "layout"], check=True, capture_output=True, text=True).stdout
    got = dict(re.findall(r"^(?:sizeof |const )?(\S+) (\d+)$", txt, re.M))
    for jl_name, c_name in (("GmlOpts", "gml_opts"), ("GmlStats", "gml_stats")):
        offs, size = c_layout(julia_struct(jl_name))
        assert int(got[c_name]) == size, (c_name, got[c_name], size)
        for f, o in offs.items():
            assert int(got[f"{c_name}.{f}"]) == o, (c_name, f)
        assert len(offs) == sum(1 for k in got if k.startswith(c_name + "."))  # no field of the C struct is missing in Julia
    # the ABI identity the .jl file checks in __init__: its constant = the header's = what the library answers, and the library's
    # struct sizes = the sizes of the Julia mirrors
    text = open(JL).read()
    calls = dict(re.findall(r"^call (\S+) (\d+)$", txt, re.M))
    assert int(re.search(r"const GML_ABI_VERSION = Cint\((\d+)\)", text).group(1)) == int(got["GML_ABI_VERSION"]) == int(calls["gml_abi_version"])
    assert int(calls["gml_sizeof_opts"]) == c_layout(julia_struct("GmlOpts"))[1]
    assert int(calls["gml_sizeof_stats"]) == c_layout(julia_struct("GmlStats"))[1]
    import gml_amd as gml
    assert gml._lib.GML_ABI_VERSION == int(got["GML_ABI_VERSION"])
    # the integer constants the .jl file hard-codes
    for names, vals in re.findall(r"const ((?:GML_\w+(?:, )?)+) = ((?:Cint\(\d+\)(?:, )?)+)", text):
        for nm, v in zip(names.split(", "), re.findall(r"Cint\((\d+)\)", vals)):
            assert int(got[nm]) == int(v), nm


@pytest.mark.gpu
@pytest.mark.parametrize("name,c,sym", [("a", 0.4, 1), ("c", 0.4, 1), ("mvt", 0.2, 0)])
def test_ccall_sequence_reproduces_the_goldens(exe, name, c, sym):
    r = subprocess.run([exe, "run", os.path.join(GOLDEN, f"{name}_samples.csv"), os.path.join(GOLDEN, f"{name}_RISE_learned.csv"), str(c), str(sym)],
                       capture_output=True, text=True)
    print(r.stdout, r.stderr)
    if name == "mvt":  # the mvt goldens carry Ipopt's termination state: 2e-4; the call sequence is what is tested
        m = re.findall(r"max_abs_diff (\S+)", r.stdout)
        assert len(m) == 2 and all(float(v) <= 3e-4 for v in m), r.stdout + r.stderr
        assert r.returncode in (0, 11)
    else:
        assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("sym", [1, 0])
def test_ccall_sequence_of_the_multirise_method(exe, sym):
    # learn(samples, multiRISE(0.2, sym, 3), HIP()) as the .jl calls the ABI -- gml_terms_count, gml_problem_create(order 3),
    # gml_learn_terms, gml_terms_keys on a column-major Int64 matrix -- against the Python front door on the same fixture
    import numpy as np

    import gml_amd as gml
    r = subprocess.run([exe, "terms", os.path.join(GOLDEN, "c_samples.csv"), "0.2", str(sym), "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr
    head = re.search(r"nterms (\d+) not_converged (\d+) t_assemble_positive (\d)", r.stdout)
    got = {tuple(int(v) for v in ln.split()[1:-1]): float(ln.split()[-1]) for ln in r.stdout.splitlines() if ln.startswith("term")}
    s = np.loadtxt(os.path.join(GOLDEN, "c_samples.csv"), delimiter=",")
    fg = gml.learn(s, gml.multiRISE(0.2, bool(sym), 3), gml.HIP(tol=1e-10))
    assert int(head.group(1)) == len(got) == len(fg) and int(head.group(2)) == 0 and int(head.group(3)) == 1
    assert got.keys() == fg.terms.keys() and all(got[k] == v for k, v in fg.terms.items())  # the same library call: the same bits
    assert list(got) == sorted(got, key=lambda k: (len(k), k))                                 # in the reference's listing order
