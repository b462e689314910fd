"""The C-ABI library loads and exports every symbol include/gml.h declares; without a GPU the
compute entry points fail loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

SO = os.path.join(ROOT, "graphicalmodellearning.jl_amd", "libgml_hip.so")


@pytest.fixture(scope="module")
def cdll():
    if not os.path.exists(SO):
        import __graft_entry__ as ge
        ge.build()
    return C.CDLL(SO)


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gml.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gml_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ["gml_problem_create", "gml_problem_create_spins", "gml_problem_destroy", "gml_objgrad_batch",
                 "gml_learn", "gml_last_error", "gml_lambda", "gml_multi_keys", "gml_bench_pass"]:
        assert must in names


def test_all_declared_symbols_exported(cdll):
    for name in declared_functions():
        assert hasattr(cdll, name), f"{name} declared in include/gml.h but not exported"


def test_lambda_and_default_opts(cdll):
    cdll.gml_lambda.restype = C.c_double
    cdll.gml_lambda.argtypes = [C.c_double, C.c_int64, C.c_double]
    # :157 lambda = c*sqrt(log(n^2/0.05)/M)
    assert cdll.gml_lambda(0.4, 1024, 1e6) == pytest.approx(0.4 * np.sqrt(np.log(1024 ** 2 / 0.05) / 1e6), rel=1e-15)


def test_no_gpu_fails_loudly(cdll):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    s = np.ascontiguousarray(np.loadtxt(os.path.join(ROOT, "tests", "golden", "a_samples.csv"), delimiter=","))
    h = C.c_void_p()
    cdll.gml_problem_create.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                        C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    rc = cdll.gml_problem_create(s.ctypes.data_as(C.c_void_p), 3, 8, 3, 4, 0, 2, 0, 3, 0, C.byref(h))
    assert rc == 3  # GML_EHIP
    cdll.gml_last_error.restype = C.c_char_p
    assert b"no HIP device" in cdll.gml_last_error()


def test_bad_arguments_rejected_before_any_device_work(cdll):
    cdll.gml_problem_create.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                        C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    assert cdll.gml_problem_create(None, 3, 8, 3, 4, 0, 2, 0, 3, 0, C.byref(h)) == 1  # GML_EINVAL
    s = np.zeros((8, 4))
    assert cdll.gml_problem_create(s.ctypes.data_as(C.c_void_p), 9, 8, 3, 4, 0, 2, 0, 3, 0, C.byref(h)) == 1
    assert cdll.gml_problem_create(s.ctypes.data_as(C.c_void_p), 3, 8, 3, 2, 0, 2, 0, 3, 0, C.byref(h)) == 1
