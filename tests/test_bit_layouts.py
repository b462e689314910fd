"""Host-side unit test of the bit-image layout helpers (graphicalmodellearning.jl_amd/csrc/gml_bits.h): the
permutations that tie the sign-bit rows of the spins to the MFMA operand images.  Pure g++, no GPU."""
import os
import subprocess

from conftest import ROOT


def test_bit_layout_helpers(tmp_path):
    exe = str(tmp_path / "bits_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "bits_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
