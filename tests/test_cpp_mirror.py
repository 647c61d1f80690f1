"""include/rdst.hpp (the C++ mirror of rdst's surface): compiles everywhere, runs on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_rdst_hpp.cpp")


def _build(tmp_path, hiplib):
    exe = str(tmp_path / "test_rdst_hpp")
    libdir = os.path.join(ROOT, "rdst_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", libdir, "-lrdst_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.run(cmd, check=True)
    return exe


def test_header_compiles_and_links(tmp_path, hiplib):
    assert os.path.exists(_build(tmp_path, hiplib))


@pytest.mark.gpu
def test_cpp_caller_matches_std_sort(tmp_path, gpu, hiplib):
    exe = _build(tmp_path, hiplib)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
