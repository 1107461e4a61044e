"""The C++ mirror (include/fiksi.hpp) compiles against the C ABI with a plain host compiler and runs the
reference's sketches: build-only on CPU, with solves on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_fiksi_cpp.cpp")


def _build(tmp_path):
    exe = str(tmp_path / "test_fiksi_cpp")
    libdir = os.path.join(ROOT, "fiksi_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-lfiksi_amd", f"-Wl,-rpath,{libdir}"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_cpp_mirror_compiles_and_builds_systems(built, tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "--build-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "build-only ok" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_solves_reference_sketches(built, tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ mirror checks passed" in out.stdout
