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


# ---- the C ABI from a plain C host --------------------------------------------------------------

def _build_c_example(tmp_path):
    exe = str(tmp_path / "solve_batch")
    libdir = os.path.join(ROOT, "fiksi_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "solve_batch.c"), "-o", exe, "-L", libdir, "-lfiksi_amd", f"-Wl,-rpath,{libdir}", "-lm"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_headers_are_c99_and_the_c_example_links(built, tmp_path):
    for h in ("fiksi_amd.h", "fiksi_amd_builder.h"):
        out = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-x", "c", "-fsyntax-only",
                              os.path.join(ROOT, "include", h)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
    _build_c_example(tmp_path)


@pytest.mark.gpu
def test_c_example_solves(built, tmp_path):
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "system 1:" in out.stdout
