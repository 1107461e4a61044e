"""The multifrontal plan of the large-component path (fiksi_amd/csrc/fx_front_plan.h) on the CPU: tests/cpp/front_harness.cpp
builds the plan of a sketch the way the library does and walks its segment blobs in scalar C++ exactly as the kernels of
fx_front.h do (staging tile, entries of A by record, the children's contribution blocks through their byte maps, the partial
Cholesky of a 16-lane row, L / contribution storage with their padding left NaN, the sweep down); the step must equal a dense
Cholesky solve of (Jt J + lambda I) x = -Jt r on random values over the sketch's pattern. No GPU."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fronts") / "front_harness")
    cmd = ["g++", "-std=c++17", "-O2", "-DFX_HOST_ONLY", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "front_harness.cpp"), "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


@pytest.mark.parametrize("kind,n", [("chain", 60), ("chain", 400), ("chain", 900), ("hinged", 40)])
def test_generated_sketches(harness, kind, n):
    """chain: BASELINE cfg2's make-up (chain + skip distances, three-point angles); 900 points = 1 800 columns is past the size
    from which the tree is cut into parts + top, so both plans are walked."""
    out = subprocess.run([harness, kind, str(n), "11"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "front plan ok" in out.stdout, out.stdout + out.stderr
    assert "no multifrontal build" not in out.stdout
    if n == 900:
        assert " parts: " in out.stdout and " solo: " in out.stdout


def _dump(b, d):
    from fiksi_amd import abi

    a = abi.normalize_batch(b)
    v1, e1 = int(a["var_off"][1]), int(a["expr_off"][1])
    a["var_fixed"][:v1].astype(np.uint8).tofile(os.path.join(d, "var_fixed.u8"))
    a["expr_tag"][:e1].astype(np.uint8).tofile(os.path.join(d, "expr_tag.u8"))
    a["expr_idx"][:4 * e1].astype(np.uint32).tofile(os.path.join(d, "expr_idx.u32"))


@pytest.mark.parametrize("name", ["hinged_64", "hinged_16", "large_300", "large_1500", "large_300_fixed_points"])
def test_the_library_s_own_workloads(harness, fiksi, tmp_path, name):
    """The reference's bench sketches (fiksi_bench.rs:15-40) and the large-sketch generator, as the GPU tests solve them."""
    from fiksi_amd import workloads

    b = {"hinged_64": lambda: workloads.hinged_triangles(1, 64), "hinged_16": lambda: workloads.hinged_triangles(1, 16),
         "large_300": lambda: workloads.large_sketch(300, seed=3), "large_1500": lambda: workloads.large_sketch(1500, seed=5),
         "large_300_fixed_points": lambda: workloads.large_sketch(300, seed=9)}[name]()
    if name == "large_300_fixed_points":
        b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
        b["var_fixed"][[0, 1, 200, 201, 202, 203]] = 1
    _dump(b, str(tmp_path))
    out = subprocess.run([harness, "file", str(tmp_path), "5"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "front plan ok" in out.stdout, out.stdout + out.stderr
    assert "no multifrontal build" not in out.stdout


def test_a_structure_with_wide_separators_keeps_the_walkers(harness):
    """A grid's separators grow with its side: a front beyond 15 columns, no multifrontal build — said, not forced."""
    out = subprocess.run([harness, "grid", "12", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "no multifrontal build" in out.stdout, out.stdout + out.stderr
