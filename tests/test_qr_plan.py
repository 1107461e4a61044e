"""The host-side symbolic phase of FX_STEP_QR (fiksi_amd/csrc/fx_qrplan.h: COLAMD, elimination tree, Davis 5.3 row
permutation, Householder / R patterns) against the oracle's restatement of SymbolicQr::build
(solvi/src/decomposition/sparse/qr.rs:118-206) — which is pinned by the reference's own KATs
(tests/test_oracle_golden.py). CPU only: no device is needed for planning."""
import numpy as np
import pytest

from helpers import Lcg, mixed_sketch, random_big_sketch, random_sketch


def _same(mine, ref):
    for k in ("col_perm", "row_perm", "h_ptr", "h_rows", "r_ptr", "r_rows"):
        assert np.array_equal(np.asarray(mine[k], dtype=np.int64), np.asarray(ref[k], dtype=np.int64)), k


def _augmented_pattern(flat, fiksi):
    """CSC pattern of [J; sqrt(lambda) I] of a one-component sketch as lm.rs:81-98 builds it."""
    from fiksi_amd import abi

    rp, ci = abi.jacobian_structure(flat)
    m = len(rp) - 1
    n = int(ci.max()) + 1 if len(ci) else 0
    cols = [[] for _ in range(n)]
    for r in range(m):
        for p in range(int(rp[r]), int(rp[r + 1])):
            cols[int(ci[p])].append(r)
    colptr, rowidx = [0], []
    for c in range(n):
        rowidx += cols[c] + [m + c]
        colptr.append(len(rowidx))
    return m + n, n, colptr, rowidx


def test_colamd_kats_of_the_reference(fiksi):
    """colamd_rs/src/lib.rs:253-282: the two documented permutations."""
    from fiksi_amd import abi

    p = abi.qr_symbolic(5, 4, [0, 3, 5, 9, 11], [0, 1, 4, 2, 4, 0, 1, 2, 3, 1, 3])
    assert p["col_perm"].tolist() == [1, 0, 2, 3]
    p = abi.qr_symbolic(3, 3, [0, 2, 4, 5], [0, 1, 1, 2, 0])
    assert sorted(p["col_perm"].tolist()) == [0, 1, 2]


@pytest.mark.parametrize("ordering", ["colamd", "natural"])
def test_random_patterns_match_the_oracle(fiksi, oracle, ordering):
    from fiksi_amd import abi

    g = Lcg(77)
    for trial in range(400):
        n = int(g.u(1, 40.99))
        m = int(g.u(0, 60.99))
        dens = g.u(0.03, 0.5)
        colptr, rowidx = [0], []
        for c in range(n):
            rows = [r for r in range(m) if g.f() < dens]
            rowidx += rows + [m + c]  # a damping row per column keeps the pattern structurally full rank
            colptr.append(len(rowidx))
        mine = abi.qr_symbolic(m + n, n, colptr, rowidx, colamd=(ordering == "colamd"))
        ref = oracle.symbolic_qr(m + n, n, colptr, rowidx, ordering)
        _same(mine, ref)


def test_dense_rows_and_columns(fiksi, oracle):
    """Rows / columns beyond COLAMD's density thresholds (10 sqrt(n), at least 16) are set aside first."""
    from fiksi_amd import abi

    g = Lcg(5)
    for n, m in ((4, 80), (30, 200), (60, 250), (64, 256)):
        colptr, rowidx = [0], []
        for c in range(n):
            dens = 0.9 if c % 5 == 0 else 0.05
            rows = [r for r in range(m) if g.f() < dens or r % 37 == 0]
            rowidx += rows + [m + c]
            colptr.append(len(rowidx))
        _same(abi.qr_symbolic(m + n, n, colptr, rowidx), oracle.symbolic_qr(m + n, n, colptr, rowidx, "colamd"))


def test_sketch_patterns_match_the_oracle(fiksi, oracle):
    from fiksi_amd import abi, workloads

    flats = [workloads.ring16(1), workloads.hinged_triangles(1, 11), workloads.hinged_triangles(1, 4), workloads.quadrilateral()]
    flats += [mixed_sketch(s).flatten() for s in range(6)]
    flats += [random_sketch(s).flatten() for s in range(40)]
    flats += [random_big_sketch(100 + s, 24).flatten() for s in range(6)]
    done = 0
    for f in flats:
        vc = f.get("var_comp")
        if vc is not None and len(set(int(c) for c in vc if c != 0xFFFF)) > 1:
            continue  # one matrix per component; the GPU tests cover multi-component sketches end to end
        nrows, n, colptr, rowidx = _augmented_pattern(f, fiksi)
        if n == 0 or n > 64:
            continue
        _same(abi.qr_symbolic(nrows, n, colptr, rowidx), oracle.symbolic_qr(nrows, n, colptr, rowidx, "colamd"))
        done += 1
    assert done >= 20


def test_malformed_patterns_are_rejected(fiksi):
    from fiksi_amd import abi
    from fiksi_amd._lib import FiksiError

    with pytest.raises(FiksiError):
        abi.qr_symbolic(3, 2, [0, 2, 4], [1, 0, 0, 2])  # rows not ascending
    with pytest.raises(FiksiError):
        abi.qr_symbolic(3, 2, [0, 2, 2], [0, 1])  # empty column: no row to pivot on
