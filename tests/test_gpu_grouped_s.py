"""The grouped kernel's SPARSE build for batches of one structure (fx_grouped_s.hip: components of 49 ... 255 free variables with
a small Cholesky factor; four Systems per wavefront, the factorisation a level schedule over tables in LDS) against the oracle and
against the paths such batches took before it (the wide kernel, the team kernels). It is the normal-equation step in a
minimum-degree elimination order: the oracle's accepted-step / trial counts, positions to 1e-8 — the bar of tests/test_gpu_wide.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("n_tri", [8, 11, 12, 13, 16, 20, 25, 31])
def test_hinged_chains_follow_the_oracle(fiksi, oracle, ctx, n_tri):
    """Chains of n hinged triangles (`add_hinged_triangles`, fiksi_bench.rs:15-40): 4n + 2 variables (34 .. 126; up to 48 variables
    the sparse build takes a structure when its factor has at most a quarter of the dense triangle's entries — these do)."""
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(24, n_tri)
    db = ctx.upload(b)
    assert db.grouped_build() == 2
    db.free()
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    assert np.array_equal(res["scale"], res_o["scale"])
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["trials"], res_o["trials"])
    assert np.array_equal(res["exit"], res_o["exit"])
    assert np.allclose(res["sse0"], res_o["sse0"], rtol=1e-12, atol=0)
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-6, atol=1e-12)
    assert np.max(np.abs(v - v_o)) < 1e-8
    assert np.all(res["sse_unscaled"] < 1e-4)


def _chain_with_diagonals(n_sys, n_pts, fix_first, seed0=0):
    """A strip of points with distances to the next and the next but one (a banded, not an arrow-shaped, factor), perturbed
    per System; optionally the first point fixed."""
    import fiksi_amd as F
    from fiksi_amd import workloads

    from helpers import Lcg

    out = []
    for k in range(n_sys):
        g = Lcg(seed0 + k)
        s = F.System()
        pts = [F.elements.Point.create(s, 1.1 * i + g.u(-0.05, 0.05), 0.35 * ((i * 7) % 5) + g.u(-0.05, 0.05)) for i in range(n_pts)]
        for i in range(n_pts - 1):
            F.constraints.PointPointDistance.create(s, pts[i], pts[i + 1], 1.3)
        for i in range(n_pts - 2):
            F.constraints.PointPointDistance.create(s, pts[i], pts[i + 2], 2.2)
        if fix_first:
            pts[0].fix(s)
        out.append(s.flatten())
    return workloads.concat(out)


@pytest.mark.parametrize("fix_first", [False, True])
def test_banded_structures_and_fixed_points_follow_the_oracle(fiksi, oracle, ctx, fix_first):
    b = _chain_with_diagonals(16, 35, fix_first)
    db = ctx.upload(b)
    assert db.grouped_build() == 2
    db.free()
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["trials"], res_o["trials"])
    assert np.array_equal(res["exit"], res_o["exit"])
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-5, atol=1e-12)
    assert np.max(np.abs(v - v_o)) < 1e-7
    if fix_first:  # fixed values stay bit-identical
        fixed = np.asarray(b["var_fixed"]) != 0
        assert np.array_equal(_bits(v[fixed]), _bits(np.asarray(b["vars"])[fixed]))


def test_against_the_paths_it_replaces_and_deterministic(fiksi, ctx):
    """The same counters as the team / wide kernels on the same batch, variables to round-off; a second solve of the resident batch,
    the host-buffer call, the ladder off / everywhere and the trial cap give the bits of the first."""
    from fiksi_amd import abi, workloads

    b = workloads.hinged_triangles(3000, 16)
    db = ctx.upload(b)
    assert db.grouped_build() == 2 and db.grouped_build(abi.solving_opts(solver=1)) != 2 and db.grouped_build(abi.solving_opts(decomposer=1)) != 2
    db.system_solve()
    v1, r1 = db.get_vars().copy(), db.get_results().copy()
    db.system_solve()
    assert np.array_equal(_bits(db.get_vars()), _bits(v1)) and db.get_results().tobytes() == r1.tobytes()
    db.free()
    v2, r2 = ctx.system_solve_batch(b)
    assert np.array_equal(_bits(v2), _bits(v1)) and r2.tobytes() == r1.tobytes()
    try:
        for ladder in ((False, 0, 16, False), (True, 1 << 30, 0, True)):
            ctx.set_ladder(*ladder)
            v3, r3 = ctx.system_solve_batch(b)
            assert np.array_equal(_bits(v3), _bits(v1)) and r3.tobytes() == r1.tobytes(), ladder
    finally:
        ctx.set_ladder()
    ctx.set_one_structure_builds(False)
    try:
        v0, r0 = ctx.system_solve_batch(b)
    finally:
        ctx.set_one_structure_builds(True)
    for f in ("accepted", "trials", "exit", "ncomp", "scale"):
        assert np.array_equal(r0[f], r1[f]), f
    assert np.max(np.abs(v0 - v1)) < 1e-10
    vc, rc = ctx.system_solve_batch(workloads.hinged_triangles(100, 20), abi.solving_opts(max_trials=3))
    assert int(rc["trials"].max()) == 3 and np.all(rc["exit"] == 4)


def test_what_does_not_qualify_keeps_its_path(fiksi, ctx):
    from fiksi_amd import abi, workloads

    for b, why in ((workloads.hinged_triangles(6, 16), "fewer than 8 Systems"), (workloads.hinged_triangles(64, 64), "more than 255 variables"),
                   (workloads.concat([workloads.hinged_triangles(20, 16), workloads.hinged_triangles(20, 17)]), "two structures")):
        db = ctx.upload(b)
        assert db.grouped_build() != 2, why
        db.free()
    db = ctx.upload(workloads.hinged_triangles(64, 16))
    assert db.grouped_build() == 2 and db.grouped_build(abi.solving_opts(f32=True)) != 2 and db.grouped_build(abi.solving_opts(optimizer=1)) != 2
    db.free()
