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


def _random_graph_batch(n_sys, n_pts, n_extra, seed, fix_first=False, angles=0):
    """n_sys sketches of ONE random structure: n_pts points, a random spanning tree of distances, n_extra more distances between
    random pairs and `angles` three-point angles — consistent targets from a jittered truth, start values perturbed per System."""
    from fiksi_amd import abi

    from helpers import Lcg

    g = Lcg(seed)
    edges = []
    for i in range(1, n_pts):
        edges.append((int(g.u(0, i - 1e-9)), i))
    have = set(edges)
    while len(edges) < n_pts - 1 + n_extra:
        a, c = int(g.u(0, n_pts - 1e-9)), int(g.u(0, n_pts - 1e-9))
        if a != c and (min(a, c), max(a, c)) not in have:
            have.add((min(a, c), max(a, c)))
            edges.append((min(a, c), max(a, c)))
    tris = [tuple(sorted({int(g.u(0, n_pts - 1e-9)) for _ in range(3)})) for _ in range(angles)]
    tris = [t for t in tris if len(t) == 3]
    m = len(edges) + len(tris)
    nv = 2 * n_pts
    vars_ = np.zeros((n_sys, nv))
    par = np.zeros((n_sys, m))
    base = np.array([[g.u(-10, 10), g.u(-10, 10)] for _ in range(n_pts)])
    for k in range(n_sys):
        truth = base + np.array([[g.u(-0.3, 0.3), g.u(-0.3, 0.3)] for _ in range(n_pts)])
        for r, (a, c) in enumerate(edges):
            par[k, r] = np.hypot(*(truth[a] - truth[c]))
        for r, (a, bq, c) in enumerate(tris):
            u, v = truth[a] - truth[bq], truth[c] - truth[bq]
            ang = np.arctan2(v[1], v[0]) - np.arctan2(u[1], u[0])
            par[k, len(edges) + r] = (ang + np.pi) % (2 * np.pi) - np.pi
        vars_[k] = (truth + np.array([[g.u(-0.15, 0.15), g.u(-0.15, 0.15)] for _ in range(n_pts)])).reshape(-1)
    tag = np.zeros((n_sys, m), dtype=np.uint8)
    idx = np.zeros((n_sys, m, 4), dtype=np.uint32)
    for r, (a, c) in enumerate(edges):
        tag[:, r] = abi.POINT_POINT_DISTANCE
        idx[:, r, 0], idx[:, r, 1] = 2 * a, 2 * c
    for r, (a, bq, c) in enumerate(tris):
        tag[:, len(edges) + r] = abi.POINT_POINT_POINT_ANGLE
        idx[:, len(edges) + r, 0], idx[:, len(edges) + r, 1], idx[:, len(edges) + r, 2] = 2 * a, 2 * bq, 2 * c
    fixed = np.zeros((n_sys, nv), dtype=np.uint8)
    if fix_first:
        fixed[:, 0:2] = 1
    return {"var_off": (np.arange(n_sys + 1, dtype=np.uint64) * nv).astype(np.uint32), "expr_off": (np.arange(n_sys + 1, dtype=np.uint64) * m).astype(np.uint32),
            "vars": vars_.reshape(-1).copy(), "var_fixed": fixed.reshape(-1), "expr_tag": tag.reshape(-1), "expr_idx": idx.reshape(-1),
            "expr_param": par.reshape(-1), "var_comp": np.zeros(n_sys * nv, dtype=np.uint16), "expr_comp": np.zeros(n_sys * m, dtype=np.uint16)}


@pytest.mark.parametrize("seed", range(8))
def test_random_structures_against_the_general_paths(fiksi, ctx, seed):
    """Random connected sketches of 25 ... 60 points (a spanning tree of distances, extra distances, a few angles, sometimes a fixed
    point): whatever the minimum-degree order, the fill and the level schedule come out as, the sparse build gives the counters
    of the paths it replaces and their positions to round-off — and batches it must refuse (a factor past 1 023 entries) take
    those paths."""
    from helpers import Lcg

    g = Lcg(900 + 7919 * seed)
    n_pts = (25, 31, 38, 44, 50, 55, 60, 28)[seed]
    b = _random_graph_batch(40, n_pts, (0, 3, 9, 14, 20, 6, 25, 12)[seed], 77 + seed, fix_first=bool(seed & 1), angles=int(g.u(0, 6.99)))
    db = ctx.upload(b)
    build = db.grouped_build()
    db.free()
    v1, r1 = ctx.system_solve_batch(b)
    ctx.set_one_structure_builds(False)
    try:
        v0, r0 = ctx.system_solve_batch(b)
    finally:
        ctx.set_one_structure_builds(True)
    if build == 2:
        same = (r0["accepted"] == r1["accepted"]) & (r0["trials"] == r1["trials"]) & (r0["exit"] == r1["exit"])
        assert same.mean() >= 0.9, (n_pts, same.mean())  # (two elimination orders: a trial on the edge of acceptance may fall either way)
        ok = same & (r0["sse_unscaled"] < 1e-6)
        if ok.any():
            nv = int(b["var_off"][1])
            d = np.abs(v0 - v1).reshape(-1, nv)[ok]
            assert d.max() < 1e-7, d.max()
        assert np.array_equal(r0["scale"], r1["scale"]) and np.allclose(r0["sse0"], r1["sse0"], rtol=1e-12, atol=0)
    else:  # not taken: the very same path either way
        assert np.array_equal(_bits(v0), _bits(v1)) and r0.tobytes() == r1.tobytes()


def test_without_perturbation_and_through_the_lm_entry_point(fiksi, ctx):
    """`perturb = False` (no LCG draws) and fx_lm_solve_batch (no scaling, no perturbation: the bare Levenberg-Marquardt of
    lm.rs on the batch's values) take the sparse build too: the counters of the general paths, positions to round-off."""
    from fiksi_amd import abi, workloads

    b = workloads.hinged_triangles(200, 16)
    for call in ("system_no_perturb", "lm"):
        outs = []
        for enable in (True, False):
            ctx.set_one_structure_builds(enable)
            try:
                if call == "lm":
                    outs.append(ctx.lm_solve_batch(b))
                else:
                    outs.append(ctx.system_solve_batch(b, abi.solving_opts(perturb=False)))
            finally:
                ctx.set_one_structure_builds(True)
        (v1, r1), (v0, r0) = outs
        for f in ("accepted", "trials", "exit"):
            assert np.array_equal(r0[f], r1[f]), (call, f)
        assert np.max(np.abs(v0 - v1)) < 1e-9, call
