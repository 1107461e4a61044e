"""GPU edge cases: maximum one-wavefront sizes, degenerate geometry (NaN / Inf paths the reference would
spin on forever), ragged batches mixing every kernel instantiation, large batches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _star(n_points, fixed_center=True):
    """One centre + n_points-1 satellites at given distances: 2*(n-1) free variables when the centre is fixed."""
    from fiksi_amd import System, constraints, elements

    s = System()
    c = elements.Point.create(s, 0.3, -0.2)
    if fixed_center:
        c.fix(s)
    for k in range(n_points - 1):
        a = 2.0 * np.pi * k / (n_points - 1)
        p = elements.Point.create(s, 2.0 * np.cos(a) + 0.1, 2.0 * np.sin(a) - 0.1)
        constraints.PointPointDistance.create(s, c, p, 1.0 + 0.01 * k)
    return s


def test_exactly_64_free_variables_uses_the_fused_kernel(fiksi, oracle, ctx):
    s = _star(33)  # 32 satellites = 64 free variables, 32 rows: the one-wavefront maximum
    b = s.flatten()
    assert int((b["var_fixed"] == 0).sum()) == 64
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3)
    assert res["accepted"][0] == res_o["accepted"][0] and res["exit"][0] == res_o["exit"][0]
    assert np.max(np.abs(v - v_o)) < 1e-8
    s2 = _star(34)  # 66 free variables: sparse path, same answer quality
    b2 = s2.flatten()
    v2, res2 = ctx.system_solve_batch(b2)
    v2_o, res2_o = oracle.solve_batch(b2, mode=3)
    assert res2["accepted"][0] == res2_o["accepted"][0]
    assert np.max(np.abs(v2 - v2_o)) < 1e-8


def test_many_rows_few_columns(fiksi, oracle, ctx):
    """256 expressions on 8 free variables (heavily over-determined, consistent): the row loop of the
    fused kernel runs 4 chunks of 64; 257 rows go through the sparse path."""
    from fiksi_amd import System, constraints, elements

    for n_rows in (256, 257):
        s = System()
        pts = [elements.Point.create(s, float(i) + 0.05 * i * i, 0.3 * i) for i in range(4)]
        target = [(0., 0.), (1., 0.2), (2.1, 0.9), (2.9, 2.2)]
        pairs = [(a, c) for a in range(4) for c in range(a + 1, 4)]
        for k in range(n_rows):
            a, c = pairs[k % len(pairs)]
            d = float(np.hypot(target[a][0] - target[c][0], target[a][1] - target[c][1]))
            constraints.PointPointDistance.create(s, pts[a], pts[c], d)
        b = s.flatten()
        v, res = ctx.system_solve_batch(b)
        v_o, res_o = oracle.solve_batch(b, mode=3)
        assert res["accepted"][0] == res_o["accepted"][0], n_rows
        assert res["exit"][0] == res_o["exit"][0] == 0
        assert abs(res["sse"][0] - res_o["sse"][0]) <= 1e-10


def test_every_padded_size_in_one_batch(fiksi, oracle, ctx):
    """Stars with 2, 6, 10, ... free variables: the batch is solved by the N=64 instantiation; each
    smaller batch by its own (8, 16, ..., 64)."""
    from fiksi_amd import flatten

    systems = [_star(n) for n in (2, 4, 6, 9, 13, 17, 21, 25, 29, 33)]
    b = flatten(systems)
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3)
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.max(np.abs(v - v_o)) < 1e-8
    for k, s in enumerate(systems):  # one system at a time -> each template instantiation
        bs = s.flatten()
        vs, rs = ctx.system_solve_batch(bs)
        v0, v1 = int(b["var_off"][k]), int(b["var_off"][k + 1])
        assert np.max(np.abs(vs - v_o[v0:v1])) < 1e-8, k


def test_degenerate_geometry_terminates(fiksi, ctx):
    """Coincident points under a distance constraint: d = 0 makes the gradient Inf/NaN
    (expressions.rs:343-349, "singular at d = 0 by design"). The reference's inner loop has no cap
    (quirk Q8) and would spin on NaN; the device path must end and say why."""
    from fiksi_amd import System, abi, constraints, elements

    s = System()
    a = elements.Point.create(s, 1.0, 1.0)
    c = elements.Point.create(s, 1.0, 1.0)
    constraints.PointPointDistance.create(s, a, c, 2.0)
    b = s.flatten()
    v, res = ctx.system_solve_batch(b, abi.solving_opts(perturb=False))
    assert res["exit"][0] in (abi.EXIT_NAN, abi.EXIT_TRIAL_CAP)
    assert np.array_equal(v, b["vars"])  # nothing accepted: the start values come back untouched
    # with the default perturbation the same sketch is solvable (that is what perturb is for)
    v, res = ctx.system_solve_batch(b)
    assert res["exit"][0] == abi.EXIT_SSE and res["sse_unscaled"][0] < 1e-8
    # zero-length line under tangency: residual and gradient are defined as 0 (quirk Q5), LM stops at once
    s = System()
    p = elements.Point.create(s, 0.0, 0.0)
    q = elements.Point.create(s, 0.0, 0.0)
    ctr = elements.Point.create(s, 1.0, 1.0)
    rad = elements.Length.create(s, 0.5)
    ln = elements.Line.create(s, p, q)
    ci = elements.Circle.create(s, ctr, rad)
    constraints.LineCircleTangency.create(s, ln, ci)
    v, res = ctx.system_solve_batch(s.flatten(), abi.solving_opts(perturb=False))
    assert res["exit"][0] == abi.EXIT_SSE and res["accepted"][0] == 0


def test_non_finite_inputs_do_not_hang(fiksi, ctx):
    from fiksi_amd import abi, workloads

    b = workloads.ring16(8)
    b["vars"] = b["vars"].copy()
    b["vars"][5] = np.nan
    b["vars"][40] = np.inf
    v, res = ctx.system_solve_batch(b)
    assert res["exit"][0] == abi.EXIT_NAN and res["exit"][1] == abi.EXIT_NAN
    assert np.all(res["exit"][2:] != abi.EXIT_NAN)


def test_one_million_systems_in_one_batch(fiksi, ctx):
    """cfg5 scale (1M sketches; here on one GPU, f32): offsets stay within u32, results are sane, and a
    checksum of per-system results is reproducible across two runs."""
    from fiksi_amd import abi, workloads

    n = 1_000_000
    b = workloads.ring16(n, inconsistent=True)
    db = ctx.upload(b)
    o = abi.solving_opts(f32=True)
    db.system_solve(o)
    r1 = db.get_results()
    db.system_solve(o)
    r2 = db.get_results()
    db.free()
    assert np.array_equal(r1, r2)
    assert np.isin(r1["exit"], (1, 2)).mean() > 0.99
    assert r1["accepted"].sum() > 4 * n


def test_fully_fixed_and_unconstrained_parts(fiksi, oracle, ctx):
    """A component whose variables are all fixed (zero free columns, constraints still present), a
    constraint between two fixed points inside a free component, and free points no constraint touches:
    nothing moves that must not move, counts and exits equal the oracle's."""
    F = fiksi
    s = F.System()
    a = F.elements.Point.create(s, 0., 0.)
    b = F.elements.Point.create(s, 3., 4.)
    F.constraints.PointPointDistance.create(s, a, b, 7.)   # infeasible, both ends fixed
    a.fix(s)
    b.fix(s)
    c = F.elements.Point.create(s, 1., 1.)
    d = F.elements.Point.create(s, 2., 3.)
    e = F.elements.Point.create(s, 5., 1.)
    F.constraints.PointPointDistance.create(s, c, d, 2.)
    F.constraints.PointPointDistance.create(s, d, e, 2.)
    F.constraints.PointPointDistance.create(s, c, e, 3.)
    c.fix(s)
    e.fix(s)
    F.constraints.PointPointDistance.create(s, c, e, 9.)   # fixed-fixed row inside a free component
    lonely = F.elements.Point.create(s, -4., 2.5)          # no constraint: not part of any component's rows
    flat = s.flatten()
    v, res = ctx.system_solve_batch(flat)
    v_o, res_o = oracle.solve_batch(flat, mode=3, trial_cap=4096)
    assert res["ncomp"][0] == res_o["ncomp"][0]
    assert res["accepted"][0] == res_o["accepted"][0] and res["exit"][0] == res_o["exit"][0]
    fx = flat["var_fixed"] == 1
    assert np.array_equal(v[fx], flat["vars"][fx])
    assert np.max(np.abs(v - v_o)) < 1e-8
    # the lonely point is perturbed-and-written-back or left alone exactly as the oracle does
    assert np.array_equal(v[-2:], v_o[-2:])
    for opts in (F.abi.solving_opts(decomposer=1), F.abi.solving_opts(optimizer=1)):
        v2, res2 = ctx.system_solve_batch(flat, opts)
        assert np.array_equal(v2[fx], flat["vars"][fx]) and np.all(np.isfinite(v2))


def test_resident_batch_with_new_parameters_and_start_values(fiksi, oracle, ctx):
    """fx_batch_set_params / fx_batch_set_vars on a resident batch == uploading the changed batch anew
    (the dragging-a-dimension workflow: same structure, new targets, warm start from the last solution)."""
    from fiksi_amd import workloads

    b = workloads.concat([workloads.ring16(200), workloads.hinged_triangles(3, 20), workloads.large_sketch(80)])
    db = ctx.upload(b)
    db.system_solve()
    v1 = db.get_vars()
    b2 = dict(b)
    b2["expr_param"] = b["expr_param"] * np.where(np.isin(b["expr_tag"], (1, 4)), 1.03, 1.0)  # distances +3 %
    b2["vars"] = v1.copy()
    db.set_params(b2["expr_param"])
    db.set_vars(v1)
    db.system_solve()
    v2, r2 = db.get_vars(), db.get_results()
    v_ref, r_ref = ctx.system_solve_batch(b2)
    assert np.array_equal(v2, v_ref) and np.array_equal(r2, r_ref)
    v_o, r_o = oracle.solve_batch(b2, mode=3, nthreads=4)
    assert np.mean(r2["accepted"] == r_o["accepted"]) > 0.95
    db.free()


def test_large_system_of_many_small_components_is_walked_on_the_device(fiksi, oracle, ctx):
    """One System of 150 separate features (600 points in triangles and quadrilaterals: 1 200 variables, far
    beyond LDS) under Decomposer::None: every component fits a wavefront, so one wavefront walks them in
    order — shared LCG stream across components, snapshot semantics of quirk Q2 — instead of 150 trips
    through the host-driven sparse path. Same answer as the oracle."""
    F = fiksi
    import time

    s = F.System()
    g = 0.0
    for k in range(150):
        n = 3 + (k % 2)
        pts = [F.elements.Point.create(s, 10.0 * k + 1.3 * i + 0.01 * ((7 * k + i) % 5), 0.9 * ((i * i + k) % 3)) for i in range(n)]
        for i in range(n):
            F.constraints.PointPointDistance.create(s, pts[i], pts[(i + 1) % n], 1.5 + 0.1 * (k % 4))
        if n == 4:
            F.constraints.PointPointDistance.create(s, pts[0], pts[2], 2.2)
        if k % 9 == 0:
            pts[0].fix(s)
    flat = s.flatten()
    assert len(flat["vars"]) > 1000 and int(flat["var_comp"].max()) == 149
    t = time.time()
    v, res = ctx.system_solve_batch(flat)
    dt = time.time() - t
    v_o, res_o = oracle.solve_batch(flat, mode=3, trial_cap=4096)
    assert res["ncomp"][0] == res_o["ncomp"][0] == 150
    assert res["accepted"][0] == res_o["accepted"][0] and res["trials"][0] == res_o["trials"][0]
    assert np.max(np.abs(v - v_o)) < 1e-8
    fx = flat["var_fixed"] == 1
    assert np.array_equal(v[fx], flat["vars"][fx])
    assert dt < 0.05  # one launch, not 150 host-driven solves
    # mixed with other size classes in one batch, and repeatable on a resident batch
    from fiksi_amd import workloads
    b = workloads.concat([workloads.ring16(3), flat, workloads.hinged_triangles(1, 20), workloads.large_sketch(60)])
    v2, res2 = ctx.system_solve_batch(b)
    assert np.array_equal(v2[96:96 + len(v)], v)
    db = ctx.upload(b)
    db.system_solve()
    db.system_solve()
    assert np.array_equal(db.get_vars(), v2)
    db.free()


def test_k1_with_hundreds_of_constraint_free_systems_between_rows(fiksi, oracle, ctx):
    """Systems without expressions take a System index without taking a row, so one 256-row block of the
    Jacobian-assembly kernel can span more than 256 Systems: the per-row System offset (one byte) must not wrap —
    such blocks read their System's first variable from the per-row table instead."""
    from fiksi_amd import workloads

    def empty(n):  # n Systems of one unconstrained point each
        return {"var_off": (2 * np.arange(n + 1)).astype(np.uint32), "expr_off": np.zeros(n + 1, dtype=np.uint32),
                "vars": np.arange(2.0 * n), "var_fixed": np.zeros(2 * n, dtype=np.uint8), "expr_tag": np.zeros(0, dtype=np.uint8),
                "expr_idx": np.zeros(0, dtype=np.uint32), "expr_param": np.zeros(0), "var_comp": np.full(2 * n, 0xFFFF, dtype=np.uint16),
                "expr_comp": np.zeros(0, dtype=np.uint16)}

    b = workloads.concat([workloads.ring16(3), empty(300), workloads.ring16(2, seed0=50), empty(700), workloads.hinged_triangles(2, 4),
                          empty(1), workloads.ring16(9, seed0=60)])
    r, (rp, ci, vals) = ctx.eval_residual_jacobian(b)
    r_o, (rp_o, ci_o, vals_o) = oracle.eval_batch(b)
    assert np.array_equal(rp.astype(np.int64), rp_o) and np.array_equal(ci.astype(np.int32), ci_o)
    assert np.array_equal(vals, vals_o)
    assert np.allclose(r, r_o, rtol=0, atol=4e-15)
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3)
    assert np.array_equal(res["accepted"], res_o["accepted"]) and np.array_equal(res["ncomp"], res_o["ncomp"])


def test_one_shot_calls_reuse_the_plan_of_a_structure_seen_before(fiksi, ctx):
    """fx_system_solve_batch on a large sketch, again and again (System::solve while a sketch is dragged): the context
    keeps the sparse path's plan per structure and solve mode (eight of them, least recently used dropped). Same bits as
    the first call and as a resident batch; new values, a changed structure and more structures than the cache holds all
    go through."""
    from fiksi_amd import abi

    from helpers import random_big_sketch

    flats = [random_big_sketch(9000 + k, 140 + 7 * k).flatten() for k in range(10)]  # ten structures > eight cached plans
    first = []
    for f in flats:
        first.append(ctx.system_solve_batch(f))
    for round_ in range(2):
        for k, f in enumerate(flats):
            v, r = ctx.system_solve_batch(f)
            assert np.array_equal(v.view(np.uint64), first[k][0].view(np.uint64)) and r.tobytes() == first[k][1].tobytes()
    # resident batch of the same sketch: the same bits
    db = ctx.upload(flats[3])
    db.system_solve()
    assert np.array_equal(db.get_vars().view(np.uint64), first[3][0].view(np.uint64))
    db.free()
    # same structure, other values: the cached plan serves it
    g = dict(flats[5])
    g["vars"] = g["vars"] * 1.25 + 0.5
    v1, r1 = ctx.system_solve_batch(g)
    db = ctx.upload(g)
    db.system_solve()
    assert np.array_equal(db.get_vars().view(np.uint64), v1.view(np.uint64))
    db.free()
    # another solve mode on a cached structure gets its own plan
    v2, r2 = ctx.system_solve_batch(flats[5], abi.solving_opts(solver=1))
    v3, r3 = ctx.system_solve_batch(flats[5], abi.solving_opts(solver=1))
    assert np.array_equal(v2.view(np.uint64), v3.view(np.uint64)) and r2.tobytes() == r3.tobytes()


def test_a_big_host_batch_solved_in_chunks_is_the_plain_solve(fiksi, ctx):
    """fx_system_solve_batch on 65 536 one-wavefront Systems and more is analysed once, then copied up and solved in two
    chunks (the second goes up while the first is solved; copy stream + two solve streams); a resident batch never chunks.
    Same bits in every variable and result field; mixed structures, a fixed gauge in some; also under SinglePass."""
    from fiksi_amd import abi, workloads

    parts = []
    for k in range(24):
        parts += [workloads.ring16(3000, seed0=100 + 5000 * k), workloads.hinged_triangles(700, 5), workloads.ring16(300, seed0=9 + k, fix_gauge=True)]
    b = workloads.concat(parts)
    assert len(b["var_off"]) - 1 == 96000
    for opts in (None, abi.solving_opts(decomposer=1)):
        v, res = ctx.system_solve_batch(b, opts)      # chunked
        db = ctx.upload(b)                              # resident: one block, one launch sequence
        db.system_solve(opts)
        v0, res0 = db.get_vars(), db.get_results()
        db.free()
        assert np.array_equal(v.view(np.uint64), v0.view(np.uint64))
        for f in res.dtype.names:
            x, y = res[f], res0[f]
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)) if x.dtype.kind == "f" else np.array_equal(x, y), f
