"""The grouped kernel (fx_grouped.hip: four Systems per wavefront, components of at most 32 free variables)
against the oracle and against the one-System-per-wavefront kernel it replaces for large batches.
The context's routing option (fx_ctx_set_routing): 1 sends every qualifying batch to the grouped kernel, 0 none;
by default (-1) batches of 8 Systems and more take it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def routing(ctx):
    def set_(v):
        ctx.set_routing(-1 if v is None else int(v))

    yield set_
    ctx.set_routing(-1)


def _solve(ctx, b, **kw):
    from fiksi_amd import abi

    db = ctx.upload(b)
    db.system_solve(abi.solving_opts(**kw))
    v, r = db.get_vars().copy(), db.get_results().copy()
    db.free()
    return v, r


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("kw", [dict(), dict(fix_gauge=True), dict(inconsistent=True)])
@pytest.mark.parametrize("f32", [False, True])
def test_bit_identical_to_the_one_system_per_wavefront_kernel(fiksi, ctx, routing, kw, f32):
    """Same operations on the same operands in the same order (Cholesky, triangular solves, the sums of the LM
    control): on the headline shape and its variants every solved variable and every result field carries the
    bits the one-System-per-wavefront kernel produces. 4099 Systems: the last wavefront is partly filled."""
    from fiksi_amd import workloads

    b = workloads.ring16(4099, **kw)
    routing("1")
    v1, r1 = _solve(ctx, b, f32=f32)
    routing("0")
    v0, r0 = _solve(ctx, b, f32=f32)
    assert np.array_equal(_bits(v1), _bits(v0))
    for f in r1.dtype.names:
        a, c = r1[f], r0[f]
        assert np.array_equal(_bits(a), _bits(c)) if a.dtype.kind == "f" else np.array_equal(a, c), f


@pytest.mark.parametrize("n_tri", [8, 10, 11])
def test_forty_eight_column_build_on_the_references_bench_sketch(fiksi, oracle, ctx, routing, n_tri):
    """Chains of hinged triangles (`add_hinged_triangles`, fiksi_bench.rs:15-40; n = 11: 46 variables, 33
    distances): 33 .. 48 free variables take three matrix columns per lane. Bit-identical to the
    one-System-per-wavefront kernel, same LM path as the oracle."""
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(1030, n_tri)
    assert 32 < int(b["var_off"][1]) <= 48
    routing("1")
    ctx.set_one_structure_builds(False)  # (the general build: a batch of one structure with so sparse a factor takes fx_grouped_s.hip)
    try:
        v1, r1 = _solve(ctx, b)
    finally:
        ctx.set_one_structure_builds(True)
    routing("0")
    v0, r0 = _solve(ctx, b)
    assert np.array_equal(_bits(v1), _bits(v0))
    for f in r1.dtype.names:
        a, c = r1[f], r0[f]
        assert np.array_equal(_bits(a), _bits(c)) if a.dtype.kind == "f" else np.array_equal(a, c), f
    sub = workloads.shard(b, 0, 16)
    n = len(sub["var_off"]) - 1
    v_o, res_o = oracle.solve_batch(sub, mode=3, nthreads=8)
    assert np.array_equal(r1["accepted"][:n], res_o["accepted"])
    assert np.array_equal(r1["trials"][:n], res_o["trials"])
    assert np.max(np.abs(v1[: len(v_o)] - v_o)) < 1e-8
    # f32 (cfg5 precision): the same two kernels, the same bits
    routing("1")
    w1, q1 = _solve(ctx, b, f32=True)
    routing("0")
    w0, q0 = _solve(ctx, b, f32=True)
    assert np.array_equal(_bits(w1), _bits(w0)) and np.array_equal(q1["trials"], q0["trials"])


@pytest.mark.parametrize("shape", ["hinged11", "hinged5", "ring16", "mixed"])
def test_single_pass_blocks_on_the_grouped_kernel(fiksi, oracle, ctx, routing, shape):
    """`Decomposer::SinglePass`: the grouped kernel walks the blocks of the host decomposition as lm_solve_kernel's
    UNITS build does (perturbation before a component's first block, write-through after each block) — same bits,
    and the oracle's solution on a sample."""
    import helpers
    from fiksi_amd import abi, workloads

    b = {"hinged11": lambda: workloads.hinged_triangles(1030, 11), "hinged5": lambda: workloads.hinged_triangles(1030, 5),
         "ring16": lambda: workloads.ring16(1030),
         "mixed": lambda: workloads.concat([helpers.mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(150)]
                                           + [helpers.random_sketch(s).flatten() for s in range(150)])}[shape]()
    o = abi.solving_opts(decomposer=1)
    out = {}
    for tag in ("1", "0"):
        routing(tag)
        db = ctx.upload(b)
        db.system_solve(o)
        assert db.solve_route(o) == int(tag)
        out[tag] = (db.get_vars().copy(), db.get_results().copy())
        db.free()
    (v1, r1), (v0, r0) = out["1"], out["0"]
    if shape != "mixed":
        assert np.array_equal(_bits(v1), _bits(v0))
        for f in r1.dtype.names:
            a, c = r1[f], r0[f]
            assert np.array_equal(_bits(a), _bits(c)) if a.dtype.kind == "f" else np.array_equal(a, c), f
        sub = workloads.shard(b, 0, 16)
        n = len(sub["var_off"]) - 1
        v_o, res_o = oracle.solve_single_pass_batch(sub, trial_cap=4096, nthreads=8)
        assert np.array_equal(r1["accepted"][:n], res_o["accepted"])
        assert np.max(np.abs(v1[: len(v_o)] - v_o)) < 1e-7
        o32 = abi.solving_opts(decomposer=1, f32=True)
        w = {}
        for tag in ("1", "0"):
            routing(tag)
            db = ctx.upload(b)
            db.system_solve(o32)
            assert db.solve_route(o32) == int(tag)
            w[tag] = (db.get_vars().copy(), db.get_results().copy())
            db.free()
        assert np.array_equal(_bits(w["1"][0]), _bits(w["0"][0])) and np.array_equal(w["1"][1]["trials"], w["0"][1]["trials"])
    else:
        assert np.array_equal(r1["scale"], r0["scale"]) and np.array_equal(r1["ncomp"], r0["ncomp"])
        same = (r1["accepted"] == r0["accepted"]) & (r1["trials"] == r0["trials"])
        assert same.mean() > 0.9
        ok = ~(np.isnan(r1["sse"]) | np.isnan(r0["sse"]))
        d = np.abs(r1["sse"] - r0["sse"])[same & ok]  # tests/test_gpu_fuzz.py's bound for these ill-conditioned sketches
        assert np.all(d <= 1e-9 + 0.25 * np.abs(r0["sse"][same & ok]))


@pytest.mark.parametrize("build", [0, 1])
def test_against_the_oracle_on_the_headline_shape(fiksi, oracle, ctx, routing, build):
    """The comparison tests/test_gpu_parity.py makes for the one-System-per-wavefront kernel, on the grouped one — on its
    general build (fx_grouped.hip, build 0) and on the build for batches of one structure the headline batch takes
    (fx_grouped_c.hip, build 1), each asserted BY NAME through fx_debug_grouped_build before the solve that is compared."""
    from fiksi_amd import workloads
    from test_gpu_parity import _compare_solves

    b = workloads.ring16(2048)
    routing("1")
    ctx.set_one_structure_builds(bool(build))
    try:
        db = ctx.upload(b)
        assert db.grouped_build() == build
        db.system_solve()
        v, res = db.get_vars().copy(), db.get_results().copy()
        db.free()
    finally:
        ctx.set_one_structure_builds(True)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    assert np.array_equal(res["scale"], res_o["scale"])  # K0: bit-identical
    _compare_solves(res, res_o, v, v_o, b, oracle)
    r = oracle.residuals_batch(b, v).reshape(len(res), -1)
    assert np.allclose(res["sse_unscaled"], (r * r).sum(1), rtol=1e-9, atol=1e-18)


@pytest.mark.parametrize("f32", [False, True])
def test_mixed_kinds_fixed_variables_and_several_components(fiksi, oracle, ctx, routing, f32):
    """All eleven constraint kinds, fixed variables, Systems of different sizes and component counts in one batch
    (no shared structure: every row builds its lists per System), against the oracle with the tolerances of
    tests/test_gpu_parity.py; the default routing gives the bits of "1" for this batch and of "0" for its first 7 Systems."""
    import helpers
    from fiksi_amd import workloads

    flats = [helpers.mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(120)]
    flats += [helpers.random_sketch(s).flatten() for s in range(200)]
    flats += [workloads.hinged_triangles(1, 3), workloads.quadrilateral(), workloads.hinged_triangles(1, 7)]
    b = workloads.concat(flats)
    routing("1")
    v, res = _solve(ctx, b, f32=f32)
    routing("0")
    v0, res0 = _solve(ctx, b, f32=f32)
    routing(None)
    vd, resd = _solve(ctx, b, f32=f32)
    assert np.array_equal(_bits(vd), _bits(v))  # 323 Systems: from 8 Systems on the default is the grouped kernel (round 4)
    small = workloads.concat(flats[:7])
    vs, _ = _solve(ctx, small, f32=f32)
    assert np.array_equal(_bits(vs), _bits(v0[: len(vs)]))  # 7 Systems: below the threshold, one System per wavefront
    # the two kernels add the products of an entry of JtJ in a different lane order: same path, last-bit sums
    assert np.array_equal(res["scale"], res0["scale"])
    assert np.array_equal(res["ncomp"], res0["ncomp"])
    # (in f32 the last-bit differences flip the exit of a few more of the ill-conditioned random sketches)
    assert (res["exit"] == res0["exit"]).mean() > (0.9 if f32 else 0.97)
    if not f32:
        # the criteria of tests/test_gpu_fuzz.py (random sketches are often ill-conditioned or degenerate; uncapped,
        # the reference loops forever on some of them)
        from helpers import compare_outcomes

        v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
        same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=False)
        assert same > 0.6 and verdict > 0.85, (same, verdict)


def test_large_batch_takes_the_grouped_kernel_by_default_and_is_deterministic(fiksi, ctx, routing):
    from fiksi_amd import workloads

    b = workloads.ring16(10000)
    routing(None)
    v_a, r_a = _solve(ctx, b)
    v_b, r_b = _solve(ctx, b)
    assert np.array_equal(_bits(v_a), _bits(v_b))
    routing("0")
    v_0, r_0 = _solve(ctx, b)
    assert np.array_equal(_bits(v_a), _bits(v_0))
    assert np.array_equal(r_a["trials"], r_0["trials"])


def test_shared_structure_is_detected_per_batch_not_assumed(fiksi, ctx, routing):
    """A batch of identical sketches keeps its row lists between Systems; one odd System in the batch (a fixed
    point) switches that off for the whole batch. Both give the bits of the one-System-per-wavefront kernel."""
    from fiksi_amd import workloads

    u = workloads.ring16(600, fix_gauge=False)
    odd = workloads.ring16(1, fix_gauge=True)
    b = workloads.concat([workloads.shard(u, 0, 2), odd, workloads.shard(u, 1, 2)])
    for batch in (u, b):
        routing("1")
        v1, r1 = _solve(ctx, batch)
        routing("0")
        v0, r0 = _solve(ctx, batch)
        assert np.array_equal(_bits(v1), _bits(v0))
        assert np.array_equal(r1["trials"], r0["trials"])


def test_structure_classes_of_a_batch_of_several_sketches(fiksi, ctx, routing):
    """Batches of 8192 Systems and more that are not of one structure get a structure class per System (the first
    System with the same sizes, flags, kinds and fields); a row keeps its lists while the class stays the same.
    Three sketches interleaved in runs of different lengths: the bits of the one-System-per-wavefront kernel."""
    from fiksi_amd import workloads

    a, g, h = workloads.ring16(7000), workloads.ring16(900, fix_gauge=True), workloads.hinged_triangles(500, 5)
    parts = []
    for k in range(100):
        parts += [workloads.shard(a, k, 100), workloads.shard(g, k, 100), workloads.shard(h, k, 100)]
    b = workloads.concat(parts)
    assert len(b["var_off"]) - 1 == 8400
    routing(None)  # the default routing takes it
    db = ctx.upload(b)
    assert db.solve_route() == 1
    db.free()
    v1, r1 = _solve(ctx, b)
    routing("0")
    v0, r0 = _solve(ctx, b)
    assert np.array_equal(_bits(v1), _bits(v0))
    for f in r1.dtype.names:
        x, y = r1[f], r0[f]
        assert np.array_equal(_bits(x), _bits(y)) if x.dtype.kind == "f" else np.array_equal(x, y), f


def _anchored_sketch(seed):
    """20 fixed anchor points and 10 free ones tied to them (and three pairs to each other): 60 variables — past
    the 48 a lane row keeps in registers —, 20 free ones, 28 expressions, fixed values that must come back untouched."""
    import helpers
    from fiksi_amd import System, constraints, elements

    g = helpers.Lcg(seed)
    s = System()
    anchors = [elements.Point.create(s, g.u(-20, 20), g.u(-20, 20)) for _ in range(20)]
    for a in anchors:
        a.fix(s)
    free = [elements.Point.create(s, g.u(-20, 20), g.u(-20, 20)) for _ in range(10)]
    for i, p in enumerate(free):
        for k in range(2):
            constraints.PointPointDistance.create(s, p, anchors[(2 * i + 7 * k + seed) % 20], g.u(5, 25))
    for i in range(0, 6, 2):
        constraints.PointPointDistance.create(s, free[i], free[i + 1], g.u(3, 12))
    for i in range(5):
        constraints.PointPointPointAngle.create(s, anchors[i], free[i], anchors[i + 10], g.u(-1.5, 1.5))
    return s


def _angle_sketch(seed):
    """16 free points under 24 three-point angles (six entries per row, all free): 504 products in the triangle of
    JtJ — more than the 448 list words the f64 build keeps in registers — and fewer rows than variables."""
    import helpers
    from fiksi_amd import System, constraints, elements

    g = helpers.Lcg(seed)
    s = System()
    P = [elements.Point.create(s, 10 * np.cos(0.39 * i) + g.u(-1, 1), 10 * np.sin(0.39 * i) + g.u(-1, 1)) for i in range(16)]
    for k in range(24):
        a, b2, c = (2 * k) % 16, (2 * k + 3 + k // 8) % 16, (2 * k + 7 + k // 8) % 16
        constraints.PointPointPointAngle.create(s, P[a], P[b2], P[c], g.u(-1.5, 1.5))
    return s


def _overdetermined_sketch(seed):
    """8 free points under 70 inconsistent distances: 16 columns, 70 rows (more than the 64 one pass of a wavefront
    covers, and more expressions than a lane row keeps in registers): a least-squares fit."""
    import helpers
    from fiksi_amd import System, constraints, elements

    g = helpers.Lcg(seed)
    s = System()
    P = [elements.Point.create(s, g.u(-10, 10), g.u(-10, 10)) for _ in range(8)]
    k = 0
    while k < 70:
        a, b2 = int(g.u(0, 8)) % 8, int(g.u(0, 8)) % 8
        if a != b2:
            constraints.PointPointDistance.create(s, P[a], P[b2], g.u(2, 12))
            k += 1
    return s


@pytest.mark.parametrize("builder,f32", [("anchored", False), ("angles", False), ("overdetermined", True)])
def test_sketches_past_the_register_held_parts(fiksi, oracle, ctx, routing, builder, f32):
    """Variables / expressions past the ones a lane row keeps in registers (loaded where they are used), product
    lists longer than their register-held part, more than 64 rows — each within the 10 KB of LDS per System the
    routing allows: against the one-System-per-wavefront kernel and (f64) the oracle."""
    from fiksi_amd import abi, workloads

    make = {"anchored": _anchored_sketch, "angles": _angle_sketch, "overdetermined": _overdetermined_sketch}[builder]
    b = workloads.concat([make(s).flatten() for s in range(40)])
    routing("1")
    db = ctx.upload(b)
    assert db.solve_route(abi.solving_opts(f32=f32)) == 1
    db.free()
    v1, r1 = _solve(ctx, b, f32=f32)
    routing("0")
    v0, r0 = _solve(ctx, b, f32=f32)
    assert np.array_equal(r1["scale"], r0["scale"]) and np.array_equal(r1["ncomp"], r0["ncomp"])
    fx_ = b["var_fixed"] == 1
    assert np.array_equal(v1[fx_], b["vars"][fx_])
    same = (r1["accepted"] == r0["accepted"]) & (r1["trials"] == r0["trials"])
    assert same.mean() > 0.85
    rtol = 1e-3 if f32 else 1e-6
    assert np.allclose(r1["sse"][same], r0["sse"][same], rtol=rtol, atol=1e-12)
    if not f32:
        v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
        assert np.array_equal(r1["scale"], res_o["scale"]) and np.array_equal(r1["ncomp"], res_o["ncomp"])
        same_o = (r1["accepted"] == res_o["accepted"]) & (r1["trials"] == res_o["trials"])
        assert same_o.mean() > 0.7
        d = np.abs(r1["sse"] - res_o["sse"])
        assert np.all(d[same_o] <= 1e-9 + 1e-4 * np.abs(res_o["sse"][same_o]))


def test_history_schedule_changes_the_order_not_the_results(fiksi, ctx):
    """fx_batch_schedule_by_last_solve: the Systems that took the most trials last time start first; every result is
    bit-identical to the index-order solve (f64 and f32, and when the data changes between solves)."""
    from fiksi_amd import abi, workloads

    b = workloads.ring16(20000, inconsistent=True)
    db = ctx.upload(b)
    for f32 in (False, True):
        o = abi.solving_opts(f32=f32)
        db.schedule_by_last_solve(False)
        db.system_solve(o)
        v0, r0 = db.get_vars().copy(), db.get_results().copy()
        db.schedule_by_last_solve(True)
        db.system_solve(o)
        v1, r1 = db.get_vars().copy(), db.get_results().copy()
        assert np.array_equal(_bits(v0), _bits(v1)) and r0.tobytes() == r1.tobytes()
    # new targets: the old schedule is only a guess now, the results are still those of a fresh solve
    p2 = b["expr_param"] * 1.01
    db.set_params(p2)
    db.system_solve(abi.solving_opts())
    v2, r2 = db.get_vars().copy(), db.get_results().copy()
    db.free()
    b2 = dict(b, expr_param=p2)
    v3, r3 = ctx.system_solve_batch(b2)
    assert np.array_equal(_bits(v2), _bits(v3)) and np.array_equal(r2["trials"], r3["trials"])


def test_presort_changes_the_order_not_the_results(fiksi, ctx):
    """fx_ctx_set_presort: the scout pass + radix sort hand big batches out most-work-first; every System's result is
    the same bits as in index order (uniform and mixed-structure batches, f64 and f32, SinglePass)."""
    from fiksi_amd import abi, workloads

    from helpers import random_sketch

    mixed = workloads.concat([workloads.ring16(6000), workloads.concat([random_sketch(s).flatten() for s in range(300)]),
                              workloads.hinged_triangles(3000, 4), workloads.ring16(3000, inconsistent=True, seed0=9)])
    for b, kw in ((workloads.ring16(20000), {}), (workloads.ring16(12000, inconsistent=True), {"f32": True}), (mixed, {}),
                  (workloads.hinged_triangles(9000, 11), {"decomposer": 1})):
        out = []
        for on in (False, True):
            ctx.set_presort(on, 8192)
            try:
                out.append(ctx.system_solve_batch(b, abi.solving_opts(**kw)))
            finally:
                ctx.set_presort(True, 8192)
        assert np.array_equal(_bits(out[0][0]), _bits(out[1][0]))
        assert out[0][1].tobytes() == out[1][1].tobytes()


def test_holding_a_finished_row_changes_the_order_not_the_results(fiksi, ctx):
    """fx_ctx_set_hold_passes: a row that is done waits a few passes for a second one before the hand-over blocks;
    every System's result is the same bits with 0, 2 and 5 passes (uniform, mixed and multi-component batches, f32,
    SinglePass blocks)."""
    from fiksi_amd import abi, workloads

    from helpers import random_sketch

    mixed = workloads.concat([workloads.ring16(3000), workloads.concat([random_sketch(s).flatten() for s in range(300)]),
                              workloads.hinged_triangles(2000, 4)])
    for b, kw in ((workloads.ring16(12000), {}), (workloads.ring16(9000, inconsistent=True), {"f32": True}), (mixed, {}),
                  (workloads.hinged_triangles(6000, 11), {"decomposer": 1}), (workloads.ring16(6000), {"decomposer": 1})):
        out = []
        for passes in (0, 2, 5):
            ctx.set_hold_passes(passes)
            try:
                out.append(ctx.system_solve_batch(b, abi.solving_opts(**kw)))
            finally:
                ctx.set_hold_passes(2)
        for other in out[1:]:
            assert np.array_equal(_bits(out[0][0]), _bits(other[0]))
            assert out[0][1].tobytes() == other[1].tobytes()


def _ladder_variants():
    # (enable, tail_systems, min_trials, spread)
    return [(False, 0, 16, False), (True, 0, 16, False), (True, 0, 16, True), (True, 4096, 8, True), (True, 1 << 30, 0, True)]


def test_the_lambda_ladder_changes_the_order_not_the_results(fiksi, ctx, routing):
    """fx_ctx_set_ladder: rows of a wavefront without a System of their own try lambda x 2, x 4, x 8 of a System that is still
    running beside its own lambda; the first verdict in the loop's order that is not a plain reject decides (lm.rs:114-190).
    Every variable, `accepted`, `trials`, `exit` and SSE of every System is the same bits with the ladder off, with rows
    helping at the end of the queue, with wavefronts leaving the queue early for a straggler, and with a straggler threshold
    of zero and no end to the tail (every wavefront becomes one ladder as soon as any of its rows runs): uniform batches,
    f32 (its stagnation exit is one of the verdicts), inconsistent targets, mixed structures with several components,
    SinglePass blocks, the trial cap."""
    from fiksi_amd import abi, workloads

    from helpers import mixed_sketch, random_sketch

    routing("1")
    mixed = workloads.concat([workloads.ring16(1500), workloads.concat([random_sketch(s).flatten() for s in range(300)]),
                              workloads.concat([mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(100)]),
                              workloads.hinged_triangles(1000, 4)])
    cases = [(workloads.ring16(12500), {}), (workloads.ring16(4099, fix_gauge=True), {}),
             (workloads.ring16(9000, inconsistent=True), {"f32": True}), (workloads.ring16(5000, inconsistent=True), {}),
             (mixed, {}), (workloads.hinged_triangles(3000, 5), {"decomposer": 1}), (workloads.ring16(3000), {"decomposer": 1}),
             (workloads.ring16(3000), {"max_trials": 21}), (workloads.hinged_triangles(2000, 3), {})]
    try:
        for b, kw in cases:
            o = abi.solving_opts(**kw)
            out = []
            for en, tail, k, spread in _ladder_variants():
                ctx.set_ladder(en, tail, k, spread)
                out.append(ctx.system_solve_batch(b, o))
            for other in out[1:]:
                assert np.array_equal(_bits(out[0][0]), _bits(other[0]))
                assert out[0][1].tobytes() == other[1].tobytes()
            if "max_trials" in kw:
                assert int(out[0][1]["trials"].max()) == 21
    finally:
        ctx.set_ladder()


def test_the_ladder_on_the_slowest_systems_of_a_shard(fiksi, oracle, ctx, routing):
    """The 192 Systems of a 12 500-System shard that take the most trials (34 ... 97), as a batch of their own: 48 wavefronts
    of four stragglers each, so every ladder that forms is a partial one (two rows, then three, then four as the
    neighbours finish). Same bits as the sequential loop, and the oracle's counters."""
    from fiksi_amd import workloads

    full = workloads.ring16(12500)
    routing("1")
    ctx.set_ladder(False)
    try:
        _, r = ctx.system_solve_batch(full)
        slow = np.argsort(-r["trials"].astype(np.int64), kind="stable")[:192]
        sb = workloads.concat([workloads.shard(full, int(s), 12500) for s in slow])
        v0, r0 = ctx.system_solve_batch(sb)
        assert np.array_equal(r0["trials"], r["trials"][slow]) and int(r0["trials"].min()) >= 30
        for variant in _ladder_variants()[1:]:
            ctx.set_ladder(*variant)
            v1, r1 = ctx.system_solve_batch(sb)
            assert np.array_equal(_bits(v1), _bits(v0)) and r1.tobytes() == r0.tobytes()
        sub = workloads.shard(sb, 0, 8)
        n = len(sub["var_off"]) - 1
        _, res_o = oracle.solve_batch(sub, mode=3, nthreads=8)
        same = (r0["accepted"][:n] == res_o["accepted"]) & (r0["trials"][:n] == res_o["trials"])
        assert same.mean() >= 0.9  # the normal-equation step against the oracle's QR step
    finally:
        ctx.set_ladder()
