"""FX_STEP_QR — the reference's own numerics on the device (fx_kernels.hip: qr_step; host symbolic phase:
fx_qrplan.h; angle residuals: fx_atan2.h). The LM step is solvi's sparse Householder QR of [J; sqrt(lambda) I]
replayed operation by operation (COLAMD column order, Davis 5.3 row order, every sum in the reference's order,
nothing fused), so the whole solve is bit-identical to the oracle: every variable, every counter, every SSE —
for all eleven expression kinds when both sides evaluate atan2 correctly rounded (the oracle's
'correctly_rounded' mode: binary128 atan2q, independent of the product's routine), and for the nine kinds that
never call atan2 in any mode. Against the platform libm (glibc misrounds ~0.07 % of atan2 arguments by an ulp)
paths still agree on all but a few percent of arbitrary sketches, and every System is checked
(tests/helpers.py: compare_outcomes)."""
import numpy as np
import pytest

from helpers import compare_outcomes, mixed_sketch, random_big_sketch, random_sketch

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def _assert_identical(b, v, res, v_o, res_o, block_sums=False):
    """Bit-identical solve: every variable of every System, and every counter. The one documented deviation: a
    component whose step turns non-finite ends with FX_EXIT_NAN after that trial here, while the reference would
    double lambda for ever (the oracle gives up at its cap of 4096 trials); both keep the component's start
    point, so the variables still agree bit for bit — only the trial counters of those Systems differ."""
    assert np.array_equal(_bits(v), _bits(v_o))
    capped = res_o["trials"] >= 4096
    assert np.array_equal(res["exit"] == 5, (res_o["exit"] == 4))
    for k in ("accepted", "trials", "exit"):
        assert np.array_equal(res[k][~capped], res_o[k][~capped]), k
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["ncomp"], res_o["ncomp"])
    for k in ("scale", "sse0", "sse"):
        eq = (_bits(res[k]) == _bits(res_o[k])) | (np.isnan(res[k]) & np.isnan(res_o[k]))
        if block_sums and k != "scale":
            # SinglePass: the result record's SSEs are sums over the blocks — bookkeeping of ours and of the oracle's,
            # not a quantity of the reference, and the two add the blocks up in different association
            eq |= np.abs(res[k] - res_o[k]) <= 1e-13 * np.abs(res_o[k])
        assert eq.all(), (k, np.nonzero(~eq)[0][:5], res[k][~eq][:3], res_o[k][~eq][:3])


def test_distance_only_sketches_are_bit_identical(fiksi, oracle, ctx):
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.hinged_triangles(8, 1), workloads.hinged_triangles(8, 4), workloads.hinged_triangles(16, 11),
                          workloads.hinged_triangles(4, 15), workloads.quadrilateral(), workloads.quadrilateral(False)])
    v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2))
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    _assert_identical(b, v, res, v_o, res_o)
    assert res["accepted"][-1] == 18 and res["trials"][-1] == 69 and res["exit"][-1] == 2


@pytest.mark.parametrize("decomposer", [0, 1])
def test_random_sketches_without_angles_are_bit_identical(fiksi, oracle, ctx, decomposer):
    """Nine of the eleven expression kinds, random topology, several components, fixed elements, under- and
    over-constrained parts, rank-deficient Jacobians with lambda down to 1e-50: the same bits as the oracle."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([random_sketch(9000 + s, angles=False).flatten() for s in range(500)])
    assert not np.isin(b["expr_tag"], (2, 7)).any() and len(np.unique(b["expr_tag"])) == 9
    v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2, decomposer=decomposer))
    if decomposer:
        v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
    else:
        v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
    _assert_identical(b, v, res, v_o, res_o, block_sums=bool(decomposer))
    assert (res["accepted"] > 30).sum() > 5  # long crawls through flat valleys are part of the sample


def test_l2_entry_point_bit_identical(fiksi, oracle, ctx):
    """fx_lm_solve_batch == levenberg_marquardt(Subsystem) on values as given."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.hinged_triangles(4, 11), workloads.quadrilateral(False)])
    v, res = ctx.lm_solve_batch(b, abi.lm_opts(solver=2))
    v_o, res_o = oracle.solve_batch(b, mode=0)
    _assert_identical(b, v, res, v_o, res_o)


def test_every_expression_kind_bit_identical_with_correctly_rounded_atan2(fiksi, oracle, ctx):
    """ring16 (the headline sketch: eight angle constraints), the mixed sketches (all eleven kinds, one expression
    reading a variable twice, fixed elements, several components) and random sketches: the same bits as the oracle."""
    from fiksi_amd import abi, workloads

    batches = [workloads.concat([workloads.ring16(1024), workloads.ring16(256, fix_gauge=True), workloads.ring16(256, inconsistent=True)]),
               workloads.concat([mixed_sketch(100 + s, fix_some=(s % 3 == 0)).flatten() for s in range(160)]
                                + [random_sketch(s).flatten() for s in range(600)])]
    with oracle.atan2_mode("correctly_rounded"):
        for b in batches:
            v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2))
            v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
            _assert_identical(b, v, res, v_o, res_o)
            # the kernel's own post-solve check (constraints/mod.rs:96-109) from the same output
            r = oracle.residuals_batch(b, v)
            for s in range(0, len(res), 7):
                e0, e1 = int(b["expr_off"][s]), int(b["expr_off"][s + 1])
                want = float((r[e0:e1] ** 2).sum())
                assert abs(res["sse_unscaled"][s] - want) <= 1e-12 * max(want, 1e-300) or not np.isfinite(want)
        b = workloads.concat([random_sketch(40 + s).flatten() for s in range(300)] + [workloads.hinged_triangles(8, 11)])
        v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2, decomposer=1))
        v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
        _assert_identical(b, v, res, v_o, res_o, block_sums=True)
    assert set(np.unique(batches[1]["expr_tag"])) == set(range(11))


def test_against_the_platform_libm_every_system_checked(fiksi, oracle, ctx):
    """The oracle in its default mode evaluates atan2 with this platform's libm, as the reference does here: an ulp
    off the correctly rounded value on a few arguments in ten thousand. Nothing else differs, so most Systems
    still agree bit for bit; the rest are checked against SURVEY 8c's tolerance — all of them, whatever path they took."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.ring16(1024)] + [mixed_sketch(100 + s, fix_some=(s % 3 == 0)).flatten() for s in range(160)]
                         + [random_sketch(s).flatten() for s in range(600)])
    v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2))
    v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
    same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=True)
    assert same >= 0.97 and verdict >= 0.995, (same, verdict)
    identical = np.array([np.array_equal(_bits(v[b["var_off"][s]:b["var_off"][s + 1]]), _bits(v_o[b["var_off"][s]:b["var_off"][s + 1]]))
                          for s in range(len(res))])
    assert identical.mean() >= 0.85, identical.mean()


def test_larger_components_and_64_columns(fiksi, oracle, ctx):
    """Components of 40 ... 64 free variables (the wider instantiations; with 64 columns the right-hand side takes
    a pass of its own) and batches mixing them with Systems beyond one wavefront, which run the refined
    normal-equation step instead."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.hinged_triangles(3, 15), workloads.hinged_triangles(2, 14), workloads.hinged_triangles(2, 9)])
    assert int(b["var_off"][1]) == 62
    v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2))
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    _assert_identical(b, v, res, v_o, res_o)
    # exactly 64 free variables: 32 points chained by distances
    s = fiksi.System()
    P = [fiksi.elements.Point.create(s, 1.1 * i, 0.3 * (i % 3)) for i in range(32)]
    for i in range(31):
        fiksi.constraints.PointPointDistance.create(s, P[i], P[i + 1], 1.0 + 0.01 * i)
    for i in range(0, 30, 2):
        fiksi.constraints.PointPointDistance.create(s, P[i], P[i + 2], 1.9)
    b64 = s.flatten()
    v, res = ctx.system_solve_batch(b64, abi.solving_opts(solver=2))
    v_o, res_o = oracle.solve_batch(b64, mode=3)
    _assert_identical(b64, v, res, v_o, res_o)
    mixed = workloads.concat([random_big_sketch(1000 * 45 + k, 45).flatten() for k in range(6)] + [random_sketch(k).flatten() for k in range(6)])
    v, res = ctx.system_solve_batch(mixed, abi.solving_opts(solver=2))
    v_o, res_o = oracle.solve_batch(mixed, mode=3, trial_cap=4096, nthreads=8)
    compare_outcomes(mixed, v, res, v_o, res_o, oracle, tight=False)


def test_unsupported_combinations_are_reported(fiksi, ctx):
    from fiksi_amd import abi, workloads
    from fiksi_amd._lib import FiksiError

    b = workloads.ring16(4)
    for opts in (abi.solving_opts(solver=2, f32=True), abi.solving_opts(solver=2, optimizer=1), abi.solving_opts(solver=7)):
        if opts.lm.precision == 32:
            opts.lm.solver = 2
        with pytest.raises(FiksiError) as e:
            ctx.system_solve_batch(b, opts)
        assert e.value.code == -6


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_two_runs_are_bit_identical_on_mixed_and_random_batches(fiksi, ctx, solver):
    """Run-to-run determinism where several rows of one wave instruction add into the same normal-matrix entry
    (LDS float atomics) and with every expression kind: two solves of the same batch give the same bits — for
    each step solver, on the one-System-per-wavefront kernels and (forced) on the grouped kernel."""
    from fiksi_amd import abi, workloads

    flats = [mixed_sketch(300 + s, fix_some=(s % 2 == 0)).flatten() for s in range(120)] + [random_sketch(700 + s).flatten() for s in range(400)]
    b = workloads.concat(flats)
    runs = []
    for grouped in (0, 1):
        ctx.set_routing(grouped)
        try:
            a = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
            c = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
        finally:
            ctx.set_routing(-1)
        assert np.array_equal(_bits(a[0]), _bits(c[0])), grouped
        assert a[1].tobytes() == c[1].tobytes(), grouped
        runs.append(a)


@pytest.mark.parametrize("shape", ["ring16", "ring16_gauge", "ring16_inconsistent", "hinged5", "hinged7", "mixed11"])
def test_four_systems_per_wavefront_is_the_same_bits(fiksi, oracle, ctx, shape):
    """Batches of ONE structure run FX_STEP_QR four Systems to a wavefront (fx_grouped.hip: lm_solve_grouped_qr_kernel, the
    program of fx_programs.cpp: build_qrg_program — lanes over the active columns of each Householder step, the matrix stored by its
    symbolic patterns): the same operations in the same order as the one-wavefront QR kernel, so every variable and every
    result field is that kernel's bits (routing 0 / 1), with the lambda ladder on and off, and the oracle's on a sample (the
    correctly rounded atan2 on both sides)."""
    import helpers
    from fiksi_amd import abi, workloads

    b = {"ring16": lambda: workloads.ring16(2050), "ring16_gauge": lambda: workloads.ring16(1500, fix_gauge=True),
         "ring16_inconsistent": lambda: workloads.ring16(1500, inconsistent=True), "hinged5": lambda: workloads.hinged_triangles(1100, 5),
         "hinged7": lambda: workloads.hinged_triangles(1100, 7),
         "mixed11": lambda: workloads.concat([helpers.mixed_sketch(3, fix_some=True).flatten()] * 1100)}[shape]()
    o = abi.solving_opts(solver=2)
    out = {}
    try:
        for tag, ladder in (("1", True), ("1", False), ("0", True)):
            ctx.set_routing(int(tag))
            ctx.set_ladder(ladder, 1 << 30, 4, True)
            db = ctx.upload(b)
            db.system_solve(o)
            if shape != "mixed11":
                assert db.solve_route(o) == int(tag)
            out[(tag, ladder)] = (db.get_vars().copy(), db.get_results().copy())
            db.free()
    finally:
        ctx.set_routing(-1)
        ctx.set_ladder()
    v0, r0 = out[("0", True)]
    for key in (("1", True), ("1", False)):
        v1, r1 = out[key]
        assert np.array_equal(_bits(v1), _bits(v0)), key
        assert r1.tobytes() == r0.tobytes(), key
    sub = workloads.shard(b, 0, 64)
    n = len(sub["var_off"]) - 1
    with oracle.atan2_mode("correctly_rounded"):
        v_o, res_o = oracle.solve_batch(sub, mode=3, nthreads=8)
    _assert_identical(sub, out[("1", True)][0][: len(v_o)], out[("1", True)][1][:n], v_o, res_o)


def test_structure_classes_run_four_systems_per_wavefront_too(fiksi, oracle, ctx):
    """A batch of SEVERAL structures under FX_STEP_QR: every big structure class (256 Systems and more, three quarters of the batch between them) gets the grouped QR
    build's program and a launch over its member list (fx_solve.cpp: launch_class_qr), everybody else — a small class, a sketch
    of another shape — the one-wavefront QR kernel, which passes the classes' Systems by. Every variable and result field is the
    one-wavefront kernel's bits (routing 0), and the oracle's on a sample drawn across the classes and the rest."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.ring16_two_structures(4400), workloads.hinged_triangles(300, 5), workloads.ring16(2100, fix_gauge=True)])
    o = abi.solving_opts(solver=2)
    out = {}
    try:
        for routing in (-1, 0):
            ctx.set_routing(routing)
            db = ctx.upload(b)
            db.system_solve(o)
            out[routing] = (db.get_vars().copy(), db.get_results().copy())
            db.free()
        vh, rh = ctx.system_solve_batch(b, o)  # (the host-buffer call: its chunks carry their own classes)
    finally:
        ctx.set_routing(-1)
    assert np.array_equal(_bits(out[-1][0]), _bits(out[0][0])) and out[-1][1].tobytes() == out[0][1].tobytes()
    assert np.array_equal(_bits(vh), _bits(out[0][0])) and rh.tobytes() == out[0][1].tobytes()
    n = len(b["var_off"]) - 1
    pick = np.sort(np.random.default_rng(11).choice(n, size=96, replace=False))
    sample = workloads.concat([workloads.shard(b, int(s), n) for s in pick])
    with oracle.atan2_mode("correctly_rounded"):
        v_o, res_o = oracle.solve_batch(sample, mode=3, nthreads=8)
    v_s = np.concatenate([out[-1][0][int(b["var_off"][s]):int(b["var_off"][s + 1])] for s in pick])
    _assert_identical(sample, v_s, out[-1][1][pick], v_o, res_o)


@pytest.mark.parametrize("wide_routing", [-1, 1])
def test_components_of_65_to_128_columns_keep_the_reference_numerics(fiksi, oracle, ctx, wide_routing):
    """FX_STEP_QR beyond one wavefront (round 4): Systems whose components have at most 128 columns and 256 rows run the
    wide kernel's QR build (fx_wide.hip: the matrix by its symbolic patterns in LDS, a lane per active column, the host's
    table program) instead of being downgraded to the refined step — whatever the wide / team routing of the plain step is.
    Every variable, counter and SSE is the oracle's bits: hinged chains of 66 ... 126 variables, a sketch with angles (the
    correctly rounded atan2 on both sides), mixed with one-wavefront Systems and a System too large for it (which still
    takes the refined step)."""
    from fiksi_amd import abi, workloads

    ctx.set_wide_routing(wide_routing)
    try:
        b = workloads.concat([workloads.hinged_triangles(3, 16), workloads.ring16(4), workloads.hinged_triangles(2, 24),
                              workloads.large_sketch(40, seed=5), workloads.hinged_triangles(2, 31), workloads.large_sketch(60, seed=9),
                              workloads.hinged_triangles(2, 5)])
        n_big = 3 + 4 + 2 + 1 + 2 + 1 + 2
        assert len(b["var_off"]) - 1 == n_big
        v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2))
        with oracle.atan2_mode("correctly_rounded"):
            v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
        sizes = np.diff(b["var_off"])
        assert sizes.max() <= 128 and (sizes > 64).sum() >= 8
        _assert_identical(b, v, res, v_o, res_o)
        # with a System beyond 128 columns in the batch: that one takes the refined step, the others keep their bits
        b2 = workloads.concat([b, workloads.hinged_triangles(1, 40)])
        v2, res2 = ctx.system_solve_batch(b2, abi.solving_opts(solver=2))
        assert np.array_equal(_bits(v2[: len(v)]), _bits(v)) and res2[:n_big].tobytes() == res.tobytes()
        assert res2["sse_unscaled"][-1] < 1e-6
    finally:
        ctx.set_wide_routing(-1)
