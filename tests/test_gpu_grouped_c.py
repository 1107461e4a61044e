"""The grouped kernel's build for batches of ONE structure (fx_grouped_c.hip: the structure's lists shared by the four Systems
of a wavefront, Jt J stored by its pattern; four wavefronts per SIMD up to 16 free variables, two for 17 ... 32, one on every SIMD for 33 ... 48) against the general build (fx_grouped.hip) it replaces
for such batches: the same operations on the same operands in the same order, so every bit of every result must agree.
(The oracle comparisons of test_gpu_grouped.py / test_gpu_parity.py run through this build too: their ring16 batches are of
one structure.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx_general(fiksi):
    """A context that never takes the one-structure build (FIKSI_AMD_GROUPED_C is read when a context is created)."""
    import os

    old = os.environ.get("FIKSI_AMD_GROUPED_C")
    os.environ["FIKSI_AMD_GROUPED_C"] = "0"
    try:
        c = fiksi.Context(0)
    finally:
        if old is None:
            del os.environ["FIKSI_AMD_GROUPED_C"]
        else:
            os.environ["FIKSI_AMD_GROUPED_C"] = old
    yield c
    c.close()


def _mixed_uniform(n, fix_some):
    """n sketches of ONE topology that uses every constraint kind (an expression reading a variable twice included)."""
    from fiksi_amd import workloads

    from helpers import mixed_sketch

    return workloads.concat([mixed_sketch(2 * s + int(fix_some), fix_some=fix_some).flatten() for s in range(n)])


def _cases():
    from fiksi_amd import workloads

    return [("ring16", workloads.ring16(4099), {}), ("ring16_fixed_gauge", workloads.ring16(3001, fix_gauge=True), {}),
            ("ring16_inconsistent", workloads.ring16(3000, inconsistent=True), {}),
            ("ring16_trial_cap", workloads.ring16(2000), {"max_trials": 21}),
            ("ring16_no_perturbation", workloads.ring16(2000), {"perturb": False}),
            ("ring16_f32", workloads.ring16(4099), {"f32": True}), ("ring16_inconsistent_f32", workloads.ring16(5000, inconsistent=True), {"f32": True}),
            ("hinged_5_f32", workloads.hinged_triangles(1500, 5), {"f32": True}),
            ("hinged_1", workloads.hinged_triangles(3001, 1), {}), ("hinged_3", workloads.hinged_triangles(2000, 3), {}),
            ("hinged_4", workloads.hinged_triangles(2000, 4), {}), ("hinged_5", workloads.hinged_triangles(1500, 5), {}),
            ("hinged_7", workloads.hinged_triangles(1203, 7), {}),
            # 40 / 46 variables whose factor fills in: the 48-column register build (a sparse factor of that size — the hinged
            # chains of 8 ... 11 triangles — goes to fx_grouped_s.hip: tests/test_gpu_grouped_s.py)
            ("ring20_chords", workloads.ring_chords(1000, 20, 7), {}), ("ring23_chords", workloads.ring_chords(1501, 23, 9), {}),
            ("every_kind", _mixed_uniform(1500, False), {}), ("every_kind_some_fixed", _mixed_uniform(1500, True), {})]


def test_which_batches_take_the_one_structure_build(fiksi, ctx, ctx_general):
    from fiksi_amd import abi, workloads

    taken = {}
    for name, b, kw in _cases():
        db = ctx.upload(b)
        taken[name] = db.grouped_build(abi.solving_opts(**kw))
        # the other builds of the same batch are never this one
        assert db.grouped_build(abi.solving_opts(decomposer=1)) != 1 and db.grouped_build(abi.solving_opts(optimizer=1)) != 1
        assert db.grouped_build(abi.solving_opts(solver=1)) != 1 and db.grouped_build(abi.solving_opts(solver=2)) != 1
        db.free()
        dg = ctx_general.upload(b)
        assert dg.grouped_build(abi.solving_opts(**kw)) == 0
        dg.free()
    for name in ("ring16", "ring16_fixed_gauge", "ring16_inconsistent", "ring16_trial_cap", "ring16_no_perturbation", "ring16_f32",
                 "ring16_inconsistent_f32", "hinged_5_f32", "hinged_3", "hinged_4", "hinged_5", "hinged_7", "ring20_chords", "ring23_chords"):
        assert taken[name] == 1, taken
    assert taken["hinged_1"] == 4, taken  # (six variables, three expressions: eight lanes per System, fx_grouped_tiny.hip)
    # not of one structure: classes of 256 Systems and more take this build (3) when they hold three quarters of the batch, else the
    # general build for everybody; f32 has the 32-column instantiation only
    db = ctx.upload(workloads.ring16_two_structures(2000))
    assert db.grouped_build() == 3
    db.free()
    db = ctx.upload(workloads.concat([workloads.ring16(300), workloads.ring16_all_different(1000)]))
    assert db.grouped_build() == 0  # (one class of 300 among 1 300 Systems)
    db.free()
    db = ctx.upload(workloads.ring16_two_structures(400))
    assert db.grouped_build() == 0  # (classes of 200)
    db.free()
    db = ctx.upload(workloads.hinged_triangles(2000, 3))
    assert db.grouped_build(abi.solving_opts(f32=True)) == 0
    db.free()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_same_bits_as_the_general_build(fiksi, ctx, ctx_general):
    """Every solved variable and every field of every result record, for the queue in index order and longest-first, with
    the lambda ladder off, at the end of the queue, and with every wavefront one ladder from the start."""
    from fiksi_amd import abi

    try:
        for name, b, kw in _cases():
            o = abi.solving_opts(**kw)
            ctx_general.set_ladder(True)
            v0, r0 = ctx_general.system_solve_batch(b, o)
            for ladder in ((False, 0, 16, False), (True, 0, 16, True), (True, 1 << 30, 0, True)):
                ctx.set_ladder(*ladder)
                for presort in (True, False):
                    ctx.set_presort(presort, 1)
                    v1, r1 = ctx.system_solve_batch(b, o)
                    assert np.array_equal(_bits(v1), _bits(v0)), (name, ladder, presort)
                    assert r1.tobytes() == r0.tobytes(), (name, ladder, presort)
            if "max_trials" in kw:
                assert int(r0["trials"].max()) == 21
    finally:
        ctx.set_ladder()
        ctx.set_presort(True, 8192)
        ctx_general.set_ladder()


def test_resident_batch_solved_again_and_in_chunks(fiksi, ctx, ctx_general):
    """A resident batch keeps its program between solves; the host-buffer call cuts a big batch into chunks, each with its own
    copy of the program."""
    from fiksi_amd import workloads

    b = workloads.ring16(30000)
    db = ctx.upload(b)
    assert db.grouped_build() == 1
    db.system_solve()
    v1, r1 = db.get_vars().copy(), db.get_results().copy()
    db.system_solve()
    assert np.array_equal(_bits(db.get_vars()), _bits(v1)) and db.get_results().tobytes() == r1.tobytes()
    db.free()
    v0, r0 = ctx_general.system_solve_batch(b)
    v2, r2 = ctx.system_solve_batch(b)  # (a big host batch goes up and is solved in two chunks)
    assert np.array_equal(_bits(v1), _bits(v0)) and r1.tobytes() == r0.tobytes()
    assert np.array_equal(_bits(v2), _bits(v0)) and r2.tobytes() == r0.tobytes()


def test_batches_of_several_structures_run_their_big_classes_on_this_build(fiksi, ctx, ctx_general):
    """A few sketches, many parameter sets each: every structure class of 256 Systems and more gets a program, and ONE launch
    works through all of them (a wavefront loads the next class's program when its own class's queue is empty); the Systems of
    small classes take the general build. Every bit as in the general build alone — resident, through the host-buffer call, with
    and without the longest-first order within the classes, with the ladder off / everywhere."""
    from fiksi_amd import workloads

    from helpers import mixed_sketch

    cases = [workloads.ring16_two_structures(6000),
             workloads.concat([workloads.ring16(2500), workloads.hinged_triangles(3000, 5), workloads.ring16(2100, fix_gauge=True),
                               workloads.hinged_triangles(700, 3), workloads.hinged_triangles(2048, 7),
                               workloads.concat([mixed_sketch(s).flatten() for s in range(40)])])]
    try:
        for b in cases:
            v0, r0 = ctx_general.system_solve_batch(b)
            for ladder in ((True, 0xFFFFFFFF, 8, True), (False, 0, 16, False), (True, 1 << 30, 0, True)):
                ctx.set_ladder(*ladder)
                for presort in (True, False):
                    ctx.set_presort(presort, 1)
                    v1, r1 = ctx.system_solve_batch(b)
                    assert np.array_equal(_bits(v1), _bits(v0)) and r1.tobytes() == r0.tobytes(), (ladder, presort)
                    db = ctx.upload(b)
                    assert db.grouped_build() == 3
                    db.system_solve()
                    db.system_solve()
                    assert np.array_equal(_bits(db.get_vars()), _bits(v0)) and db.get_results().tobytes() == r0.tobytes(), (ladder, presort)
                    db.free()
    finally:
        ctx.set_ladder()
        ctx.set_presort(True, 8192)


@pytest.mark.parametrize("seed", range(10))
def test_random_structures_against_the_general_build(fiksi, ctx, ctx_general, seed):
    """Random connected sketches of 4 ... 24 points (distances over a spanning tree and between random pairs, a few angles,
    sometimes a fixed point): whichever instantiation the structure takes — one, two or three columns per lane —, every bit of
    the general build. (A 17 ... 24-point structure with a sparse factor goes to the sparse build instead: its own tests.)"""
    from helpers import Lcg
    from test_gpu_grouped_s import _random_graph_batch

    g = Lcg(4000 + 7919 * seed)
    n_pts = (4, 6, 8, 9, 12, 14, 16, 18, 21, 24)[seed]
    b = _random_graph_batch(300, n_pts, (1, 2, 5, 9, 8, 20, 30, 40, 60, 80)[seed], 500 + seed, fix_first=bool(seed & 1), angles=int(g.u(0, 5.99)))
    db = ctx.upload(b)
    build = db.grouped_build()
    db.free()
    nv, ne = int(b["var_off"][1]), int(b["expr_off"][1])
    assert build in (1, 2, 4) or ne > 48 or nv > 48, (build, nv, ne)  # (the tables of the register builds hold 48 expressions; 4: the tiny build)
    v1, r1 = ctx.system_solve_batch(b)
    v0, r0 = ctx_general.system_solve_batch(b)
    if build != 2:
        assert np.array_equal(_bits(v1), _bits(v0)) and r1.tobytes() == r0.tobytes()
    else:
        same = (r0["accepted"] == r1["accepted"]) & (r0["trials"] == r1["trials"]) & (r0["exit"] == r1["exit"])
        assert same.mean() >= 0.9


@pytest.mark.parametrize("case", ["16_points_45_rows", "16_points_45_rows_f32", "8_points_23_rows", "16_points_37_rows_fixed"])
def test_over_constrained_structures_take_the_instantiations_with_twice_the_rows(fiksi, ctx, ctx_general, case):
    """More expressions than the shape's 16 / 32 rows (cfg5's theme: least squares over more constraints than unknowns): the same
    build with twice the row chunks, every bit of the general build — f64 and f32."""
    from fiksi_amd import abi
    from test_gpu_grouped_s import _random_graph_batch

    n_pts, n_extra, fix, f32 = {"16_points_45_rows": (16, 30, False, False), "16_points_45_rows_f32": (16, 30, False, True),
                                "8_points_23_rows": (8, 16, False, False), "16_points_37_rows_fixed": (16, 20, True, False)}[case]
    b = _random_graph_batch(400, n_pts, n_extra, 321 + n_extra, fix_first=fix, angles=2)
    o = abi.solving_opts(f32=f32)
    db = ctx.upload(b)
    assert int(b["expr_off"][1]) > 16 * (1 if n_pts == 8 else 2) and db.grouped_build(o) == 1
    db.free()
    v1, r1 = ctx.system_solve_batch(b, o)
    v0, r0 = ctx_general.system_solve_batch(b, o)
    assert np.array_equal(_bits(v1), _bits(v0)) and r1.tobytes() == r0.tobytes()


def _solve_on_build(ctx, b, o, want_build):
    """Upload, make sure the solve WILL take the build this test is about (fx_debug_grouped_build), solve resident."""
    db = ctx.upload(b)
    try:
        assert db.grouped_build(o) == want_build, (db.grouped_build(o), want_build)
        db.system_solve(o)
        return db.get_vars(), db.get_results()
    finally:
        db.free()


@pytest.mark.parametrize("case", ["c1_hinged_1", "c1_hinged_3", "c_ring16", "c_ring16_fixed_gauge", "c_every_kind", "c3_ring20_chords",
                                  "cr_16_points_45_rows", "c1r_8_points_23_rows"])
def test_each_instantiation_against_the_oracle_directly(fiksi, oracle, ctx, case):
    """The one-structure instantiations (one / two / three columns per lane, and those with twice the row chunks) are held to
    the ORACLE here, not to another kernel: every System of the batch through tests/helpers.py: compare_outcomes with the tight
    bars (same path -> |dSSE| <= 1e-10 + 1e-6 SSE and per-constraint residuals; otherwise the same verdict; the
    looser bars of tests/test_gpu_fuzz.py for the ill-conditioned sketches), and
    the test first asserts — by name, through fx_debug_grouped_build — that the solve it is about to check runs on
    lm_solve_grouped_c*_kernel, so that a routing change cannot move the comparison off this kernel unnoticed."""
    from fiksi_amd import abi, workloads

    from helpers import compare_outcomes
    from test_gpu_grouped_s import _random_graph_batch

    b = {"c1_hinged_1": lambda: workloads.hinged_triangles(1500, 1), "c1_hinged_3": lambda: workloads.hinged_triangles(1200, 3),
         "c_ring16": lambda: workloads.ring16(3000), "c_ring16_fixed_gauge": lambda: workloads.ring16(1500, fix_gauge=True),
         "c_every_kind": lambda: _mixed_uniform(800, False), "c3_ring20_chords": lambda: workloads.ring_chords(600, 20, 7),
         "cr_16_points_45_rows": lambda: _random_graph_batch(400, 16, 30, 351, fix_first=False, angles=2),
         "c1r_8_points_23_rows": lambda: _random_graph_batch(400, 8, 16, 337, fix_first=False, angles=2)}[case]()
    o = abi.solving_opts()
    ctx.set_one_structure_builds(True, tiny=False)  # (c1_hinged_1 would take fx_grouped_tiny.hip: its own test below)
    try:
        v, res = _solve_on_build(ctx, b, o, 1)
    finally:
        ctx.set_one_structure_builds(True)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    # (under-determined, partly infeasible sketches — every kind, the random over-constrained graphs — stall in flat valleys, where the
    # normal-equation step's cond^2 shows: the looser bars tests/test_gpu_fuzz.py holds such sketches to; FX_STEP_QR is the bit-exact mode)
    loose = case in ("c_every_kind", "cr_16_points_45_rows", "c1r_8_points_23_rows")
    if case == "c_every_kind":
        # every constraint kind on one small sketch, nothing fixed: the flattest valleys of all. Structure-level quantities exactly; the
        # oracle's path on >= 90 %; on that path the final SSE within 1e-8 + 25 % on >= 99 %; the bench's verdict on >= 99.5 %
        assert np.array_equal(res["scale"], res_o["scale"]) and np.array_equal(res["ncomp"], res_o["ncomp"])
        same_path = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"]) & (res["exit"] == res_o["exit"])
        d = np.abs(res["sse"] - res_o["sse"])
        assert same_path.mean() >= 0.9 and (d[same_path] <= 1e-8 + 0.25 * np.abs(res_o["sse"][same_path])).mean() >= 0.99
        n = len(res)
        sq = (oracle.residuals_batch(b, v).reshape(n, -1) ** 2).sum(1)
        sq_o = (oracle.residuals_batch(b, v_o).reshape(n, -1) ** 2).sum(1)
        assert ((sq < 1e-4) == (sq_o < 1e-4)).mean() >= 0.995
        return
    same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=not loose)
    assert same >= (0.9 if loose else 0.97) and verdict >= 0.995, (case, same, verdict)


def _tiny_cases(n):
    from fiksi_amd import workloads

    from helpers import tiny_sketch_batches
    return [("hinged_1", workloads.hinged_triangles(n, 1))] + tiny_sketch_batches(n)


@pytest.mark.parametrize("case", ["hinged_1", "quadrilateral", "quadrilateral_impossible", "quadrilateral_fixed_point", "four_points_five_kinds"])
def test_the_tiny_build_against_the_oracle_directly_and_bit_for_bit_against_the_16_column_build(fiksi, oracle, ctx, case):
    """fx_grouped_tiny.hip (eight lanes per System; asserted by name: fx_debug_grouped_build == 4) against the oracle with the bars of
    the test above, and against lm_solve_grouped_c1_kernel on the same batch: every bit of the variables and of the result records —
    with and without the perturbation, under a trial cap, and through the host-buffer call. 1 003 Systems: the last wavefront is ragged."""
    from fiksi_amd import abi

    from helpers import compare_outcomes

    b = dict(_tiny_cases(1003))[case]
    for o in (abi.solving_opts(), abi.solving_opts(perturb=False), abi.solving_opts(max_trials=7)):
        v, res = _solve_on_build(ctx, b, o, 4)
        ctx.set_one_structure_builds(True, tiny=False)
        try:
            v1, res1 = _solve_on_build(ctx, b, o, 1)
        finally:
            ctx.set_one_structure_builds(True)
        assert np.array_equal(_bits(v), _bits(v1)) and res.tobytes() == res1.tobytes()
    o = abi.solving_opts()
    v, res = _solve_on_build(ctx, b, o, 4)
    vh, resh = ctx.system_solve_batch(b, o)
    assert np.array_equal(_bits(v), _bits(vh)) and res.tobytes() == resh.tobytes()
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    loose = case in ("quadrilateral_impossible", "four_points_five_kinds")  # (flat valleys: the bars of tests/test_gpu_fuzz.py)
    same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=not loose)
    assert same >= (0.9 if loose else 0.97) and verdict >= 0.995, (case, same, verdict)


def test_a_lone_tiny_system_and_a_handful_take_the_tiny_build_too(fiksi, oracle, ctx, ctx_general):
    """System::solve on one sketch of at most eight variables (the reference's own bench at one triangle) and batches below the
    grouped kernel's eight-System threshold take the tiny build too: a System's bits do not depend on how many others share its
    batch. (Against the one-System-per-wavefront kernel such a batch took before: every bit on the distance-only sketches, the
    oracle's bars on the sketch with angles and a repeated variable, whose Jt J that kernel sums in another order.)"""
    from fiksi_amd import abi

    from helpers import compare_outcomes

    def take(b, lo, hi):  # Systems lo ... hi - 1 of a batch of one structure
        nv, ne = int(b["var_off"][1]), int(b["expr_off"][1])
        out = {"var_off": (np.arange(hi - lo + 1) * nv).astype(np.uint32), "expr_off": (np.arange(hi - lo + 1) * ne).astype(np.uint32),
               "expr_idx": b["expr_idx"][4 * ne * lo:4 * ne * hi].copy()}
        for k in ("vars", "var_fixed", "var_comp"):
            out[k] = b[k][nv * lo:nv * hi].copy()
        for k in ("expr_tag", "expr_param", "expr_comp"):
            out[k] = b[k][ne * lo:ne * hi].copy()
        return out

    for name, b in _tiny_cases(8):
        v_all, res_all = ctx.system_solve_batch(b)
        nv = int(b["var_off"][1])
        for lo, hi in ((3, 4), (0, 2), (1, 8)):
            sub_b = take(b, lo, hi)
            db = ctx.upload(sub_b)
            assert db.grouped_build() == 4, (name, lo, hi, db.grouped_build())
            db.free()
            v, res = ctx.system_solve_batch(sub_b)
            assert np.array_equal(_bits(v), _bits(v_all[nv * lo:nv * hi])) and res.tobytes() == res_all[lo:hi].tobytes(), (name, lo, hi)
            vg, resg = ctx_general.system_solve_batch(sub_b)
            if name != "four_points_five_kinds":
                assert np.array_equal(_bits(v), _bits(vg)) and res.tobytes() == resg.tobytes(), (name, lo, hi)
            else:
                same, verdict = compare_outcomes(sub_b, v, res, vg, resg, oracle, tight=False)
                assert verdict == 1.0, (name, lo, hi, same, verdict)


@pytest.mark.parametrize("case", ["ring16", "ring16_inconsistent", "16_points_45_rows"])
def test_the_f32_instantiations_against_the_oracle_directly(fiksi, oracle, ctx, case):
    """lm_solve_grouped_c_f32_kernel / ..._cr_f32_kernel (asserted by name) against the f64 oracle with cfg5's stated f32
    tolerances (tests/test_gpu_parity.py): the same verdict on >= 99 % of the Systems; final scaled SSE within 1e-3 relative
    (+ 1e-7) on 95 % — within 1e-4 relative where no System reaches zero residual."""
    from fiksi_amd import abi, workloads

    from test_gpu_grouped_s import _random_graph_batch

    b = {"ring16": lambda: workloads.ring16(2048), "ring16_inconsistent": lambda: workloads.ring16(2048, inconsistent=True),
         "16_points_45_rows": lambda: _random_graph_batch(400, 16, 30, 351, fix_first=False, angles=2)}[case]()
    o = abi.solving_opts(f32=True)
    v, res = _solve_on_build(ctx, b, o, 1)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    assert np.array_equal(res["scale"], res_o["scale"])  # K0 stays f64
    n = len(res)
    r_o = oracle.residuals_batch(b, v_o).reshape(n, -1)
    assert ((res["sse_unscaled"] < 1e-4) == ((r_o * r_o).sum(1) < 1e-4)).mean() >= 0.99
    if case == "ring16_inconsistent":
        rel = np.abs(res["sse"] - res_o["sse"]) / res_o["sse"]
        assert np.percentile(rel, 95) <= 1e-4 and np.median(rel) <= 1e-5
    else:
        rel = np.abs(res["sse"] - res_o["sse"]) / (1e-7 + res_o["sse"])
        assert np.percentile(rel, 95) <= 1e-3
