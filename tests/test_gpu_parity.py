"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on identical inputs. Bit-exact where the arithmetic is IEEE-deterministic (scale,
perturbation, Jacobian values, non-angle residuals); stated tolerances elsewhere."""
import numpy as np
import pytest

from helpers import Lcg, mixed_sketch

pytestmark = pytest.mark.gpu

ANGLE_TAGS = (2, 7)  # residuals that go through atan2 (device libm vs glibc): ulp-level differences


def _flatten(fiksi, systems):
    return fiksi.flatten(systems)


def test_context_reports_gfx950(ctx):
    assert "gfx950" in ctx.name()


def test_k1_residual_jacobian_ring16_bit_exact(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.ring16(512)
    r, (rp, ci, vals) = ctx.eval_residual_jacobian(b)
    r_o, (rp_o, ci_o, vals_o) = oracle.eval_batch(b)
    assert np.array_equal(rp.astype(np.int64), rp_o)
    assert np.array_equal(ci.astype(np.int32), ci_o)
    assert np.array_equal(vals, vals_o)  # gradients never touch atan2: bit-exact
    ang = np.isin(b["expr_tag"], ANGLE_TAGS)
    assert np.array_equal(r[~ang], r_o[~ang])
    assert np.max(np.abs(r[ang] - r_o[ang])) <= 4e-15  # |angle| <= pi, a few ulp of atan2


@pytest.mark.parametrize("builder, n", [("ring16", 1500), ("hinged11", 2100), ("hinged3", 5000), ("gauge", 900), ("mixed_copies", 700)])
def test_k1_one_structure_batches_read_their_structure_from_the_first_system(fiksi, oracle, ctx, builder, n):
    """A batch of ONE structure (one sketch, many parameter sets) reads kinds, fields and the tag-sorted order of a block
    from the first System / the first `period` blocks (fx_eval.hip, round 3) instead of streaming them per row. Row counts
    of 32, 33, 9 and 36 per System against blocks of 256 rows: periods of 4, 132, 36 and 12 blocks, batches several
    periods long with a partial last block; fixed variables (the gauge) and a mixed sketch with every kind. Bit for bit
    the oracle's values (angle residuals to 4e-15), residual-only kernel included."""
    from fiksi_amd import workloads

    if builder == "ring16":
        b = workloads.ring16(n, seed0=77)
    elif builder == "gauge":
        b = workloads.ring16(n, seed0=5, fix_gauge=True)
    elif builder == "hinged11":
        b = workloads.hinged_triangles(n, 11)
    elif builder == "hinged3":
        b = workloads.hinged_triangles(n, 3)
    else:
        one = mixed_sketch(3, fix_some=True).flatten()
        parts = []
        g = Lcg(11)
        for k in range(n):
            c = {key: val.copy() for key, val in one.items()}
            c["vars"] = c["vars"] * (1.0 + 0.01 * g.u(-1, 1))
            parts.append(c)
        b = workloads.concat(parts)
    nsys = len(b["var_off"]) - 1
    assert nsys >= 2 and int(b["expr_off"][-1]) > 3 * 256
    r, (rp, ci, vals) = ctx.eval_residual_jacobian(b)
    r_o, (rp_o, ci_o, vals_o) = oracle.eval_batch(b)
    assert np.array_equal(rp.astype(np.int64), rp_o) and np.array_equal(ci.astype(np.int32), ci_o)
    assert np.array_equal(vals, vals_o)
    ang = np.isin(b["expr_tag"], ANGLE_TAGS)
    assert np.array_equal(r[~ang], r_o[~ang])
    assert not ang.any() or np.max(np.abs(r[ang] - r_o[ang])) <= 4e-15
    r2, _ = ctx.eval_residual_jacobian(b, want_jacobian=False)
    assert np.array_equal(r, r2)


def test_k1_all_eleven_expression_kinds(fiksi, oracle, ctx):
    systems = [mixed_sketch(seed, fix_some=(seed % 2 == 1)) for seed in range(40)]
    b = _flatten(fiksi, systems)
    assert set(np.unique(b["expr_tag"])) == set(range(11))
    r, (rp, ci, vals) = ctx.eval_residual_jacobian(b)
    r_o, (rp_o, ci_o, vals_o) = oracle.eval_batch(b)
    assert np.array_equal(rp.astype(np.int64), rp_o)
    assert np.array_equal(ci.astype(np.int32), ci_o)
    assert np.array_equal(vals, vals_o)
    ang = np.isin(b["expr_tag"], ANGLE_TAGS)
    assert np.array_equal(r[~ang], r_o[~ang])
    assert np.max(np.abs(r[ang] - r_o[ang])) <= 4e-15


def test_residual_only_kernel_matches_jacobian_kernel(fiksi, ctx):
    from fiksi_amd import workloads

    b = workloads.ring16(128)
    r1, _ = ctx.eval_residual_jacobian(b, want_jacobian=True)
    r2, _ = ctx.eval_residual_jacobian(b, want_jacobian=False)
    assert np.array_equal(r1, r2)


def test_scale_and_perturbation_bit_exact(fiksi, oracle, ctx):
    """K0: with zero LM iterations the output is scale * perturbed(x / scale)."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.ring16(64), workloads.hinged_triangles(3, 11), workloads.quadrilateral()])
    v, res = ctx.system_solve_batch(b, abi.solving_opts(perturb=True, max_outer=0))
    scale = oracle.system_scale_batch(b)
    assert np.array_equal(res["scale"], scale)
    draws = oracle.rng_f64(42, 2 * 64)
    for s in range(len(scale)):
        v0, v1 = int(b["var_off"][s]), int(b["var_off"][s + 1])
        x = b["vars"][v0:v1] * (1.0 / scale[s])
        k = np.arange(v1 - v0)
        x = x + (x * (1.0 / 8196.0) * draws[2 * k] + (1.0 / 65568.0) * draws[2 * k + 1])
        assert np.array_equal(v[v0:v1], scale[s] * x), f"system {s}"


def _compare_solves(res, res_o, v, v_o, b, oracle, min_same=0.97):
    """Every System of the batch against the oracle (tests/helpers.py: compare_outcomes with the tight bars): none is
    dropped for having left the oracle's path — those are held to the same verdict and a bounded SSE difference."""
    from helpers import compare_outcomes

    same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=True)
    assert same >= min_same, f"on the oracle's path: {same:.3%}"
    assert verdict == 1.0, verdict


def test_system_solve_ring16_matches_oracle(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.ring16(2048)
    db = ctx.upload(b)
    assert db.grouped_build() == 1  # (the default route of this batch: the grouped kernel's one-structure build, fx_grouped_c.hip)
    db.free()
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    _compare_solves(res, res_o, v, v_o, b, oracle)
    # the kernel's own unscaled SSE equals a recomputation from its output
    r = oracle.residuals_batch(b, v).reshape(len(res), -1)
    assert np.allclose(res["sse_unscaled"], (r * r).sum(1), rtol=1e-9, atol=1e-18)


def test_gauge_fixed_positions_match_oracle(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.ring16(512, fix_gauge=True)
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    same = (res["accepted"] == res_o["accepted"]) & (res_o["sse"] < 1e-8)
    assert same.mean() > 0.9
    vv, vo = v.reshape(len(res), -1), v_o.reshape(len(res), -1)
    err = np.max(np.abs(vv - vo), axis=1)
    assert np.all(err[same] <= 1e-6 * res_o["scale"][same])
    # fixed points are bit-identical to the input (fiksi/src/tests/fixed.rs:36-40)
    assert np.array_equal(vv[:, :4], b["vars"].reshape(len(res), -1)[:, :4])


def test_lm_solve_l2_matches_oracle(fiksi, oracle, ctx):
    """L2 boundary: levenberg_marquardt(Subsystem) on values as given (no scale, no perturbation)."""
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(4, 11)
    v, res = ctx.lm_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=0)
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["trials"], res_o["trials"])
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-6, atol=1e-12)
    assert np.all(res["scale"] == 1.0)


def test_inconsistent_targets_exit_like_oracle(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.concat([workloads.quadrilateral(False), workloads.ring16(256, inconsistent=True)])
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    # cfg1's impossible quadrilateral: 18 accepted steps, 69 trials, ftol exit, SSE 1.1214
    assert res["accepted"][0] == res_o["accepted"][0] == 18
    assert res["trials"][0] == res_o["trials"][0] == 69
    assert res["exit"][0] == res_o["exit"][0] == 2
    assert abs(res["sse"][0] - res_o["sse"][0]) < 1e-9
    same = res["accepted"] == res_o["accepted"]
    assert same.mean() > 0.9
    assert np.allclose(res["sse"][same], res_o["sse"][same], rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_mixed_sketches_and_multi_component(fiksi, oracle, ctx, solver):
    """All eleven kinds, fixed elements, several components, arbitrary (often infeasible) targets. Every System is
    compared (helpers.compare_outcomes), whatever path it took.
      solver 2 (FX_STEP_QR, reference numerics): the oracle's bits;
      solver 1 (refined normal equations): same path on >= 90 %, SURVEY 8c's SSE tolerance on those;
      solver 0 (plain normal equations, cond^2): under-determined, partly infeasible sketches crawl through flat
      valleys for 20-40 accepted steps; rounding differences against the reference's QR step are amplified along
      the way — same path on >= 80 %, same-path SSE within 25 % (median 1e-4), equal verdicts on >= 95 %."""
    from fiksi_amd import abi

    from helpers import compare_outcomes

    systems = [mixed_sketch(100 + seed, fix_some=(seed % 3 == 0)) for seed in range(64)]
    b = _flatten(fiksi, systems)
    with oracle.atan2_mode("correctly_rounded" if solver == 2 else "libm"):
        v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
        v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
        same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=(solver == 2))
    if solver == 2:
        assert same == 1.0 and verdict == 1.0
        assert np.array_equal(v.view(np.uint64), v_o.view(np.uint64))
        for k in ("accepted", "trials", "exit"):
            assert np.array_equal(res[k], res_o[k]), k
    elif solver == 1:
        assert same >= 0.9 and verdict >= 0.95, (same, verdict)
        sp = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"])
        d = np.abs(res["sse"] - res_o["sse"])[sp]
        assert np.mean(d <= 1e-10 + 1e-6 * np.abs(res_o["sse"][sp])) >= 0.9
    else:
        assert same > 0.8 and verdict >= 0.95, (same, verdict)
        sp = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"])
        d = np.abs(res["sse"][sp] - res_o["sse"][sp])
        assert np.median(d / (1e-12 + np.abs(res_o["sse"][sp]))) <= 1e-4


def test_device_batch_resolve_is_repeatable(fiksi, ctx):
    from fiksi_amd import workloads

    b = workloads.ring16(300)
    db = ctx.upload(b)
    db.system_solve()
    v1, r1 = db.get_vars(), db.get_results()
    db.system_solve()
    v2, r2 = db.get_vars(), db.get_results()
    assert np.array_equal(v1, v2) and np.array_equal(r1, r2)  # deterministic, start values untouched
    db.free()


def test_empty_and_ragged_batches(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    empty = {k: np.zeros(0, dtype=v.dtype) for k, v in workloads.quadrilateral().items()}
    empty["var_off"] = np.zeros(1, dtype=np.uint32)
    empty["expr_off"] = np.zeros(1, dtype=np.uint32)
    v, res = ctx.system_solve_batch(empty)
    assert len(v) == 0 and len(res) == 0
    # a system with variables but no expressions, one with a single constraint, one bigger
    s0 = fiksi.System()
    fiksi.elements.Point.create(s0, 1.0, 2.0)
    s1 = fiksi.System()
    a = fiksi.elements.Point.create(s1, 0.0, 0.0)
    c = fiksi.elements.Point.create(s1, 1.0, 0.5)
    fiksi.constraints.PointPointCoincidence.create(s1, a, c)
    b = workloads.concat([fiksi.flatten([s0, s1]), workloads.hinged_triangles(1, 15), workloads.ring16(3)])
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3)
    assert np.array_equal(v[:2], b["vars"][:2])  # unconstrained element untouched
    assert res["ncomp"][0] == 0 and res_o["ncomp"][0] == 0
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-6, atol=1e-12)


def test_limits_are_reported_not_crashed(fiksi, ctx):
    from fiksi_amd import workloads
    from fiksi_amd._lib import FiksiError

    bad = workloads.quadrilateral()
    bad["expr_idx"] = bad["expr_idx"].copy()
    bad["expr_idx"][0] = 99
    with pytest.raises(FiksiError) as e:
        ctx.system_solve_batch(bad)
    assert e.value.code == -1
    # more than 65535 variables in one System: 16-bit local indices are exhausted
    n = 70000
    huge = {
        "var_off": np.array([0, n], dtype=np.uint32), "expr_off": np.array([0, 1], dtype=np.uint32),
        "vars": np.zeros(n), "var_fixed": np.zeros(n, dtype=np.uint8), "expr_tag": np.ones(1, dtype=np.uint8),
        "expr_idx": np.array([0, 2, 0, 0], dtype=np.uint32), "expr_param": np.ones(1),
    }
    with pytest.raises(FiksiError) as e:
        ctx.system_solve_batch(huge)
    assert e.value.code == -4


# ---- sparse large-sketch path (components beyond the one-wavefront limits) -------------------------

def test_large_sketch_path_matches_oracle(fiksi, oracle, ctx):
    """Systems with more than 64 free variables go through fx_sparse.hip: same iteration counts as the
    oracle, positions to ~1e-10 (these sketches are well conditioned)."""
    from fiksi_amd import workloads

    for b in (workloads.hinged_triangles(1, 16), workloads.hinged_triangles(2, 64), workloads.large_sketch(300)):
        v, res = ctx.system_solve_batch(b)
        v_o, res_o = oracle.solve_batch(b, mode=3)
        assert np.array_equal(res["accepted"], res_o["accepted"])
        assert np.array_equal(res["trials"], res_o["trials"])
        assert np.array_equal(res["exit"], res_o["exit"])
        assert np.array_equal(res["scale"], res_o["scale"])
        assert np.allclose(res["sse"], res_o["sse"], rtol=1e-6, atol=1e-12)
        assert np.max(np.abs(v - v_o)) <= 1e-9 * res_o["scale"].max()
        r = oracle.residuals_batch(b, v)
        assert float((r * r).sum()) < 1e-4  # fiksi_bench.rs:65-72 spot-check (n = 16, 64)


def test_mixed_small_and_large_systems_in_one_batch(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.concat([workloads.ring16(5), workloads.hinged_triangles(1, 20), workloads.ring16(3, seed0=77),
                          workloads.large_sketch(120), workloads.quadrilateral()])
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3)
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["exit"], res_o["exit"])
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-6, atol=1e-12)
    # device-resident batch: same answer, repeatable
    db = ctx.upload(b)
    db.system_solve()
    assert np.array_equal(db.get_vars(), v)
    db.free()


def test_component_quirks_q1_q2_match_oracle(fiksi, oracle, ctx):
    """graph.rs:211-222 (stale component label, quirk Q1) + assemble/mod.rs:161-166 (later components
    read the pre-solve snapshot, quirk Q2): d belongs to the first component and is read as a fixed
    value, at its PRE-solve position, by the second one."""
    from fiksi_amd import System, constraints, elements

    s = System()
    a, b_, c, d, e = (elements.Point.create(s, 1.3 * i, 0.4 * i * i) for i in range(5))
    constraints.PointPointDistance.create(s, a, b_, 2.)
    constraints.PointPointDistance.create(s, c, d, 2.)
    constraints.PointPointDistance.create(s, a, c, 3.)
    constraints.PointPointDistance.create(s, d, e, 1.)
    flat = s.flatten()
    assert flat["var_comp"].tolist() == [0] * 8 + [1, 1]
    v, res = ctx.system_solve_batch(flat)
    v_o, res_o = oracle.solve_batch(flat, mode=3)
    assert res["ncomp"][0] == res_o["ncomp"][0] == 2
    assert res["accepted"][0] == res_o["accepted"][0]
    assert np.max(np.abs(v - v_o)) < 1e-9


def test_cfg2_one_large_sketch_matches_oracle_fixture(fiksi, ctx):
    """BASELINE cfg2: ONE sketch of 5 000 points / 10 000 mixed distance + angle constraints (10 000
    variables, 46 000 Jacobian non-zeros), f64, one MI355X. The oracle needs ~85 s for it, so its outcome
    is a committed fixture (tests/golden/cfg2_oracle.json, made by tests/golden/make_cfg2_golden.py)."""
    import json
    import os

    from fiksi_amd import workloads

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cfg2_oracle.json")))
    b = workloads.large_sketch(5000)
    assert int(b["var_off"][-1]) == 10000 and int(b["expr_off"][-1]) == 10000
    v, res = ctx.system_solve_batch(b)
    r = res[0]
    assert (int(r["accepted"]), int(r["trials"]), int(r["exit"])) == (gold["accepted"], gold["trials"], gold["exit"])
    assert r["scale"] == gold["scale"]
    assert abs(r["sse0"] - gold["sse0"]) <= 1e-9 * gold["sse0"]
    assert abs(r["sse"] - gold["sse"]) <= 1e-9 + 1e-6 * gold["sse"]
    assert abs(r["sse_unscaled"] - gold["sse_unscaled"]) <= 1e-6 * gold["sse_unscaled"] + 1e-9
    assert np.max(np.abs(v[::97] - np.array(gold["vars_every_97th"]))) <= 1e-6 * gold["scale"]


# ---- f32 compute (BASELINE cfg5) -----------------------------------------------------------------------

def test_f32_mode_matches_f64_oracle_within_f32_tolerances(fiksi, oracle, ctx):
    """precision = 32: f32 residuals / Jacobian / Cholesky, f64 scale + perturbation + storage. Stated
    tolerances vs the f64 oracle: same verdict (converged or not) on >= 99 % of systems, final scaled SSE
    within 1e-3 relative (+1e-7) on 95 %, solved positions of gauge-fixed sketches within 1e-3 * scale on
    90 %, 1e-2 on 99 % (median 1e-5 * scale)."""
    from fiksi_amd import abi, workloads

    o32 = abi.solving_opts(f32=True)
    assert o32.lm.precision == 32 and o32.lm.ftol == 1e-4 and o32.lm.lambda_min == 1e-7
    b = workloads.ring16(2048)
    v, res = ctx.system_solve_batch(b, o32)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    assert np.array_equal(res["scale"], res_o["scale"])  # K0 stays f64
    r_o = oracle.residuals_batch(b, v_o).reshape(len(res), -1)
    conv_o = (r_o * r_o).sum(1) < 1e-4
    conv = res["sse_unscaled"] < 1e-4
    assert (conv == conv_o).mean() >= 0.99
    rel = np.abs(res["sse"] - res_o["sse"]) / (1e-7 + res_o["sse"])
    assert np.percentile(rel, 95) <= 1e-3
    g = workloads.ring16(512, fix_gauge=True)
    vg, rg = ctx.system_solve_batch(g, o32)
    vo, ro = oracle.solve_batch(g, mode=3, nthreads=8)
    ok = (ro["sse"] < 1e-8) & (rg["sse_unscaled"] < 1e-4)
    assert ok.mean() > 0.9
    d = np.abs(vg - vo).reshape(len(rg), -1).max(1) / ro["scale"]
    # both solvers stop anywhere inside SSE < 1e-8: soft directions leave ~1e-3 * scale of slack
    assert np.percentile(d[ok], 99) <= 1e-2 and np.percentile(d[ok], 90) <= 1e-3 and np.median(d[ok]) <= 1e-5
    assert np.array_equal(vg.reshape(len(rg), -1)[:, :4], g["vars"].reshape(len(rg), -1)[:, :4])  # fixed: bit-identical


def test_cfg5_overconstrained_f32_batch(fiksi, oracle, ctx):
    """cfg5 shape: inconsistent ring16 targets (every distance off by up to 2 %), f32, LM damping: no
    system reaches zero residual; the f32 minimum agrees with the f64 oracle's to 1e-4 relative on 95 %."""
    from fiksi_amd import abi, workloads

    b = workloads.ring16(4096, inconsistent=True)
    v, res = ctx.system_solve_batch(b, abi.solving_opts(f32=True))
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    assert np.all(res["exit"] != 0) and np.all(res_o["exit"] != 0)  # nobody hits SSE < 1e-8
    assert np.isin(res["exit"], (1, 2)).mean() >= 0.99               # ftol / step exits, not caps
    rel = np.abs(res["sse"] - res_o["sse"]) / res_o["sse"]
    assert np.percentile(rel, 95) <= 1e-4
    assert np.median(rel) <= 1e-5


# ---- System::analyze -------------------------------------------------------------------------

def test_analyze_verdicts_identical_to_oracle(fiksi, oracle, ctx):
    """Over-constraint detection (analyze/numerical/mod.rs:33-163): lanes own matrix columns and every
    element sees the reference's exact operation sequence, so the per-expression verdict is identical."""
    from fiksi_amd import workloads

    batches = [
        workloads.concat([workloads.quadrilateral(False), workloads.quadrilateral(True), workloads.ring16(50),
                          workloads.hinged_triangles(2, 11), workloads.ring16(20, inconsistent=True)]),
        fiksi.flatten([mixed_sketch(300 + k, fix_some=bool(k % 2)) for k in range(40)]),
    ]
    for b in batches:
        dep = ctx.analyze_batch(b)
        dep_o = oracle.analyze_batch(b)
        assert np.array_equal(dep, dep_o)
    assert ctx.analyze_batch(workloads.quadrilateral(False)).tolist() == [0, 0, 0, 0, 0, 1]


def test_refined_step_follows_the_qr_oracle_more_closely(fiksi, oracle, ctx):
    """fx_lm_opts.solver = FX_STEP_CHOLESKY_REFINED: one corrected-semi-normal-equation refinement per
    step. On arbitrary (ill-conditioned, partly infeasible) sketches the final SSE then agrees with the
    reference's QR-based LM orders of magnitude more closely than the plain normal-equation step."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([mixed_sketch(100 + s, fix_some=s % 3 == 0).flatten() for s in range(160)])
    v_o, r_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
    out = {}
    for solver in (0, 1):
        v, r = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
        ok = ~(np.isnan(r_o["sse"]) | np.isnan(r["sse"]))
        d = (np.abs(r["sse"] - r_o["sse"]) / np.maximum(np.abs(r_o["sse"]), 1e-12))[ok]
        same = ((r["accepted"] == r_o["accepted"]) & (r["trials"] == r_o["trials"]))[ok]
        out[solver] = (float(np.quantile(d, 0.9)), float(same.mean()))
    assert out[1][0] < 1e-4 and out[1][0] < 0.1 * out[0][0], out
    assert out[1][1] >= out[0][1] - 0.01 and out[1][1] > 0.9, out
    # well-conditioned sketches: the refinement changes nothing visible
    ring = workloads.ring16(300)
    v0, r0 = ctx.system_solve_batch(ring)
    v1, r1 = ctx.system_solve_batch(ring, abi.solving_opts(solver=1))
    assert np.array_equal(r0["accepted"], r1["accepted"]) and np.max(np.abs(v0 - v1)) < 1e-8


def test_refined_step_on_the_sparse_path(fiksi, oracle, ctx):
    """Systems beyond one wavefront (fx_sparse.hip) honour FX_STEP_CHOLESKY_REFINED too — it is also what FX_STEP_QR
    runs them with: on large ill-conditioned sketches the final SSE follows the oracle an order of magnitude more
    closely than the plain normal-equation step did in round 2 (p90 1.9e-10 -> 1.4e-11); since round 3 the plain step
    of this path sums shorter lists (finer nested dissection) and sits at 4e-11 itself, so the bar is: the refined step
    is at least as close, and within 1e-9. Well-conditioned sketches keep their path."""
    from fiksi_amd import abi, workloads

    from helpers import random_big_sketch

    b = workloads.concat([random_big_sketch(1000 * 90 + s, 90).flatten() for s in range(24)])
    v_o, r_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
    q = {}
    for solver in (0, 1, 2):
        v, r = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
        ok = np.isfinite(r["sse"]) & np.isfinite(r_o["sse"])
        d = (np.abs(r["sse"] - r_o["sse"]) / np.maximum(np.abs(r_o["sse"]), 1e-12))[ok]
        q[solver] = (float(np.median(d)), float(np.quantile(d, 0.9)), v)
    assert q[1][1] <= q[0][1] and q[1][1] <= 1e-9 and q[1][0] <= q[0][0], q
    assert np.array_equal(q[1][2], q[2][2])  # FX_STEP_QR == the refined step on these
    for big in (workloads.hinged_triangles(2, 64), workloads.large_sketch(300)):
        v, r = ctx.system_solve_batch(big, abi.solving_opts(solver=1))
        vo, ro = oracle.solve_batch(big, mode=3)
        assert np.array_equal(r["accepted"], ro["accepted"]) and np.array_equal(r["trials"], ro["trials"])
        assert np.allclose(r["sse"], ro["sse"], rtol=1e-6, atol=1e-12)


def test_dense_jacobian_entry_point_bit_exact(fiksi, oracle, ctx):
    """fx_eval_residual_dense_jacobian == Problem::calculate_residuals_and_jacobian (subsystem.rs:106-124):
    every variant, fixed variables dropped, and the dense scatter's overwrite of a repeated column (the
    sparse entry point sums it) — bit-identical to the oracle; consistent with the CSR entry point where
    no column repeats."""
    from fiksi_amd import workloads

    b = workloads.concat([mixed_sketch(seed, fix_some=(seed % 2 == 1)).flatten() for seed in range(24)]
                         + [workloads.ring16(5), workloads.quadrilateral()])
    r, blocks = ctx.eval_residual_dense_jacobian(b)
    r_o, blocks_o = oracle.eval_dense_batch(b)
    ang = np.isin(b["expr_tag"], ANGLE_TAGS)
    assert np.array_equal(r[~ang], r_o[~ang]) and np.max(np.abs(r[ang] - r_o[ang])) <= 4e-15
    assert len(blocks) == len(blocks_o)
    for s, (jb, jo) in enumerate(zip(blocks, blocks_o)):
        assert jb.shape == jo.shape and np.array_equal(jb, jo), s
    # against the sparse entry point: equal wherever a row does not read a variable twice
    r2, (rp, ci, vals) = ctx.eval_residual_jacobian(b)
    assert np.array_equal(r, r2)
    n_dup = 0
    for s in range(len(blocks)):
        e0, e1 = int(b["expr_off"][s]), int(b["expr_off"][s + 1])
        for row in range(e0, e1):
            cols, v = ci[rp[row]:rp[row + 1]], vals[rp[row]:rp[row + 1]]
            dense_row = blocks[s][row - e0]
            same = np.array_equal(dense_row[cols], v)
            n_dup += 0 if same else 1
            assert np.count_nonzero(dense_row) <= len(cols)
    assert 0 < n_dup <= 24  # the mixed sketches hold exactly one such row each
    # size query without a device call
    import ctypes as C
    from fiksi_amd import abi
    from fiksi_amd._lib import lib
    st = abi.as_struct(abi.normalize_batch(b))
    total = C.c_uint64(0)
    assert lib.fx_eval_residual_dense_jacobian(None, C.byref(st), None, None, None, C.byref(total)) == 0
    assert total.value == sum(x.size for x in blocks)
