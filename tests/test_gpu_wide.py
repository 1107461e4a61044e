"""Medium components (65 .. 128 free variables): the LDS-resident wide kernel (fx_wide.hip) against the
oracle, and its hand-over points to the fused kernel below and the sparse path above."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def wide_kernel(ctx):
    """These tests are about the wide kernel: pin it (by default a small batch of such Systems goes to the team kernels,
    which finish one of them in half the time — fx_ctx_set_wide_routing; the last test covers that choice)."""
    ctx.set_wide_routing(1)
    ctx.set_one_structure_builds(False)  # (batches of one structure would take fx_grouped_s.hip: tests/test_gpu_grouped_s.py)
    yield
    ctx.set_wide_routing(-1)
    ctx.set_one_structure_builds(True)


def _rms(x):
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.mean(x * x))) if len(x) else 0.0


@pytest.mark.parametrize("n_tri", [16, 20, 25, 31])
def test_hinged_chains_just_above_the_one_wavefront_limit(fiksi, oracle, ctx, n_tri):
    """Chains of n hinged triangles: 4n + 2 variables (66 .. 126). Same LM path as the oracle: identical
    accepted-step and trial counts, solved positions to 1e-8 (normal equations vs the reference's QR)."""
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(24, n_tri)
    assert 64 < int(b["var_off"][1]) <= 128
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


def test_wide_and_narrow_and_large_systems_share_a_batch(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.concat([workloads.ring16(6), workloads.hinged_triangles(3, 20), workloads.large_sketch(100),
                          workloads.hinged_triangles(2, 5), workloads.hinged_triangles(2, 30), workloads.quadrilateral()])
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=4)
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["exit"], res_o["exit"])
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-5, atol=1e-12)
    db = ctx.upload(b)
    db.system_solve()
    assert np.array_equal(db.get_vars(), v)  # resident path: same answer, deterministic
    db.system_solve()
    assert np.array_equal(db.get_vars(), v)
    db.free()


def test_wide_multi_component_fixed_points_and_snapshot_quirk(fiksi, oracle, ctx):
    """Two components in one System — a 70-variable chain and a small triangle — plus fixed points: the
    shared LCG stream continues across components (assemble/mod.rs:47), fixed values stay bit-identical,
    and the second component is solved against the pre-solve snapshot (quirk Q2)."""
    F = fiksi
    s = F.System()
    pts = [F.elements.Point.create(s, 1.1 * i, 0.35 * ((i * 7) % 5)) for i in range(35)]
    for i in range(34):
        F.constraints.PointPointDistance.create(s, pts[i], pts[i + 1], 1.3)
    for i in range(33):
        F.constraints.PointPointDistance.create(s, pts[i], pts[i + 2], 2.2)
    pts[0].fix(s)
    q = [F.elements.Point.create(s, 50. + i, 3. * (i % 2)) for i in range(3)]
    for a, c in ((0, 1), (0, 2), (1, 2)):
        F.constraints.PointPointDistance.create(s, q[a], q[c], 1.5)
    flat = s.flatten()
    assert flat["var_comp"].max() == 1
    v, res = ctx.system_solve_batch(flat)
    v_o, res_o = oracle.solve_batch(flat, mode=3)
    assert res["ncomp"][0] == res_o["ncomp"][0] == 2
    assert res["accepted"][0] == res_o["accepted"][0] and res["trials"][0] == res_o["trials"][0]
    assert np.max(np.abs(v - v_o)) < 1e-8
    assert np.array_equal(v[:2], flat["vars"][:2])


def test_wide_kernel_other_modes_take_the_general_paths(fiksi, oracle, ctx):
    """f32, L-BFGS and SinglePass on a medium System are served by the paths that implement them (sparse
    path / block walker); results stay consistent with the oracle."""
    from fiksi_amd import abi, workloads

    b = workloads.hinged_triangles(4, 20)
    v_sp, res_sp = ctx.system_solve_batch(b, abi.solving_opts(decomposer=1))
    v_sp_o, res_sp_o = oracle.solve_single_pass_batch(b, trial_cap=4096)
    assert np.array_equal(res_sp["accepted"], res_sp_o["accepted"]) and np.max(np.abs(v_sp - v_sp_o)) < 1e-8
    v_l, res_l = ctx.system_solve_batch(b, abi.solving_opts(optimizer=1))
    v_l_o, res_l_o = oracle.solve_batch(b, mode=7)
    assert np.all(res_l["sse"] <= np.maximum(2.0 * res_l_o["sse"], 1e-6))
    v_f, res_f = ctx.system_solve_batch(b, abi.solving_opts(f32=True))
    assert _rms(oracle.residuals_batch(b, v_f)) < 1e-4


def test_wide_kernel_l2_entry_point_and_limits(fiksi, oracle, ctx):
    """fx_lm_solve_batch (no scaling, no perturbation) on medium Systems; 129 free variables go to the
    sparse path."""
    F = fiksi
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(8, 24)
    v, res = ctx.lm_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=0, nthreads=4)
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.max(np.abs(v - v_o)) < 1e-8
    big = workloads.hinged_triangles(1, 32)  # 130 variables: one component of 130 free columns
    assert int(big["var_off"][1]) == 130
    v2, res2 = ctx.system_solve_batch(big)
    v2_o, res2_o = oracle.solve_batch(big, mode=3)
    assert res2["accepted"][0] == res2_o["accepted"][0]
    assert np.max(np.abs(v2 - v2_o)) < 1e-7


def test_f32_requests_on_medium_and_large_systems_take_the_f64_device_kernels(fiksi, ctx):
    """Every path beyond the one-wavefront limits computes in f64. A request for f32 compute (`precision = 32`, with its
    ftol / max_outer) therefore gives, on Systems of 65 ... 128 free variables (wide kernel) and on large Systems cut into
    small blocks by SinglePass (the walker), exactly the bits of the same options with `precision = 64` — and no longer
    goes through the host-driven loop of the sparse path (20 000 sketches of 66 variables: seconds -> milliseconds)."""
    import time

    from fiksi_amd import abi, workloads

    b = workloads.hinged_triangles(400, 16)  # 66 variables each
    for decomposer in (0, 1):
        o32 = abi.solving_opts(f32=True, decomposer=decomposer)
        o64 = abi.solving_opts(f32=True, decomposer=decomposer)
        o64.lm.precision = 64
        t0 = time.perf_counter()
        v32, r32 = ctx.system_solve_batch(b, o32)
        dt = time.perf_counter() - t0
        v64, r64 = ctx.system_solve_batch(b, o64)
        assert np.array_equal(v32.view(np.uint64), v64.view(np.uint64)) and r32.tobytes() == r64.tobytes()
        assert np.all(r32["sse_unscaled"] < 1e-4)
        assert dt < 0.5, dt  # the host-driven path took ~1 ms per System


@pytest.mark.parametrize("n_tri", [16, 24, 31])
def test_either_home_of_a_medium_component_follows_the_oracle(fiksi, oracle, ctx, n_tri):
    """Components of 65 ... 128 columns run on the wide kernel or on the team kernels (fx_ctx_set_wide_routing; by default
    by measured cost: few of them -> team kernels, thousands -> wide kernel, from ~112 columns on always the team kernels).
    Either way the oracle's accepted / trial counts and SSE; the two agree to rounding, and each is deterministic."""
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(12, n_tri)
    v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8)
    got = {}
    for wide in (1, 0, -1):
        ctx.set_wide_routing(wide)
        v, res = ctx.system_solve_batch(b)
        v2, res2 = ctx.system_solve_batch(b)
        assert np.array_equal(v.view(np.uint64), v2.view(np.uint64)) and np.array_equal(res, res2)
        for f in ("scale", "accepted", "trials", "exit"):
            assert np.array_equal(res[f], res_o[f]), (wide, f)
        assert np.allclose(res["sse"], res_o["sse"], rtol=1e-6, atol=1e-12)
        assert np.max(np.abs(v - v_o)) < 1e-8
        got[wide] = v
    assert np.max(np.abs(got[1] - got[0])) < 1e-9
    # twelve such Systems are few: the default took the team kernels
    assert np.array_equal(got[-1].view(np.uint64), got[0].view(np.uint64))
