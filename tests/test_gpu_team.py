"""The workgroup ("team") kernels of the sparse path (fx_sparse_team.h, round 3): Systems beyond one wavefront solved
with one plan per STRUCTURE, a whole Levenberg-Marquardt loop per launch (one workgroup per System), larger Systems as
parts + top. What a grouping or a placement decides must not show in the results. All through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def team_kernels(ctx):
    """These tests are about the team kernels: batches of one structure with a small factor would take the grouped kernel's
    sparse build instead (fx_grouped_s.hip; tests/test_gpu_grouped_s.py)."""
    ctx.set_one_structure_builds(False)
    yield
    ctx.set_one_structure_builds(True)


def _singles(ctx, batches, opts=None):
    vs, rs = [], []
    for b in batches:
        v, r = ctx.system_solve_batch(b, opts)
        vs.append(v)
        rs.append(r)
    return np.concatenate(vs), np.concatenate(rs)


def test_the_reference_bench_sketch_of_64_triangles_as_a_batch(fiksi, oracle, ctx):
    """fiksi_bench.rs:46-73, size 64 (258 variables — beyond one wavefront): a batch of them, every one with the
    oracle's step counts and the bench's own check (sum of squared residuals < 1e-4)."""
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(40, 64)
    v, res = ctx.system_solve_batch(b)
    one = workloads.hinged_triangles(1, 64)
    v_o, res_o = oracle.solve_batch(one, mode=3)
    assert np.all(res["accepted"] == res_o["accepted"][0]) and np.all(res["trials"] == res_o["trials"][0])
    assert np.all(res["exit"] == res_o["exit"][0]) and np.all(res["scale"] == res_o["scale"][0])
    assert np.all(res["sse_unscaled"] < 1e-4)
    assert np.allclose(res["sse"], res_o["sse"][0], rtol=1e-6, atol=1e-12)
    nv = len(v_o)
    assert np.max(np.abs(v.reshape(40, nv) - v_o[None, :])) < 1e-9 * res_o["scale"][0]


@pytest.mark.parametrize("solver", [0, 1])
def test_a_group_of_one_structure_equals_its_systems_solved_alone(fiksi, ctx, solver):
    """Systems of one structure share a plan and every launch (grid y / one workgroup each): bit for bit what each
    gives on its own; and the Systems of a mixed batch find their groups whatever their order."""
    from fiksi_amd import abi, workloads

    opts = abi.solving_opts(solver=solver)
    a = [workloads.large_sketch(150, seed=7 + k) for k in range(5)]       # 300 variables, one structure
    c = [workloads.large_sketch(260, seed=40 + k) for k in range(3)]      # another one
    h = [workloads.hinged_triangles(1, 64)]
    v1, r1 = _singles(ctx, a + c + h, opts)
    v2, r2 = ctx.system_solve_batch(workloads.concat(a + c + h), opts)
    assert np.array_equal(v1.view(np.uint64), v2.view(np.uint64))
    assert np.array_equal(r1, r2)
    order = [a[0], c[0], a[1], h[0], c[1], a[2], a[3], c[2], a[4]]
    v3, r3 = ctx.system_solve_batch(workloads.concat(order), opts)
    v4, r4 = _singles(ctx, order, opts)
    assert np.array_equal(v3.view(np.uint64), v4.view(np.uint64)) and np.array_equal(r3, r4)


def test_mid_size_sketches_follow_the_oracle(fiksi, oracle, ctx):
    """Chains of 40 ... 700 points (80 ... 1 400 variables: the wide kernel, then the team kernel with the factor in
    LDS, then in HBM): the oracle's accepted / trial counts on every one, SSE within SURVEY 8c's tolerance."""
    from fiksi_amd import workloads

    for n_pts in (40, 100, 300, 700):
        b = workloads.concat([workloads.large_sketch(n_pts, seed=7 + k) for k in range(3)])
        v, res = ctx.system_solve_batch(b)
        v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=3)
        assert np.array_equal(res["accepted"], res_o["accepted"]) and np.array_equal(res["trials"], res_o["trials"]), n_pts
        assert np.array_equal(res["exit"], res_o["exit"]) and np.array_equal(res["scale"], res_o["scale"])
        assert np.all(np.abs(res["sse"] - res_o["sse"]) <= 1e-10 + 1e-6 * np.abs(res_o["sse"])), n_pts


def test_parts_and_top_agree_with_one_workgroup(fiksi, oracle, ctx):
    """A large System alone is spread over the chip (parts + top: each segment's values in LDS, the parts' share of the
    top's sums handed over through a contribution buffer); eight or more of one structure keep to one workgroup each
    (one launch). Same plan and trial sequence; the top's entries are summed in another order, so the two agree to
    rounding, not bit for bit — and both follow the oracle's path."""
    from fiksi_amd import abi, workloads

    one = workloads.large_sketch(1200, seed=11)   # 2 400 columns: has a parts schedule
    many = workloads.concat([workloads.large_sketch(1200, seed=11) for _ in range(8)])
    v_o, r_o = oracle.solve_batch(one, mode=3)
    for solver in (0, 1):
        opts = abi.solving_opts(solver=solver)
        v1, r1 = ctx.system_solve_batch(one, opts)
        v8, r8 = ctx.system_solve_batch(many, opts)
        assert r1["trials"][0] > 3
        for f in ("accepted", "trials", "exit", "ncomp"):
            assert np.all(r8[f] == r1[f][0]) and r1[f][0] == r_o[f][0], f
        assert np.all(r8["scale"] == r1["scale"][0])
        assert np.all(np.abs(r8["sse"] - r1["sse"][0]) <= 1e-9 * r1["sse"][0] + 1e-14)
        assert abs(r1["sse"][0] - r_o["sse"][0]) <= 1e-10 + 1e-6 * r_o["sse"][0]
        assert np.max(np.abs(v8.reshape(8, -1) - v1[None, :])) <= 1e-8 * r1["scale"][0]
        # the eight copies of a group are the same bits
        assert np.all(v8.reshape(8, -1).view(np.uint64) == v8.reshape(8, -1)[0].view(np.uint64))
        # two runs of the spread-out solve are the same bits too (every sum has a fixed order)
        v1b, r1b = ctx.system_solve_batch(one, opts)
        assert np.array_equal(v1.view(np.uint64), v1b.view(np.uint64)) and np.array_equal(r1, r1b)


def test_more_distinct_large_structures_than_the_context_keeps_plans(fiksi, oracle, ctx):
    """One one-shot call with 11 different large structures (the context keeps 8 plans): a plan this call still uses
    must not be evicted under it — LM (grouped solves) and L-BFGS / SinglePass (one host loop per System, worker
    threads). Outcomes against the oracle."""
    from fiksi_amd import abi, workloads

    sketches = [workloads.large_sketch(70 + 3 * k, seed=100 + k) for k in range(11)]
    b = workloads.concat(sketches)
    for opts, kind in ((abi.solving_opts(), "lm"), (abi.solving_opts(decomposer=1), "single_pass"), (abi.solving_opts(optimizer=1), "lbfgs")):
        v, res = ctx.system_solve_batch(b, opts)
        v2, res2 = ctx.system_solve_batch(b, opts)  # plans seen before (those that were kept)
        assert np.array_equal(v.view(np.uint64), v2.view(np.uint64)) and np.array_equal(res, res2)
        if kind == "lbfgs":
            continue  # (L-BFGS on these: compared by its own tests; here the point is that nothing was freed under it)
        v_o, res_o = oracle.solve_batch(b, mode=3, nthreads=8) if kind == "lm" else oracle.solve_single_pass_batch(b, nthreads=8)
        assert np.array_equal(res["scale"], res_o["scale"])
        same = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"])
        assert same.mean() >= 0.9, (res["accepted"], res_o["accepted"])
        assert np.all(np.abs(res["sse"][same] - res_o["sse"][same]) <= 1e-9 + 1e-5 * np.abs(res_o["sse"][same]))


def test_single_pass_with_the_qr_step_refines_large_systems(fiksi, ctx):
    """FX_STEP_QR stops at one wavefront; larger Systems of the batch take FX_STEP_CHOLESKY_REFINED — also under
    SinglePass, where the block walker solves them (round 2 ran them with the plain step there)."""
    from fiksi_amd import abi, workloads

    from helpers import random_big_sketch

    big = [random_big_sketch(4000 + s, 300).flatten() for s in range(4)]   # beyond 512 variables, small blocks: the walker
    small = [workloads.hinged_triangles(3, 4)]
    b = workloads.concat(big + small)
    v1, r1 = ctx.system_solve_batch(b, abi.solving_opts(decomposer=1, solver=1))
    v2, r2 = ctx.system_solve_batch(b, abi.solving_opts(decomposer=1, solver=2))
    n_big = sum(len(x["var_off"]) - 1 for x in big)
    nv_big = int(b["var_off"][n_big])
    assert np.array_equal(v1[:nv_big].view(np.uint64), v2[:nv_big].view(np.uint64))
    assert np.array_equal(r1[:n_big], r2[:n_big])


@pytest.mark.parametrize("n_tri", [16, 31, 64])
def test_narrow_teams_give_the_bits_of_the_sixteen_wavefront_team(fiksi, ctx, n_tri):
    """From 768 Systems of a structure on, a System's workgroup is 2, 4 or 8 wavefronts instead of 16 (fx_sparse.hip:
    team_waves_for): each wavefront walks every n-th list of a level's schedule, the sums of squares keep their 1 024 strided
    partial sums and their tree. Which wavefront walks a column changes nothing in its arithmetic — a System solved in a batch
    of 800 (narrow teams) carries the bits of the same System in a batch of 3 (16 wavefronts): 66, 126 and 258 variables,
    plain and refined step."""
    from fiksi_amd import abi, workloads

    ctx.set_wide_routing(0)
    try:
        for solver in (0, 1):
            o = abi.solving_opts(solver=solver)
            vb, rb = ctx.system_solve_batch(workloads.hinged_triangles(800, n_tri), o)
            vs, rs = ctx.system_solve_batch(workloads.hinged_triangles(3, n_tri), o)
            nv = 2 + 4 * n_tri
            assert np.array_equal(vb[:nv].view(np.uint64), vs[:nv].view(np.uint64)), solver
            assert rb[0].tobytes() == rs[0].tobytes() and rb[799].tobytes() == rs[0].tobytes(), solver
            assert rb["sse_unscaled"][0] < 1e-6
    finally:
        ctx.set_wide_routing(-1)
