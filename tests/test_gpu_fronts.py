"""The multifrontal build of the large-component path (fx_front.h, fx_front_plan.h: the elimination tree cut into fronts of at
most 15 columns, a front factored in the registers of one row of 16 lanes) against the oracle and against the column walkers
it replaces (fx_sparse_team.h, fx_ctx_set_sparse_fronts(ctx, 0)). Both are the normal-equation step of lm.rs:28-63 with the
sums in another order: the oracle's accepted-step / trial counts and exit codes, variables to 1e-9 of the scale. All through
the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def team_paths(ctx):
    """These tests are about the Systems the team kernels take: batches of one structure with a small factor would go to the
    grouped kernel's sparse build instead (fx_grouped_s.hip)."""
    ctx.set_one_structure_builds(False)
    yield
    ctx.set_one_structure_builds(True)
    ctx.set_sparse_fronts(True)


def _both(ctx, b, opts=None):
    out = {}
    for fronts in (False, True):
        ctx.set_sparse_fronts(fronts)
        out[fronts] = ctx.system_solve_batch(b, opts)
    ctx.set_sparse_fronts(True)
    return out[True], out[False]


@pytest.mark.parametrize("case", ["hinged_64", "hinged_64_x40", "hinged_16", "large_150_x5", "large_1500"])
def test_fronts_follow_the_oracle_and_the_walkers(fiksi, oracle, ctx, case):
    from fiksi_amd import workloads

    b = {"hinged_64": lambda: workloads.hinged_triangles(1, 64), "hinged_64_x40": lambda: workloads.hinged_triangles(40, 64),
         "hinged_16": lambda: workloads.hinged_triangles(1, 16),
         "large_150_x5": lambda: workloads.concat([workloads.large_sketch(150, seed=7 + k) for k in range(5)]),
         "large_1500": lambda: workloads.large_sketch(1500, seed=5)}[case]()
    (v, res), (v_w, res_w) = _both(ctx, b)
    assert not np.array_equal(v.view(np.uint64), v_w.view(np.uint64)), "the two builds sum in different orders: equal bits mean one of them did not run"
    for f in ("accepted", "trials", "exit", "ncomp"):
        assert np.array_equal(res[f], res_w[f]), f
    assert np.array_equal(res["scale"], res_w["scale"]) and np.array_equal(res["sse0"], res_w["sse0"])
    assert np.max(np.abs(v - v_w)) <= 1e-9 * res["scale"].max()
    assert np.allclose(res["sse"], res_w["sse"], rtol=1e-6, atol=1e-12)
    if case != "large_1500":  # (the oracle takes minutes there: the walkers, held to it elsewhere, stand in)
        one = b if case != "hinged_64_x40" else workloads.hinged_triangles(1, 64)
        v_o, res_o = oracle.solve_batch(one, mode=3, nthreads=8)
        n = len(res_o)
        for f in ("accepted", "trials", "exit"):
            assert np.array_equal(res[f][:n], res_o[f]), f
        assert np.allclose(res["sse"][:n], res_o["sse"], rtol=1e-6, atol=1e-12)
        assert np.max(np.abs(v[:len(v_o)] - v_o)) <= 1e-9 * res_o["scale"].max()
    assert np.array_equal(res["sse_unscaled"] < 1e-4, res_w["sse_unscaled"] < 1e-4)  # the bench's verdict, fiksi_bench.rs:65-72
    if case != "large_1500":  # (that sketch's noisy start ends in a local minimum, on every path)
        assert np.all(res["sse_unscaled"] < 1e-4)


def test_cfg2_with_fronts_is_on_the_oracle_s_path(fiksi, ctx):
    """BASELINE cfg2 through the parts + top build of the fronts: the fixture's 16 accepted steps / 89 trials (tests/golden/
    cfg2_oracle.json, as tests/test_gpu_parity.py checks for the default path), and the walkers' variables to 1e-9."""
    import json
    import os

    from fiksi_amd import workloads

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cfg2_oracle.json")))
    b = workloads.large_sketch(5000)
    (v, res), (v_w, res_w) = _both(ctx, b)
    r = res[0]
    assert (int(r["accepted"]), int(r["trials"]), int(r["exit"])) == (gold["accepted"], gold["trials"], gold["exit"])
    assert abs(r["sse"] - gold["sse"]) <= 1e-9 + 1e-6 * gold["sse"]
    assert np.max(np.abs(v[::97] - np.array(gold["vars_every_97th"]))) <= 1e-6 * gold["scale"]
    assert not np.array_equal(v.view(np.uint64), v_w.view(np.uint64))
    assert np.max(np.abs(v - v_w)) <= 1e-9 * gold["scale"]


def test_two_solves_give_the_same_bits(fiksi, ctx):
    """Every sum of the fronts has a fixed order (records, children one after the other, LDS additions in program order)."""
    from fiksi_amd import workloads

    for b in (workloads.hinged_triangles(3, 64), workloads.large_sketch(1700, seed=2)):
        v1, r1 = ctx.system_solve_batch(b)
        v2, r2 = ctx.system_solve_batch(b)
        assert np.array_equal(v1.view(np.uint64), v2.view(np.uint64)) and r1.tobytes() == r2.tobytes()


def test_what_the_fronts_do_not_cover_keeps_the_walkers(fiksi, oracle, ctx):
    """The refined step, SinglePass blocks beyond a row of lanes, L-BFGS and structures with wide separators run as before."""
    from fiksi_amd import abi, workloads

    b = workloads.hinged_triangles(2, 64)
    for o in (abi.solving_opts(solver=1), abi.solving_opts(optimizer=1)):
        (v, res), (v_w, res_w) = _both(ctx, b, o)
        assert np.array_equal(v.view(np.uint64), v_w.view(np.uint64)) and res.tobytes() == res_w.tobytes()


def test_the_launch_level_lambda_ladder_is_the_sequential_loop(fiksi, ctx):
    """A large System alone: 1 ... 8 lambda trials per launch (fx_ctx_set_sparse_fronts' ranks) — every variable, counter,
    exit code and SSE the bits of one trial per launch, also under a trial cap that falls inside a round of ranks."""
    from fiksi_amd import abi, workloads

    try:
        for b, kw in ((workloads.large_sketch(5000), {}), (workloads.large_sketch(1700, seed=2), {}), (workloads.large_sketch(1700, seed=2), {"max_trials": 14}),
                      (workloads.large_sketch(1700, seed=2), {"max_trials": 15})):
            o = abi.solving_opts(**kw)
            ctx.set_sparse_fronts(True, 1)
            v1, r1 = ctx.system_solve_batch(b, o)
            assert int(r1["trials"][0]) > int(r1["accepted"][0]) + 3  # (there ARE rejected trials to run side by side)
            for ranks in (2, 3, 4, 6, 8, 0):
                ctx.set_sparse_fronts(True, ranks)
                v, r = ctx.system_solve_batch(b, o)
                assert np.array_equal(v.view(np.uint64), v1.view(np.uint64)), (kw, ranks)
                assert r.tobytes() == r1.tobytes(), (kw, ranks, r, r1)
            if "max_trials" in kw:
                assert int(r1["trials"][0]) == kw["max_trials"] and int(r1["exit"][0]) == 4  # FX_EXIT_TRIAL_CAP
    finally:
        ctx.set_sparse_fronts(True, 0)
