"""Decomposer::SinglePass: oracle pinning, the product's host-side decomposition (CPU) and
the block-by-block device solve against the oracle (GPU).

Reference: fiksi/src/analyze/graph/equations.rs:186-550 (matching + strongly connected blocks),
fiksi/src/assemble/mod.rs:169-210 (one LM per block, results written through). The reference's own
tests for this arm are fixed.rs (three sketches, both decomposers) and triangles.rs:9-37; they assert
thresholds, which is what pins the oracle here — the reference cannot be run in this image."""
import math

import numpy as np
import pytest

from helpers import mixed_sketch, random_sketch

RESIDUAL_THRESHOLD = 1e-4  # fiksi/src/tests/mod.rs


def rms(x):
    x = np.asarray(list(x), dtype=np.float64)
    return float(np.sqrt(np.mean(x * x))) if len(x) else 0.0


def _reference_sketches(F):
    """The four sketches the reference solves with Decomposer::SinglePass."""
    out = {}
    s = F.System()  # fixed.rs:10-43
    p0, p1, p2 = (F.elements.Point.create(s, x, y) for x, y in ((0., 0.), (1., .5), (2., 1.)))
    p1.fix(s)
    for a, b in ((p0, p1), (p0, p2), (p1, p2)):
        F.constraints.PointPointDistance.create(s, a, b, 1.)
    out["triangle_fixed_point"] = s
    s = F.System()  # fixed.rs:47-82
    p0 = F.elements.Point.create(s, 0., 0.)
    center = F.elements.Point.create(s, 4., 3.)
    radius = F.elements.Length.create(s, 1.)
    circle = F.elements.Circle.create(s, center, radius)
    p0.fix(s)
    center.fix(s)
    F.constraints.PointCircleIncidence.create(s, p0, circle)
    out["fixed_circle"] = s
    s = F.System()  # fixed.rs:94-127
    p0, p1, p2, p3 = (F.elements.Point.create(s, x, y) for x, y in ((0., 0.), (1., .5), (2., 1.), (5., 5.)))
    p3.fix(s)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    F.constraints.PointPointDistance.create(s, p1, p2, 1.)
    F.constraints.PointPointCoincidence.create(s, p2, p3)
    out["fixed_with_coincidence"] = s
    s = F.System()  # triangles.rs:9-37
    p0, p1, p2 = (F.elements.Point.create(s, x, y) for x, y in ((0., 0.), (1., .5), (2., 1.)))
    for a, b in ((p0, p1), (p0, p2), (p1, p2)):
        F.constraints.PointPointDistance.create(s, a, b, 1.)
    out["single_triangle"] = s
    return out


def _oracle_blocks(oracle, b, system=0):
    vc = b.get("var_comp")
    v0, v1 = int(b["var_off"][system]), int(b["var_off"][system + 1])
    comps = [int(c) for c in (vc[v0:v1] if vc is not None else [0] * (v1 - v0)) if c != 0xFFFF]
    out = []
    for c in range(max(comps, default=-1) + 1):
        out += [(c, r, v) for r, v in oracle.single_pass_units(b, system, c)]
    return out


# ---- CPU: the oracle against the reference's own SinglePass tests --------------------------------

def test_oracle_maximum_matching_known_answer(oracle):
    """equations.rs:574-601: three variables, expressions on (0,1), (0,2), (1): a perfect matching of
    size 3, every pair an edge of the graph, no expression used twice."""
    exprs = [[0, 1], [0, 2], [1]]
    card, a_to_b = oracle.maximum_matching(3, exprs)
    assert card == 3
    assert sorted(a_to_b) == [0, 1, 2]
    assert all(v in exprs[e] for v, e in enumerate(a_to_b))
    # masking a variable out (fixed) leaves a maximum matching of the rest
    card, a_to_b = oracle.maximum_matching(3, exprs, free=[0, 2])
    assert card == 2 and a_to_b[1] == -1


def test_oracle_single_pass_meets_reference_test_thresholds(oracle, fiksi):
    sk = _reference_sketches(fiksi)
    for name, s in sk.items():
        b = s.flatten()
        v, res = oracle.solve_single_pass_batch(b, trial_cap=4096)
        assert rms(oracle.residuals_batch(b, v)) < RESIDUAL_THRESHOLD, name
        fx = b["var_fixed"] == 1
        assert np.array_equal(v[fx], b["vars"][fx]), name  # fixed values stay bit-identical
    b = sk["fixed_circle"].flatten()
    v, _ = oracle.solve_single_pass_batch(b, trial_cap=4096)
    assert abs(v[4] - 5.) < RESIDUAL_THRESHOLD  # the radius, fixed.rs:76-80
    b = sk["fixed_with_coincidence"].flatten()
    v, _ = oracle.solve_single_pass_batch(b, trial_cap=4096)
    assert math.hypot(v[4] - 5., v[5] - 5.) < RESIDUAL_THRESHOLD  # p2 on p3, fixed.rs:121-125


def test_oracle_single_pass_solve_order_of_fixed_with_coincidence(oracle, fiksi):
    """fixed.rs:84-93 describes the order a structural decomposition should find: p2 onto the fixed p3
    first (two 1x1 blocks), then the rest. The distance chain p0-p1-p2 is one under-determined block."""
    b = _reference_sketches(fiksi)["fixed_with_coincidence"].flatten()
    blocks = oracle.single_pass_units(b)
    assert [(sorted(r), v) for r, v in blocks] == [([3], [5]), ([2], [4]), ([0, 1], [0, 1, 2, 3])]


def test_oracle_single_pass_equals_none_on_a_single_block(oracle, fiksi):
    """A sketch whose expressions form one block is the same LM problem either way (same free
    variables in the same order, rows possibly permuted): identical step counts, same answer."""
    from fiksi_amd import workloads

    b = workloads.ring16(8)
    blocks = oracle.single_pass_units(b, 3)
    assert len(blocks) == 1 and len(blocks[0][0]) == 32 and blocks[0][1] == list(range(32))
    v1, r1 = oracle.solve_single_pass_batch(b, trial_cap=4096)
    v0, r0 = oracle.solve_batch(b, mode=3)
    assert np.array_equal(r1["accepted"], r0["accepted"]) and np.array_equal(r1["trials"], r0["trials"])
    assert np.max(np.abs(v1 - v0)) < 1e-9


# ---- CPU: the product's decomposition (fx_single_pass_blocks, host-only) -------------------------

def test_blocks_match_oracle_on_random_sketches(fiksi, oracle):
    from fiksi_amd import abi

    shapes = set()
    for seed in range(250):
        b = random_sketch(seed).flatten()
        mine = abi.single_pass_blocks(b, 0)
        assert mine == _oracle_blocks(oracle, b), seed
        shapes |= {(len(r), len(v)) for _, r, v in mine}
    assert len(shapes) > 25  # the generator really produces blocks of many shapes


def test_blocks_match_oracle_on_workloads(fiksi, oracle):
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.hinged_triangles(2, 7), workloads.ring16(2), workloads.quadrilateral(),
                          workloads.large_sketch(400)])
    for s in range(len(b["var_off"]) - 1):
        assert abi.single_pass_blocks(b, s) == _oracle_blocks(oracle, b, s), s
    for seed in range(20):
        b = mixed_sketch(seed, fix_some=seed % 2 == 0).flatten()
        assert abi.single_pass_blocks(b, 0) == _oracle_blocks(oracle, b)


def test_blocks_cover_matched_expressions_once(fiksi):
    """Structural invariants: within a component every expression is in at most one block, every free
    variable is in at most one block overall, a block never has fewer free variables than expressions
    (its matching is perfect on the expression side), and fixed variables never appear. (Across
    components an expression CAN appear twice: the stale component label of graph.rs:211-222 lets one
    component's matching claim an expression that is listed under another.)"""
    from fiksi_amd import abi

    for seed in range(100):
        b = random_sketch(1000 + seed).flatten()
        blocks = abi.single_pass_blocks(b, 0)
        rows = [(c, r) for c, rs, _ in blocks for r in rs]
        vs = [v for _, _, v_ in blocks for v in v_]
        assert len(rows) == len(set(rows)) and len(vs) == len(set(vs))
        assert all(len(v) >= len(r) >= 1 for _, r, v in blocks)
        assert all(b["var_fixed"][v] == 0 for v in vs)
        assert all(v == sorted(v) for _, _, v in blocks)


def test_blocks_argument_errors(fiksi):
    from fiksi_amd import abi, workloads
    with pytest.raises(IndexError):
        abi.single_pass_blocks(workloads.quadrilateral(), 1)  # only system 0 exists
    import ctypes as C
    from fiksi_amd._lib import lib
    st = abi.as_struct(abi.normalize_batch(workloads.quadrilateral()))
    assert lib.fx_single_pass_blocks(C.byref(st), 1, None, None, None, None, None, None) == -1  # FX_ERR_INVALID
    nb = C.c_uint32(99)
    assert lib.fx_single_pass_blocks(C.byref(st), 0, C.byref(nb), None, None, None, None, None) == 0  # sizes only
    assert nb.value == len(abi.single_pass_blocks(workloads.quadrilateral(), 0))
    empty = {k: np.zeros(0, dtype=v.dtype) for k, v in workloads.quadrilateral().items()}
    empty["var_off"] = np.zeros(2, dtype=np.uint32)
    empty["expr_off"] = np.zeros(2, dtype=np.uint32)
    assert abi.single_pass_blocks(empty, 0) == []


# ---- GPU: block-by-block solve against the oracle ------------------------------------------------

def _sp_opts(fiksi, **kw):
    from fiksi_amd import abi

    return abi.solving_opts(decomposer=1, **kw)


@pytest.mark.gpu
def test_gpu_single_pass_reference_sketches(fiksi, oracle, ctx):
    for name, s in _reference_sketches(fiksi).items():
        b = s.flatten()
        v, res = ctx.system_solve_batch(b, _sp_opts(fiksi))
        v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096)
        assert res["accepted"][0] == res_o["accepted"][0] and res["trials"][0] == res_o["trials"][0], name
        assert res["exit"][0] == res_o["exit"][0] and res["ncomp"][0] == res_o["ncomp"][0], name
        assert np.max(np.abs(v - v_o)) < 1e-9, name
        assert rms(oracle.residuals_batch(b, v)) < RESIDUAL_THRESHOLD, name


@pytest.mark.gpu
def test_gpu_single_pass_hinged_triangles_bit_level(fiksi, oracle, ctx):
    """Chains of triangles decompose into one 3x6 block and then 3x4 blocks hanging off solved points:
    small, well-conditioned blocks, so the device's normal-equation step follows the reference's QR
    step to rounding — identical step counts, positions to 1e-9."""
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(64, 12)
    v, res = ctx.system_solve_batch(b, _sp_opts(fiksi))
    v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
    assert np.array_equal(res["scale"], res_o["scale"])
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["trials"], res_o["trials"])
    assert np.array_equal(res["exit"], res_o["exit"])
    assert np.allclose(res["sse0"], res_o["sse0"], rtol=1e-12, atol=0)
    assert np.max(np.abs(v - v_o)) < 1e-9
    # and it differs from Decomposer::None (otherwise this test would prove nothing)
    v_n, res_n = ctx.system_solve_batch(b)
    assert not np.array_equal(res_n["trials"], res["trials"])


@pytest.mark.gpu
def test_gpu_single_pass_random_sketches(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    flats = [random_sketch(seed).flatten() for seed in range(300)]
    b = workloads.concat(flats)
    v, res = ctx.system_solve_batch(b, _sp_opts(fiksi))
    v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
    assert np.array_equal(res["ncomp"], res_o["ncomp"])
    assert np.array_equal(res["scale"], res_o["scale"])
    # the reference would spin forever on a NaN block (lm.rs:115-191 has no exit for it); the oracle
    # stops at the trial cap, the device reports FX_EXIT_NAN — both leave the block's variables alone
    nan = np.isnan(res_o["sse"])
    assert np.array_equal(nan, np.isnan(res["sse"])) and nan.mean() < 0.1
    same = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"]) & ~nan
    assert same.mean() > 0.8
    d = np.abs(res["sse"][same] - res_o["sse"][same])
    assert np.all(d <= 1e-9 + 0.25 * np.abs(res_o["sse"][same])), d.max()
    assert np.median(d / (1e-12 + np.abs(res_o["sse"][same]))) <= 1e-4
    # sse0 sums the blocks' starting SSE; later blocks start from what earlier ones produced
    d0 = np.abs(res["sse0"][same] - res_o["sse0"][same])
    assert np.all(d0 <= 1e-9 + 0.25 * np.abs(res_o["sse0"][same])), d0.max()
    assert np.median(d0 / (1e-12 + np.abs(res_o["sse0"][same]))) <= 1e-6
    assert np.mean((res["exit"] == 0) == (res_o["exit"] == 0)) >= 0.95
    fx = b["var_fixed"] == 1
    assert np.array_equal(v[fx], b["vars"][fx])
    # variables no block owns keep their (unperturbed) input value bit for bit — only matched / block
    # variables are ever written back (assemble/mod.rs:201-207)
    owned = np.zeros(len(v), dtype=bool)
    from fiksi_amd import abi
    for s in range(len(flats)):
        v0 = int(b["var_off"][s])
        for _, _, vs in abi.single_pass_blocks(b, s):
            owned[v0 + np.asarray(vs, dtype=np.int64)] = True
    assert np.array_equal(v[~owned], b["vars"][~owned])
    assert np.array_equal(v_o[~owned], b["vars"][~owned])


@pytest.mark.gpu
def test_gpu_single_pass_one_block_equals_none(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.ring16(500)
    v1, r1 = ctx.system_solve_batch(b, _sp_opts(fiksi))
    v0, r0 = ctx.system_solve_batch(b)
    assert np.array_equal(r1["accepted"], r0["accepted"]) and np.array_equal(r1["trials"], r0["trials"])
    # rows are summed in block order instead of ascending; the gauge-free ring lets that rounding drift
    assert np.max(np.abs(v1 - v0)) < 1e-7
    assert np.allclose(r1["sse"], r0["sse"], rtol=1e-6, atol=1e-14)


@pytest.mark.gpu
def test_gpu_single_pass_device_batch_and_repeatability(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.concat([workloads.hinged_triangles(40, 8), workloads.ring16(10), workloads.quadrilateral()])
    db = ctx.upload(b)
    db.system_solve(_sp_opts(fiksi))
    v1, r1 = db.get_vars(), db.get_results()
    db.system_solve()  # Decomposer::None on the same resident batch
    vn = db.get_vars()
    db.system_solve(_sp_opts(fiksi))
    v2, r2 = db.get_vars(), db.get_results()
    assert np.array_equal(v1, v2) and np.array_equal(r1, r2)
    assert not np.array_equal(v1, vn)
    v_h, r_h = ctx.system_solve_batch(b, _sp_opts(fiksi))
    assert np.array_equal(v_h, v1)
    db.free()


@pytest.mark.gpu
def test_gpu_single_pass_large_system_with_a_large_block(fiksi, oracle, ctx):
    """A closed chain of 40 points with only neighbour distances: 80 variables, one block of 40
    expressions and 80 free variables — the System is too large for LDS AND its block is too large for
    one wavefront, so it is solved block by block through the sparse path."""
    from fiksi_amd import System, abi, constraints, elements, workloads

    s = System()
    n = 40
    pts = [elements.Point.create(s, 10. * math.cos(2 * math.pi * i / n) + 0.3 * math.sin(7. * i),
                                 10. * math.sin(2 * math.pi * i / n) + 0.3 * math.cos(5. * i)) for i in range(n)]
    for i in range(n):
        constraints.PointPointDistance.create(s, pts[i], pts[(i + 1) % n], 1.6)
    flat = s.flatten()
    blocks = abi.single_pass_blocks(flat, 0)
    assert len(blocks) == 1 and len(blocks[0][2]) == 80
    b = workloads.concat([flat, workloads.hinged_triangles(2, 4)])
    v, res = ctx.system_solve_batch(b, _sp_opts(fiksi))
    v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096)
    assert np.array_equal(res["accepted"], res_o["accepted"]) and np.array_equal(res["exit"], res_o["exit"])
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-5, atol=1e-12)
    assert np.max(np.abs(v - v_o)) < 1e-6


@pytest.mark.gpu
def test_gpu_single_pass_large_sketch_is_walked_on_the_device(fiksi, oracle, ctx):
    """A sketch beyond the LDS limits whose blocks are all small: one wavefront walks its blocks with
    the System-wide vectors in HBM scratch; mixed with small Systems in the same batch."""
    from fiksi_amd import workloads

    b = workloads.concat([workloads.hinged_triangles(3, 5), workloads.large_sketch(150), workloads.ring16(2)])
    v, res = ctx.system_solve_batch(b, _sp_opts(fiksi))
    v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096)
    assert np.array_equal(res["ncomp"], res_o["ncomp"])
    assert np.array_equal(res["accepted"], res_o["accepted"])
    assert np.array_equal(res["exit"], res_o["exit"])
    assert np.allclose(res["sse"], res_o["sse"], rtol=1e-5, atol=1e-12)
    assert np.max(np.abs(v - v_o)) < 1e-6


@pytest.mark.gpu
def test_gpu_single_pass_f32(fiksi, oracle, ctx):
    from fiksi_amd import workloads

    b = workloads.hinged_triangles(64, 10)
    v, res = ctx.system_solve_batch(b, _sp_opts(fiksi, f32=True))
    v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
    r = oracle.residuals_batch(b, v)
    assert rms(r) < 1e-3
    assert np.median(np.abs(v - v_o)) < 1e-4
    assert np.array_equal(res["ncomp"], res_o["ncomp"])


@pytest.mark.gpu
def test_gpu_single_pass_through_the_system_api(fiksi, oracle):
    F = fiksi
    s = _reference_sketches(F)["fixed_with_coincidence"]
    before = s.flatten()
    s.solve(F.SolvingOptions(decomposer=F.Decomposer.SinglePass))
    v_o, _ = oracle.solve_single_pass_batch(before, trial_cap=4096)
    assert np.max(np.abs(s.flatten()["vars"] - v_o)) < 1e-9
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD
