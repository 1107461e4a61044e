"""The reference's own end-to-end solver tests (fiksi/src/tests/{basic,triangles,singular,fixed,
magnitude}.rs and the bench spot-check), written against the drop-in mirror API and run on the GPU:
same sketches, same thresholds. Each test also solves the identical System with the CPU oracle and
compares. triangles.rs:10-14 runs its sketch under all three decomposers; so does test_single_triangle."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RESIDUAL_THRESHOLD = 1e-4  # fiksi/src/tests/mod.rs:13


def rms(values):
    v = np.asarray(list(values), dtype=np.float64)
    return float(np.sqrt(np.mean(v * v)))


@pytest.fixture()
def F(fiksi, ctx):
    """The mirror API bound to the test context."""
    import fiksi_amd.system as sysmod

    sysmod._default_ctx = ctx
    return fiksi


def oracle_check(oracle, fiksi, s_before_flat, s, tol=1e-6):
    """Solve the same flat System with the oracle and compare per-constraint residuals."""
    v_o, res_o = oracle.solve_batch(s_before_flat, mode=3)
    r_gpu = oracle.residuals_batch(s_before_flat, s.flatten()["vars"])
    r_ref = oracle.residuals_batch(s_before_flat, v_o)
    scale = res_o["scale"][0]
    assert s.last_result["scale"] == scale
    return r_gpu, r_ref, res_o


# ---- basic.rs ------------------------------------------------------------------------------------

def test_coincident_points(F, oracle):  # basic.rs:10-33
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    coincidence = F.constraints.PointPointCoincidence.create(s, p0, p1)
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms([coincidence.calculate_residual(s)]) < RESIDUAL_THRESHOLD
    a, b = p0.get_value(s), p1.get_value(s)
    assert math.hypot(a[0] - b[0], a[1] - b[1]) < RESIDUAL_THRESHOLD


def test_underconstrained_triangle(F, oracle):  # basic.rs:36-53
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    p2 = F.elements.Point.create(s, 2., 1.)
    angle0 = F.constraints.PointPointPointAngle.create(s, p0, p1, p2, math.radians(40))
    angle1 = F.constraints.PointPointPointAngle.create(s, p1, p2, p0, math.radians(80))
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms([angle0.calculate_residual(s), angle1.calculate_residual(s)]) < RESIDUAL_THRESHOLD
    r_gpu, r_ref, res_o = oracle_check(oracle, F, before, s)
    assert s.last_result["accepted"] == res_o["accepted"][0]


def test_overconstrained_triangle_line_incidence(F, oracle):  # basic.rs:56-87
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    p2 = F.elements.Point.create(s, 2., 1.)
    p3 = F.elements.Point.create(s, 3., 1.5)
    line0 = F.elements.Line.create(s, p2, p3)
    angle0 = F.constraints.PointPointPointAngle.create(s, p0, p1, p2, math.radians(40))
    angle1 = F.constraints.PointPointPointAngle.create(s, p1, p2, p0, math.radians(80))
    angle2 = F.constraints.PointPointPointAngle.create(s, p2, p0, p1, math.radians(100))
    incidence = F.constraints.PointLineIncidence.create(s, p1, line0)
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    rms_angles = rms([angle0.calculate_residual(s), angle1.calculate_residual(s), angle2.calculate_residual(s)])
    assert rms_angles >= RESIDUAL_THRESHOLD  # geometrically impossible
    assert incidence.calculate_residual(s) < RESIDUAL_THRESHOLD
    r_gpu, r_ref, res_o = oracle_check(oracle, F, before, s)
    assert abs(s.last_result["sse"] - res_o["sse"][0]) <= 1e-6 * res_o["sse"][0] + 1e-10


def test_overconstrained_analysis(F, oracle):  # basic.rs:90-112
    s = F.System()
    p0 = F.elements.Point.create(s, 0.123, 0.1)
    p1 = F.elements.Point.create(s, 1.2, 0.)
    p2 = F.elements.Point.create(s, -0.5, 1.1)
    p3 = F.elements.Point.create(s, 1.599, 1.2)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    F.constraints.PointPointDistance.create(s, p0, p2, 1.5)
    F.constraints.PointPointDistance.create(s, p1, p3, 1.7)
    F.constraints.PointPointDistance.create(s, p2, p3, 1.2)
    F.constraints.PointPointDistance.create(s, p1, p2, 2.)
    p0p3 = F.constraints.PointPointDistance.create(s, p0, p3, 5.)
    analysis = s.analyze()
    assert analysis.overconstrained == [p0p3.as_any_constraint()]


def test_triangle_inscribed_circle(F, oracle):  # basic.rs:116-149
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    p2 = F.elements.Point.create(s, 1.5, 1.)
    p3 = F.elements.Point.create(s, 2.8, 1.5)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    F.constraints.PointPointDistance.create(s, p0, p2, 1.)
    F.constraints.PointPointDistance.create(s, p1, p2, 1.)
    line0 = F.elements.Line.create(s, p0, p1)
    line1 = F.elements.Line.create(s, p0, p2)
    line2 = F.elements.Line.create(s, p1, p2)
    circle_radius = F.elements.Length.create(s, 1.)
    circle = F.elements.Circle.create(s, p3, circle_radius)
    F.constraints.LineCircleTangency.create(s, line0, circle)
    F.constraints.LineCircleTangency.create(s, line1, circle)
    F.constraints.LineCircleTangency.create(s, line2, circle)
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD
    r_gpu, r_ref, res_o = oracle_check(oracle, F, before, s)
    assert rms(r_ref) < RESIDUAL_THRESHOLD


def test_two_connected_components(F, oracle):  # basic.rs:152-170
    s = F.System()
    p0 = F.elements.Point.create(s, 0.123, 0.1)
    p1 = F.elements.Point.create(s, 1.2, 0.)
    p2 = F.elements.Point.create(s, -0.5, 1.1)
    p3 = F.elements.Point.create(s, 1.599, 1.2)
    p0p1 = F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    p2p3 = F.constraints.PointPointDistance.create(s, p2, p3, 1.2)
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms([p0p1.calculate_residual(s), p2p3.calculate_residual(s)]) < RESIDUAL_THRESHOLD
    assert s.last_result["ncomp"] == 2
    # one Rng shared by the two components (assemble/mod.rs:47): positions equal the oracle's
    v_o, res_o = oracle.solve_batch(before, mode=3)
    assert res_o["ncomp"][0] == 2
    assert np.max(np.abs(s.flatten()["vars"] - v_o)) < 1e-9


# ---- triangles.rs --------------------------------------------------------------------------------

@pytest.mark.parametrize("decomposer", ["NONE", "SinglePass", "RecursiveAssembly"])
def test_single_triangle(F, oracle, decomposer):  # triangles.rs:9-37, all three arms
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    p2 = F.elements.Point.create(s, 2., 1.)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    F.constraints.PointPointDistance.create(s, p0, p2, 1.)
    F.constraints.PointPointDistance.create(s, p1, p2, 1.)
    s.solve(F.SolvingOptions(decomposer=F.Decomposer[decomposer]))
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD


def test_unimplemented_options_are_reported_unsupported(F):
    from fiksi_amd._lib import FiksiError

    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    with pytest.raises(FiksiError) as e:  # RecursiveAssembly plans from elements: builder API only, not a flat batch
        F.default_context().system_solve_batch(s.flatten(), F.abi.solving_opts(decomposer=2))
    assert e.value.code == -6
    with pytest.raises(FiksiError) as e:  # L-BFGS is f64 only
        F.default_context().system_solve_batch(s.flatten(), F.abi.solving_opts(optimizer=1, f32=True))
    assert e.value.code == -6


def test_connected_triangles(F, oracle):  # triangles.rs:40-70
    s = F.System()
    p = [F.elements.Point.create(s, float(i), 0.5 * i) for i in range(6)]
    F.constraints.PointPointPointAngle.create(s, p[5], p[0], p[1], math.radians(-135))
    F.constraints.PointPointPointAngle.create(s, p[1], p[2], p[3], math.radians(-120))
    F.constraints.PointPointPointAngle.create(s, p[3], p[4], p[5], math.radians(-115))
    for a, b, d in ((0, 1, 7.), (1, 2, 5.), (2, 3, 9.), (3, 4, 8.), (4, 5, 6.), (5, 0, 7.)):
        F.constraints.PointPointDistance.create(s, p[a], p[b], d)
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD
    r_gpu, r_ref, res_o = oracle_check(oracle, F, before, s)
    assert rms(r_ref) < RESIDUAL_THRESHOLD


def test_hinged_triangles(F, oracle):  # triangles.rs:73-104
    s = F.System()
    pts = [(0.5, 0.), (1.1, 0.5), (2.1, 1.), (3.1, 1.5), (4.1, 2.), (5.1, 2.5), (6.1, 3.)]
    p = [F.elements.Point.create(s, x, y) for x, y in pts]
    for a, b in ((0, 1), (0, 2), (1, 2), (0, 3), (0, 4), (3, 4), (0, 5), (0, 6), (5, 6)):
        F.constraints.PointPointDistance.create(s, p[a], p[b], 1.)
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD


# ---- singular.rs ---------------------------------------------------------------------------------

def test_collinear_points(F, oracle):  # singular.rs:19-40 (needs the perturbation)
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 3., 0.)
    p2 = F.elements.Point.create(s, 6., 0.)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    F.constraints.PointPointDistance.create(s, p0, p2, 1.)
    F.constraints.PointPointDistance.create(s, p1, p2, 1.)
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD
    r_gpu, r_ref, res_o = oracle_check(oracle, F, before, s)
    assert rms(r_ref) < RESIDUAL_THRESHOLD


# ---- fixed.rs ------------------------------------------------------------------------------------

@pytest.mark.parametrize("decomposer", ["NONE", "SinglePass"])  # fixed.rs loops over both
def test_single_triangle_with_fixed_point(F, oracle, decomposer):  # fixed.rs:10-43
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    p2 = F.elements.Point.create(s, 2., 1.)
    p1.fix(s)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    F.constraints.PointPointDistance.create(s, p0, p2, 1.)
    F.constraints.PointPointDistance.create(s, p1, p2, 1.)
    s.solve(F.SolvingOptions(decomposer=F.Decomposer[decomposer]))
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD
    assert p1.get_value(s) == (1., 0.5)  # bit-identical


@pytest.mark.parametrize("decomposer", ["NONE", "SinglePass"])  # fixed.rs loops over both
def test_fixed_point_and_circle_center_incidence(F, oracle, decomposer):  # fixed.rs:47-82
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    center = F.elements.Point.create(s, 4., 3.)
    radius = F.elements.Length.create(s, 1.)
    circle = F.elements.Circle.create(s, center, radius)
    p0.fix(s)
    center.fix(s)
    F.constraints.PointCircleIncidence.create(s, p0, circle)
    s.solve(F.SolvingOptions(decomposer=F.Decomposer[decomposer]))
    assert p0.get_value(s) == (0., 0.)
    assert center.get_value(s) == (4., 3.)
    assert abs(radius.get_value(s) - 5.) < RESIDUAL_THRESHOLD


@pytest.mark.parametrize("decomposer", ["NONE", "SinglePass"])  # fixed.rs loops over both
def test_fixed_with_coincidence(F, oracle, decomposer):  # fixed.rs:94-127
    s = F.System()
    p0 = F.elements.Point.create(s, 0., 0.)
    p1 = F.elements.Point.create(s, 1., 0.5)
    p2 = F.elements.Point.create(s, 2., 1.)
    p3 = F.elements.Point.create(s, 5., 5.)
    p3.fix(s)
    F.constraints.PointPointDistance.create(s, p0, p1, 1.)
    F.constraints.PointPointDistance.create(s, p1, p2, 1.)
    F.constraints.PointPointCoincidence.create(s, p2, p3)
    s.solve(F.SolvingOptions(decomposer=F.Decomposer[decomposer]))
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < RESIDUAL_THRESHOLD
    x, y = p2.get_value(s)
    assert math.hypot(x - 5., y - 5.) < RESIDUAL_THRESHOLD


# ---- magnitude.rs --------------------------------------------------------------------------------

def test_large_order_of_magnitude(F, oracle):  # magnitude.rs:13-35
    FACTOR = 1e20
    s = F.System()
    p0 = F.elements.Point.create(s, 1.5 * FACTOR, 6.5 * FACTOR)
    p1 = F.elements.Point.create(s, 3.2 * FACTOR, 0.8 * FACTOR)
    p2 = F.elements.Point.create(s, 2.2 * FACTOR, -1.5 * FACTOR)
    F.constraints.PointPointDistance.create(s, p0, p1, 5. * FACTOR)
    F.constraints.PointPointDistance.create(s, p0, p2, 3. * FACTOR)
    F.constraints.PointPointDistance.create(s, p1, p2, 4. * FACTOR)
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < FACTOR * RESIDUAL_THRESHOLD


def _four_points(F, s, FACTOR):
    p0 = F.elements.Point.create(s, 1.5 * FACTOR, 6.5 * FACTOR)
    p1 = F.elements.Point.create(s, 3.2 * FACTOR, 0.8 * FACTOR)
    p2 = F.elements.Point.create(s, 2.2 * FACTOR, -1.5 * FACTOR)
    p3 = F.elements.Point.create(s, 1.2 * FACTOR, 0.5 * FACTOR)
    ppd = [F.constraints.PointPointDistance.create(s, p0, p1, 5. * FACTOR),
           F.constraints.PointPointDistance.create(s, p1, p2, 4. * FACTOR),
           F.constraints.PointPointDistance.create(s, p2, p3, 3. * FACTOR),
           F.constraints.PointPointDistance.create(s, p3, p1, 1. * FACTOR)]
    line0 = F.elements.Line.create(s, p0, p1)
    line1 = F.elements.Line.create(s, p2, p3)
    return ppd, line0, line1


def test_distance_and_angle(F, oracle):  # magnitude.rs:45-83
    FACTOR = 1e10
    s = F.System()
    ppd, line0, line1 = _four_points(F, s, FACTOR)
    angle = F.constraints.LineLineAngle.create(s, line0, line1, math.radians(30))
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in ppd) < FACTOR * RESIDUAL_THRESHOLD
    assert abs(angle.calculate_residual(s)) < RESIDUAL_THRESHOLD


def test_metric_and_singular(F, oracle):  # magnitude.rs:93-133
    FACTOR = 1e7
    s = F.System()
    ppd, line0, line1 = _four_points(F, s, FACTOR)
    llp = F.constraints.LineLineParallelism.create(s, line0, line1)
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in ppd) < FACTOR * RESIDUAL_THRESHOLD
    assert abs(llp.calculate_residual(s)) < FACTOR * FACTOR * RESIDUAL_THRESHOLD
    r_gpu, r_ref, res_o = oracle_check(oracle, F, before, s)


def test_near_degenerate_isosceles_triangle(F, oracle):  # magnitude.rs:143-166
    FACTOR = 1e13
    s = F.System()
    p0 = F.elements.Point.create(s, 1.5 * FACTOR, 6.5 * FACTOR)
    p1 = F.elements.Point.create(s, 3.2 * FACTOR, 0.8 * FACTOR)
    p2 = F.elements.Point.create(s, 2.2, -1.5)
    F.constraints.PointPointDistance.create(s, p0, p1, 4. * FACTOR + 1.)
    F.constraints.PointPointDistance.create(s, p1, p2, 4. * FACTOR + 1.)
    F.constraints.PointPointDistance.create(s, p0, p2, 1.)
    s.solve(F.SolvingOptions.DEFAULT)
    assert rms(c.calculate_residual(s) for c in s.get_constraint_handles()) < FACTOR * RESIDUAL_THRESHOLD


# ---- fiksi_svg_tests example sketch and the bench spot-check -----------------------------------------

def test_circle_triangle_line_example(F, oracle):  # examples/fiksi_svg_tests/src/main.rs:9-44
    s = F.System()
    p1 = F.elements.Point.create(s, 10., 0.)
    p2 = F.elements.Point.create(s, 20., 10.)
    p3 = F.elements.Point.create(s, 30., -10.)
    p4 = F.elements.Point.create(s, -40., -50.)
    p5 = F.elements.Point.create(s, 40., -50.)
    F.constraints.PointPointPointAngle.create(s, p1, p2, p3, math.radians(40))
    F.constraints.PointPointPointAngle.create(s, p2, p3, p1, math.radians(70))
    F.constraints.PointPointDistance.create(s, p1, p2, 70.)
    side1 = F.elements.Line.create(s, p1, p2)
    F.elements.Line.create(s, p2, p3)
    side3 = F.elements.Line.create(s, p1, p3)
    radius = F.elements.Length.create(s, 5.)
    circle = F.elements.Circle.create(s, p3, radius)
    F.constraints.LineCircleTangency.create(s, side1, circle)
    line = F.elements.Line.create(s, p4, p5)
    F.constraints.LineLineAngle.create(s, side3, line, math.radians(-90))
    F.constraints.PointLineIncidence.create(s, p3, line)
    F.constraints.PointPointDistance.create(s, p3, p4, 40.)
    F.constraints.PointPointDistance.create(s, p4, p5, 80.)
    before = s.flatten()
    s.solve(F.SolvingOptions.DEFAULT)
    v_o, res_o = oracle.solve_batch(before, mode=3)
    r_gpu = oracle.residuals_batch(before, s.flatten()["vars"])
    r_ref = oracle.residuals_batch(before, v_o)
    # same outcome as the reference algorithm on this mixed sketch
    assert abs(rms(r_gpu) - rms(r_ref)) <= 1e-6 * max(1.0, rms(r_ref)) + 1e-7


def test_bench_hinged_triangles_spot_check(F, oracle):  # fiksi/benches/fiksi_bench.rs:15-40,65-72
    from fiksi_amd import workloads

    for n in (1, 4, 16, 64):  # the reference's sizes; 16 and 64 go through the sparse large-sketch path
        b = workloads.hinged_triangles(1, n)
        s = F.System()
        hinge = F.elements.Point.create(s, 0., 0.)
        for t in range(n):
            a = F.elements.Point.create(s, -1., float(t))
            c = F.elements.Point.create(s, 1., float(t))
            F.constraints.PointPointDistance.create(s, hinge, a, 2.)
            F.constraints.PointPointDistance.create(s, hinge, c, 2.)
            F.constraints.PointPointDistance.create(s, a, c, 3.)
        assert np.array_equal(s.flatten()["vars"], b["vars"])
        s.solve(F.SolvingOptions.DEFAULT)
        sse = sum(c.calculate_residual(s) ** 2 for c in s.get_constraint_handles())
        assert sse < 1e-4
        _, res_o = oracle.solve_batch(b, mode=3)
        assert s.last_result["accepted"] == res_o["accepted"][0]
