"""Per-expression VALUES of the oracle pinned by an independent restatement: the geometric definition of each of the
eleven expressions (fiksi/src/constraints/expressions.rs:291-874, doc comments of constraints/mod.rs:317-891) written
in mpmath at 60 digits, and its gradient obtained by high-precision numerical differentiation of that definition — so
neither the residual formulas nor the analytic partials of the oracle (or of the kernels, which equal the oracle bit
for bit: tests/test_gpu_parity.py) are trusted. The reference itself holds no known-answer vectors for this boundary
(its tests are first-order consistency checks at 1e-3, expressions.rs:1196-1509); this closes that gap to ~1e-12."""
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")
mp.mp.dps = 60

from helpers import Lcg


def _ang(x, y):
    return mp.atan2(y, x)


def _wrap(a):
    if a > mp.pi:
        return a - 2 * mp.pi
    if a < -mp.pi:
        return a + 2 * mp.pi
    return a


def _dist(ax, ay, bx, by):
    return mp.sqrt((ax - bx) ** 2 + (ay - by) ** 2)


def _cross(ux, uy, vx, vy):
    return ux * vy - uy * vx


# residual(v, param) by definition; v in variable_indices order (expressions.rs:48-182)
DEFS = {
    0: (2, lambda v, p: v[1] - v[0]),                                                      # VariableVariableEquality
    1: (4, lambda v, p: _dist(v[0], v[1], v[2], v[3]) - p),                                # PointPointDistance
    2: (6, lambda v, p: _wrap(_ang(v[4] - v[2], v[5] - v[3]) - _ang(v[0] - v[2], v[1] - v[3])) - p),  # PointPointPointAngle (at p2)
    3: (6, lambda v, p: _cross(v[4] - v[2], v[5] - v[3], v[0] - v[2], v[1] - v[3])),      # PointLineIncidence
    4: (6, lambda v, p: _cross(v[4] - v[2], v[5] - v[3], v[0] - v[2], v[1] - v[3]) / _dist(v[2], v[3], v[4], v[5]) - p),  # PointLineDistance
    5: (5, lambda v, p: _dist(v[0], v[1], v[2], v[3]) - v[4]),                             # PointCircleIncidence
    6: (8, lambda v, p: _dist(v[4], v[5], v[6], v[7]) - _dist(v[0], v[1], v[2], v[3])),    # SegmentSegmentLengthEquality
    7: (8, lambda v, p: _wrap(_ang(v[6] - v[4], v[7] - v[5]) - _ang(v[2] - v[0], v[3] - v[1])) - p),  # LineLineAngle
    8: (8, lambda v, p: _cross(v[6] - v[4], v[7] - v[5], v[2] - v[0], v[3] - v[1])),       # LineLineParallelism: w x u
    9: (8, lambda v, p: (v[6] - v[4]) * (v[2] - v[0]) + (v[7] - v[5]) * (v[3] - v[1])),    # LineLinePerpendicularity: w . u
    10: (7, lambda v, p: abs(v[0] * (v[3] - v[5]) + v[2] * (v[5] - v[1]) + v[4] * (v[1] - v[3])) / _dist(v[0], v[1], v[2], v[3]) - v[6]),  # LineCircleTangency
}


@pytest.mark.parametrize("tag", sorted(DEFS))
def test_oracle_values_match_the_definition(oracle, tag):
    k, f = DEFS[tag]
    g = Lcg(1234 + tag)
    for trial in range(40):
        scale = [1.0, 1e-3, 1e3][trial % 3]
        v = [scale * g.u(-10, 10) for _ in range(k)]
        if tag in (5, 10):
            v[-1] = scale * g.u(0.5, 4)  # a radius
        p = g.u(-2, 2) if tag in (2, 7) else scale * g.u(0.5, 6)
        r, grad = oracle.expr_eval(tag, v, p)
        mv = [mp.mpf(x) for x in v]
        mp_r = f(mv, mp.mpf(p))
        # (the error of a difference is that of its terms: coordinates for distances, their products for the cross /
        # dot kinds, pi for the angle kinds)
        size = max(abs(float(mp_r)), 3.2 if tag in (2, 7) else max(abs(x) for x in v) ** (2 if tag in (3, 8, 9) else 1), 1e-300)
        assert abs(mp.mpf(r) - mp_r) <= 8e-16 * size, (tag, trial, r, mp_r)
        for i in range(k):
            def fi(t, i=i):
                w = list(mv)
                w[i] = t
                return f(w, mp.mpf(p))
            d = mp.diff(fi, mv[i], h=mp.mpf(10) ** -25 * max(abs(mv[i]), 1))
            tol = 1e-11 * max(abs(float(d)), max(abs(float(x)) for x in grad), 1e-300)
            assert abs(mp.mpf(float(grad[i])) - d) <= tol, (tag, trial, i, grad[i], d)
