"""Shared builders for the test-suite."""
import numpy as np

from fiksi_amd import System, constraints, elements
from fiksi_amd.workloads import LcgVec


class Lcg:
    """Scalar reference LCG (fiksi/src/rand.rs) for test-data generation."""

    def __init__(self, seed):
        self.v = LcgVec(np.array([seed], dtype=np.uint64))

    def f(self):
        return float(self.v.next_f64()[0])

    def u(self, lo, hi):
        return lo + (hi - lo) * self.f()


def mixed_sketch(seed: int, fix_some: bool = False) -> System:
    """A sketch that uses every one of the eleven constraint kinds at least once (plus lines,
    circles, a shared endpoint so one expression reads the same variable twice, and optionally
    fixed elements)."""
    g = Lcg(seed)
    s = System()
    P = [elements.Point.create(s, g.u(-10, 10), g.u(-10, 10)) for _ in range(10)]
    L = [elements.Line.create(s, P[0], P[1]), elements.Line.create(s, P[2], P[3]),
         elements.Line.create(s, P[4], P[5]), elements.Line.create(s, P[1], P[6])]
    rad = elements.Length.create(s, g.u(1, 4))
    circ = elements.Circle.create(s, P[7], rad)
    if fix_some:
        P[0].fix(s)
        rad.fix(s)
    constraints.PointPointDistance.create(s, P[0], P[1], g.u(2, 8))
    constraints.PointPointDistance.create(s, P[2], P[3], g.u(2, 8))
    constraints.PointPointPointAngle.create(s, P[0], P[1], P[2], g.u(-2, 2))
    constraints.PointLineIncidence.create(s, P[8], L[0])
    constraints.PointLineDistance.create(s, P[9], L[1], g.u(-3, 3))
    constraints.PointCircleIncidence.create(s, P[6], circ)
    # segments share P[1]: the same variable appears twice in one expression
    constraints.SegmentSegmentLengthEquality.create(s, P[0], P[1], P[1], P[6])
    constraints.LineLineAngle.create(s, L[0], L[1], g.u(-2, 2))
    constraints.LineLineParallelism.create(s, L[1], L[2])
    constraints.LineLinePerpendicularity.create(s, L[2], L[3])
    constraints.LineCircleTangency.create(s, L[2], circ)
    constraints.PointPointCoincidence.create(s, P[8], P[9])
    constraints.PointPointDistance.create(s, P[4], P[7], g.u(2, 8))
    constraints.PointPointDistance.create(s, P[5], P[3], g.u(2, 8))
    return s


def csr_to_dense(row_ptr, col, vals, rows, ncols):
    d = np.zeros((rows, ncols))
    for r in range(rows):
        for p in range(int(row_ptr[r]), int(row_ptr[r + 1])):
            d[r, int(col[p])] += vals[p]
    return d


def random_sketch(seed: int, angles: bool = True) -> System:
    """(angles=False: the two kinds whose residual goes through atan2 — PointPointPointAngle, LineLineAngle —
    become distances, so that every operation of the solve is IEEE-deterministic.)
    A sketch with random topology: 3..12 points, lines and circles over them, 2..14 constraints of
    random kinds between random (distinct) elements, a few fixed elements — so the constraint graph
    falls into several components, is under-, well- and over-constrained in places, and the SinglePass
    decomposition produces blocks of many shapes."""
    g = Lcg(seed)

    def pick(seq, k=1):
        pool = list(seq)
        out = []
        for _ in range(k):
            out.append(pool.pop(int(g.u(0, len(pool) - 1e-9))))
        return out[0] if k == 1 else out

    s = System()
    P = [elements.Point.create(s, g.u(-10, 10), g.u(-10, 10)) for _ in range(int(g.u(4, 12.99)))]
    L = [elements.Line.create(s, *pick(P, 2)) for _ in range(int(g.u(2, 4.99)))]
    C = [elements.Circle.create(s, pick(P), elements.Length.create(s, g.u(1, 4))) for _ in range(int(g.u(0, 2.99)))]
    for p in P:
        if g.u(0, 1) < 0.15:
            p.fix(s)
    for _ in range(int(g.u(2, 14.99))):
        k = int(g.u(0, 10.99))
        if not angles and k in (2, 7):
            k = 0
        if k == 0:
            constraints.PointPointDistance.create(s, *pick(P, 2), g.u(2, 8))
        elif k == 1:
            constraints.PointPointCoincidence.create(s, *pick(P, 2))
        elif k == 2:
            constraints.PointPointPointAngle.create(s, *pick(P, 3), g.u(-2, 2))
        elif k == 3:
            constraints.PointLineIncidence.create(s, pick(P), pick(L))
        elif k == 4:
            constraints.PointLineDistance.create(s, pick(P), pick(L), g.u(-3, 3))
        elif k == 5 and C:
            constraints.PointCircleIncidence.create(s, pick(P), pick(C))
        elif k == 6:
            a, b, c, d = pick(P, 4)
            constraints.SegmentSegmentLengthEquality.create(s, a, b, c, d)
        elif k == 7:
            constraints.LineLineAngle.create(s, *pick(L, 2), g.u(-2, 2))
        elif k == 8:
            constraints.LineLineParallelism.create(s, *pick(L, 2))
        elif k == 9:
            constraints.LineLinePerpendicularity.create(s, *pick(L, 2))
        elif k == 10 and C:
            constraints.LineCircleTangency.create(s, pick(L), pick(C))
    return s


def random_big_sketch(seed: int, n_points: int) -> System:
    """Like random_sketch, scaled up: n_points points (wide / sparse-path sizes), a locally connected
    constraint graph (neighbours in index order, so components stay sparse), a few lines and circles,
    a few fixed points, occasionally a second component."""
    g = Lcg(seed)
    s = System()
    P = [elements.Point.create(s, 3.0 * i + g.u(-1, 1), g.u(-6, 6)) for i in range(n_points)]
    near = lambda i, w=4: P[max(0, min(n_points - 1, i + int(g.u(-w, w + 0.99))))]
    L = []
    for _ in range(max(2, n_points // 8)):
        i = int(g.u(0, n_points - 2))
        L.append(elements.Line.create(s, P[i], P[i + 1]))
    C = [elements.Circle.create(s, P[int(g.u(0, n_points - 0.01))], elements.Length.create(s, g.u(1, 4)))
         for _ in range(max(1, n_points // 20))]
    for p in P:
        if g.u(0, 1) < 0.05:
            p.fix(s)
    split = int(g.u(0.3, 0.7) * n_points) if g.u(0, 1) < 0.4 else -1  # no constraint crosses the split
    for i in range(n_points - 1):
        if i + 1 == split:
            continue
        constraints.PointPointDistance.create(s, P[i], P[i + 1], g.u(2, 4))
    for _ in range(int(0.8 * n_points)):
        i = int(g.u(1, n_points - 2))
        lo, hi = (0, split) if 0 < split and i < split else (max(split, 0), n_points)
        pick = lambda: P[max(lo, min(hi - 1, i + int(g.u(-3, 3.99))))]
        k = int(g.u(0, 6.99))
        a, b, c = pick(), pick(), pick()
        if k == 0 and a is not b:
            constraints.PointPointDistance.create(s, a, b, g.u(2, 8))
        elif k == 1 and len({id(a), id(b), id(c)}) == 3:
            constraints.PointPointPointAngle.create(s, a, b, c, g.u(-2, 2))
        elif k == 2:
            constraints.PointLineDistance.create(s, a, L[int(g.u(0, len(L) - 0.01))], g.u(-3, 3))
        elif k == 3:
            constraints.PointCircleIncidence.create(s, a, C[int(g.u(0, len(C) - 0.01))])
        elif k == 4 and len(L) > 1:
            l1, l2 = L[int(g.u(0, len(L) - 0.01))], L[int(g.u(0, len(L) - 0.01))]
            if l1 is not l2:
                constraints.LineLineAngle.create(s, l1, l2, g.u(-1, 1))
        elif k == 5 and a is not b:
            constraints.PointPointCoincidence.create(s, a, b)
    return s


def compare_outcomes(b, v, res, v_o, res_o, oracle, tight: bool):
    """GPU solve against the oracle over ALL Systems of a batch — none is dropped for taking a different path.
    Structure-level quantities exactly; then
      * Systems on the same path (equal accepted / trial counts): final SSE within SURVEY 8c's
        1e-10 + 1e-6 * SSE (`tight`), per-constraint unscaled residuals within 1e-7 * scale (+ 1e-4 sqrt(SSE));
      * Systems on a different path: the same verdict (reached SSE < 1e-8 or not) and a bounded SSE difference.
    A System whose step turns non-finite ends with FX_EXIT_NAN here, while the reference would double lambda for
    ever (the oracle stops it at its trial cap): both leave the component at its start point.
    Returns (fraction on the same path, fraction with the same verdict)."""
    assert np.array_equal(res["ncomp"], res_o["ncomp"])
    assert np.array_equal(res["scale"], res_o["scale"])
    fx = b["var_fixed"] == 1
    assert np.array_equal(v[fx], b["vars"][fx])  # fixed variables never move (fiksi/src/tests/fixed.rs:36-40)
    nan_gpu, nan_ref = res["exit"] == 5, res_o["exit"] == 4
    same = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"]) & (res["exit"] == res_o["exit"])
    finite = np.isfinite(res["sse"]) & np.isfinite(res_o["sse"])
    assert np.array_equal(np.isnan(res_o["sse0"]), np.isnan(res["sse0"]))
    d = np.abs(res["sse"] - res_o["sse"])
    sp = same & finite
    tol_same = (1e-10 + 1e-6 * np.abs(res_o["sse"])) if tight else (1e-9 + 0.25 * np.abs(res_o["sse"]))
    within = d[sp] <= tol_same[sp]
    if tight:
        # an ulp in one atan2 can grow along a path through a flat valley (under-determined, partly infeasible
        # sketches): SURVEY 8c's bound on >= 99.5 % of the Systems on the same path, a hundred times it on all
        assert within.mean() >= 0.995, within.mean()
        assert np.all(d[sp] <= 100. * tol_same[sp]), (d[sp].max(), np.nonzero(sp)[0][np.argmax(d[sp] - 100. * tol_same[sp])])
    else:
        assert np.all(within), (d[sp].max(), np.nonzero(sp)[0][np.argmax(d[sp] - tol_same[sp])])
    other = ~same & finite & ~nan_gpu & ~nan_ref
    # the verdict: the bench's own predicate (fiksi_bench.rs:65-72) — sum of squared unscaled residuals < 1e-4 —
    # on the solved variables of either side (the exit code only tells about the last component / block)
    r = oracle.residuals_batch(b, v)
    r_o = oracle.residuals_batch(b, v_o)
    n = len(res)
    sq = np.array([float((r[b["expr_off"][s]:b["expr_off"][s + 1]] ** 2).sum()) for s in range(n)])
    sq_o = np.array([float((r_o[b["expr_off"][s]:b["expr_off"][s + 1]] ** 2).sum()) for s in range(n)])
    solved, solved_o = sq < 1e-4, sq_o < 1e-4
    if tight:
        assert np.array_equal(solved[other], solved_o[other]), np.nonzero(other & (solved != solved_o))[0]
        assert np.all(d[other] <= 1e-9 + 1e-3 * np.abs(res_o["sse"][other])), d[other].max()
    for s in np.nonzero(sp | (other & solved & solved_o))[0][:4000]:
        e0, e1 = int(b["expr_off"][s]), int(b["expr_off"][s + 1])
        if e1 == e0:
            continue
        tol = 1e-7 * max(1.0, res_o["scale"][s]) + 1e-4 * np.sqrt(res_o["sse"][s])
        if not tight:  # same counts, but a cond^2 step may sit elsewhere in a flat valley: half the largest residual
            tol += 0.5 * np.sqrt(res_o["sse"][s]) + 0.5 * max(np.max(np.abs(r_o[e0:e1])), np.max(np.abs(r[e0:e1])))
        assert np.max(np.abs(np.abs(r[e0:e1]) - np.abs(r_o[e0:e1]))) <= tol, f"system {s}"
    both_nan = nan_gpu & nan_ref  # same start point kept on both sides
    for s in np.nonzero(both_nan & (res["ncomp"] == 1))[0]:
        v0, v1 = int(b["var_off"][s]), int(b["var_off"][s + 1])
        assert np.array_equal(v[v0:v1], v_o[v0:v1]), s
    verdict = ((solved == solved_o) | (nan_gpu & nan_ref)).mean()
    return float((same | both_nan).mean()), float(verdict)


def tile_with_noise(one, n: int, seed: int = 1, var_noise: float = 0.05, param_noise: float = 0.02):
    """n Systems of the structure of the one-System batch `one`, start values and parameters jittered per System."""
    import numpy as np
    rng = np.random.default_rng(seed)
    nv, ne = int(one["var_off"][1]), int(one["expr_off"][1])
    out = {
        "var_off": (np.arange(n + 1, dtype=np.uint64) * nv).astype(np.uint32),
        "expr_off": (np.arange(n + 1, dtype=np.uint64) * ne).astype(np.uint32),
        "vars": np.tile(np.asarray(one["vars"], dtype=np.float64), n) + var_noise * rng.standard_normal(n * nv),
        "var_fixed": np.tile(np.asarray(one["var_fixed"], dtype=np.uint8), n),
        "expr_tag": np.tile(np.asarray(one["expr_tag"], dtype=np.uint8), n),
        "expr_idx": np.tile(np.asarray(one["expr_idx"], dtype=np.uint32).reshape(-1), n),
        "expr_param": np.tile(np.asarray(one["expr_param"], dtype=np.float64), n) * (1.0 + param_noise * rng.standard_normal(n * ne)),
    }
    for k, dt in (("var_comp", np.uint16), ("expr_comp", np.uint16)):
        if one.get(k) is not None:
            out[k] = np.tile(np.asarray(one[k], dtype=dt), n)
    return out


def tiny_sketch_batches(n: int):
    """Batches of one structure of at most eight variables and eight expressions — what fx_grouped_tiny.hip takes: the reference's
    quadrilateral (consistent / impossible targets: rejected trials, singular steps), one with a fixed point, and a sketch of four
    points under five constraint kinds."""
    from fiksi_amd import workloads
    out = [("quadrilateral", tile_with_noise(workloads.quadrilateral(True), n, seed=11)),
           ("quadrilateral_impossible", tile_with_noise(workloads.quadrilateral(False), n, seed=12))]
    q = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in workloads.quadrilateral(True).items()}
    q["var_fixed"][:2] = 1
    q["var_fixed"][3] = 1
    out.append(("quadrilateral_fixed_point", tile_with_noise(q, n, seed=13)))
    s = System()
    P = [elements.Point.create(s, x, y) for x, y in ((0.1, 0.0), (2.0, 0.2), (1.9, 1.8), (-0.2, 2.1))]
    ln = elements.Line.create(s, P[0], P[1])
    constraints.PointPointDistance.create(s, P[0], P[1], 2.0)
    constraints.PointPointPointAngle.create(s, P[0], P[1], P[2], 1.5)
    constraints.PointLineDistance.create(s, P[3], ln, 2.0)
    constraints.PointPointDistance.create(s, P[2], P[3], 2.1)
    constraints.SegmentSegmentLengthEquality.create(s, P[0], P[1], P[1], P[2])
    out.append(("four_points_five_kinds", tile_with_noise(s.flatten(), n, seed=14, var_noise=0.03)))
    return out
