"""RecursiveAssembly on the device vs the oracle on a handful of canonical sketches (prints, no assertions)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fiksi_amd as F
from oracle import oracle

P = F.elements.Point.create
D = F.constraints.PointPointDistance.create


def sketches():
    s = F.System(); p = [P(s, 0, 0), P(s, 1, .5), P(s, 2, 1)]
    for a, b in ((0, 1), (0, 2), (1, 2)): D(s, p[a], p[b], 1.)
    yield "triangle", s
    s = F.System(); p = [P(s, 0, 0), P(s, 1, .2), P(s, .4, 1.1), P(s, 1.5, 1.2)]
    for a, b in ((0, 1), (0, 2), (1, 2), (1, 3), (2, 3)): D(s, p[a], p[b], 1.)
    yield "two_triangles", s
    s = F.System(); p = [P(s, 0, 0), P(s, 1.1, 0), P(s, 1, 1.2), P(s, 0.1, 1)]
    for a, b, d in ((0, 1, 1), (1, 2, 1), (2, 3, 1), (3, 0, 1), (0, 2, 2 ** .5)): D(s, p[a], p[b], d)
    yield "square_diag", s
    s = F.System(); c = P(s, 0.5, 0.)
    for t in range(3):
        a = P(s, 1.1 + t, 0.5 + 0.3 * t); b = P(s, 2.1 + t, 1. + 0.2 * t)
        D(s, c, a, 1.); D(s, c, b, 1.); D(s, a, b, 1.)
    yield "hinged3", s


for solver in (0, 2):
    for name, s in sketches():
        g = s.graph()
        v, plan, steps, fl = oracle.solve_recursive(g, trial_cap=4096)
        opts = F.SolvingOptions(decomposer=F.Decomposer.RecursiveAssembly)
        o = opts._to_abi(); o.lm.solver = solver
        import ctypes as C
        from fiksi_amd._lib import lib, check, FxResult
        res = FxResult()
        t = time.time()
        rc = lib.fxs_system_solve(s._h, F.default_context().handle, C.byref(o), C.byref(res))
        dt = time.time() - t
        got = s.flatten()["vars"]
        rms = float(np.sqrt(np.mean(np.square(s.constraint_residuals()))))
        print(name, "solver", solver, "rc", rc, "flags", fl, "steps", len(steps), "res", res.accepted, res.trials, res.ncomp,
              "oracle", int(steps["accepted"].sum()), int(steps["trials"].sum()), "maxdiff", float(np.max(np.abs(got - v))),
              "rms", rms, "sse_unscaled", res.sse_unscaled, f"{dt*1e3:.2f} ms", flush=True)
