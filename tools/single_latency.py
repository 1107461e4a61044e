"""One System::solve through the builder API (the reference bench's sizes 1 / 4 / 16 / 64, fiksi_bench.rs:49), per-call latency."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd

ctx = fiksi_amd.Context(0)
for n_tri in (1, 4, 11, 16, 64):
    s = fiksi_amd.System()
    hinge = fiksi_amd.elements.Point.create(s, 0., 0.)
    pts = [(hinge, 0., 0.)]
    for t in range(n_tri):
        p1 = fiksi_amd.elements.Point.create(s, -1., float(t)); p2 = fiksi_amd.elements.Point.create(s, 1., float(t))
        pts += [(p1, -1., float(t)), (p2, 1., float(t))]
        fiksi_amd.constraints.PointPointDistance.create(s, hinge, p1, 2.)
        fiksi_amd.constraints.PointPointDistance.create(s, hinge, p2, 2.)
        fiksi_amd.constraints.PointPointDistance.create(s, p1, p2, 3.)
    s.solve(ctx=ctx)
    reps, best, tot = 200, 1e9, 0.0
    for _ in range(reps):
        for h, x, y in pts:
            h.update_value(s, x, y)
        t0 = time.perf_counter(); s.solve(ctx=ctx); dt = time.perf_counter() - t0
        best = min(best, dt); tot += dt
    print(f"hinged_triangles_{n_tri}: mean {tot / reps * 1e3:.4f} ms, best {best * 1e3:.4f} ms", flush=True)
