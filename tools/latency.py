"""Diagnostic: latency of one System::solve through the builder API and through the batch ABI."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import fiksi_amd as F
from fiksi_amd import abi, workloads
ctx = F.default_context()
for n in (1, 100, 10000):
    b = abi.normalize_batch(workloads.ring16(n))
    ctx.system_solve_batch(b)
    t = time.time(); reps = 20
    for _ in range(reps): ctx.system_solve_batch(b)
    dt = (time.time() - t) / reps
    print(f"fx_system_solve_batch, {n} ring16 systems: {dt*1e3:.3f} ms per call")
s = F.System()
pts = [F.elements.Point.create(s, float(i), float(i * i % 3)) for i in range(4)]
for i in range(4):
    F.constraints.PointPointDistance.create(s, pts[i], pts[(i + 1) % 4], 1.0)
s.solve()
t = time.time()
for _ in range(50): s.solve()
print(f"System.solve (4 points, 4 distances): {(time.time()-t)/50*1e3:.3f} ms per call")
