"""Diagnostic: a batch of several large sketches (sparse path, host threads + streams)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import fiksi_amd
from fiksi_amd import workloads
ctx = fiksi_amd.Context(0)
for count, n in ((1, 1000), (16, 1000), (64, 300)):
    b = workloads.concat([workloads.large_sketch(n, seed=7 + k) for k in range(count)])
    ctx.system_solve_batch(b)
    t = time.time(); v, res = ctx.system_solve_batch(b); dt = time.time() - t
    print(f"{count} sketches of {n} points: {dt*1e3:.1f} ms ({dt/count*1e3:.1f} ms each), trials {res['trials'].sum()}")
