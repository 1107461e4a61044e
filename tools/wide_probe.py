"""Diagnostic: latency and throughput of the wide kernel (65..128 free variables)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
for n_tri in (16, 20, 31):
    for count in (1, 256, 2048):
        b = workloads.hinged_triangles(count, n_tri)
        db = ctx.upload(b)
        db.system_solve(); ctx.synchronize()
        ctx.timer_begin()
        for _ in range(3): db.system_solve()
        ms = ctx.timer_end() / 3
        res = db.get_results()
        print(f"hinged({n_tri}) {int(b['var_off'][1])} vars x{count}: {ms:.3f} ms per solve, trials/system {res['trials'].mean():.1f}")
        db.free()
