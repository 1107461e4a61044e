"""Diagnostic: Systems just above / below the one-wavefront limit (hinged-triangle chains of n triangles:
2n + 2 points... 2(2n+1) variables)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for n in (15, 16, 20, 30):
    b = workloads.hinged_triangles(count, n)
    nv = int(b["var_off"][1]); ne = int(b["expr_off"][1])
    ctx.system_solve_batch(b)
    t = time.time(); v, res = ctx.system_solve_batch(b); dt = time.time() - t
    print(f"hinged_triangles({n}) x{count}: {nv} variables, {ne} expressions per System: {dt*1e3:.1f} ms "
          f"({dt/count*1e6:.0f} us per System), converged {float((res['sse_unscaled'] < 1e-4).mean()):.2f}")
