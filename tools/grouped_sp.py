"""Diagnostic: Decomposer::SinglePass through the grouped kernel and through the one-System-per-wavefront kernel."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import fiksi_amd
from fiksi_amd import workloads, abi
ctx = fiksi_amd.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for name, b in (('hinged11', workloads.hinged_triangles(n, 11)), ('ring16', workloads.ring16(n)), ('hinged5', workloads.hinged_triangles(n, 5))):
    out = {}
    for tag in ('1', '0'):
        ctx.set_routing(int(tag))
        db = ctx.upload(b)
        o = abi.solving_opts(decomposer=1)
        db.system_solve(o); ctx.synchronize()
        route = db.solve_route(o)
        ctx.timer_begin()
        for _ in range(3): db.system_solve(o)
        ms = ctx.timer_end() / 3
        out[tag] = (db.get_vars().copy(), db.get_results().copy(), ms, route)
        db.free()
    (v1, r1, m1, q1), (v0, r0, m0, q0) = out['1'], out['0']
    same = np.array_equal(v1.view(np.uint64), v0.view(np.uint64))
    diff = [f for f in r1.dtype.names if not np.array_equal(r1[f], r0[f], equal_nan=True)]
    print(f"{name:9s} SinglePass: grouped (route {q1}) {m1:.3f} ms, single (route {q0}) {m0:.3f} ms, x{m0/m1:.2f}; vars bit-identical {same}, max |d| {np.nanmax(np.abs(v1-v0)):.2e}; result fields that differ: {diff}")
