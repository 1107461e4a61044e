"""Diagnostic: a large batch WITHOUT shared structure (600 distinct random sketches + 300 mixed-kind ones, repeated)
through the grouped kernel and through the one-System-per-wavefront kernel."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import fiksi_amd
from fiksi_amd import workloads
import helpers
ctx = fiksi_amd.Context(0)
flats = [helpers.random_sketch(s).flatten() for s in range(600)] + [helpers.mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(300)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
b = workloads.concat(flats * reps)
n = len(b['var_off']) - 1
for tag in ('1', '0'):
    ctx.set_routing(int(tag))
    db = ctx.upload(b)
    db.system_solve(); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(3): db.system_solve()
    ms = ctx.timer_end() / 3
    r = db.get_results()
    print(f"FIKSI_AMD_GROUPED={tag}: {n} Systems, route {db.solve_route()}, {ms:.3f} ms, trials/system {r['trials'].mean():.1f}, max trials {r['trials'].max()}")
    db.free()
