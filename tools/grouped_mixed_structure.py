"""Diagnostic: 100 000 ring16 sketches with and without a shared structure (every 50th sketch gauge-fixed switches
the shared-structure shortcut off for the whole batch) through the grouped kernel."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import fiksi_amd
from fiksi_amd import workloads
ctx = fiksi_amd.Context(0)
u = workloads.ring16(98000)
g = workloads.ring16(2000, fix_gauge=True)
parts = []
for k in range(50):
    parts += [workloads.shard(u, k, 50), workloads.shard(g, k, 50)]
mixed = workloads.concat(parts)
for name, b in (("shared structure", workloads.ring16(100000)), ("every 50th sketch differs", mixed)):
    for tag in ('1', '0'):
        ctx.set_routing(int(tag))
        db = ctx.upload(b)
        db.system_solve(); ctx.synchronize()
        ctx.timer_begin()
        for _ in range(5): db.system_solve()
        ms = ctx.timer_end() / 5
        print(f"{name:28s} FIKSI_AMD_GROUPED={tag}: {len(b['var_off']) - 1} Systems, {ms:.3f} ms")
        db.free()
