"""Diagnostic: where the fused solve kernel spends its cycles (stamped build). Run on the GPU box."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import workloads, abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = fiksi_amd.Context(0)
b = workloads.ring16(n)
db = ctx.upload(b)
db.system_solve(); ctx.synchronize()
ctx.timer_begin()
for _ in range(5): db.system_solve()
ms = ctx.timer_end() / 5
res = db.get_results()
print(f"solve {ms:.3f} ms / {n} systems -> {n/ms*1e3/1e6:.2f} M systems/s; trials/system {res['trials'].mean():.2f} accepted {res['accepted'].mean():.2f}")
ph = db.phase_cycles()
tot = sum(ph.values())
print({k: f"{v/tot:.1%}" for k, v in ph.items()}, f"cycles/system {tot/n:.0f}; per trial {tot/res['trials'].sum():.0f}")
print({k: int(v/n) for k, v in ph.items()})
