"""Diagnostic: phase cycles of the grouped kernel on a small-System batch (hinged triangles, <= 16 free variables)."""
import sys
sys.path.insert(0, '.')
import fiksi_amd
from fiksi_amd import workloads
n = 100000
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = fiksi_amd.Context(0)
b = workloads.hinged_triangles(n, k)
db = ctx.upload(b)
db.system_solve(); ctx.synchronize()
ctx.timer_begin()
for _ in range(5): db.system_solve()
ms = ctx.timer_end() / 5
res = db.get_results()
print(f"hinged({k}): vars/system {len(b['vars'])//n}; solve {ms:.3f} ms; trials/system {res['trials'].mean():.2f} accepted {res['accepted'].mean():.2f}")
ph = db.phase_cycles()
tot = sum(ph.values())
print({kk: f"{v/tot:.1%}" for kk, v in ph.items()}, f"cycles/system {tot/n:.0f}")
print({kk: int(v/n) for kk, v in ph.items()})
