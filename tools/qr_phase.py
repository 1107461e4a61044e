"""Diagnostic: where the FX_STEP_QR kernel spends its cycles (stamped build, 32-column instantiation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads, abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = fiksi_amd.Context(0)
db = ctx.upload(workloads.ring16(n))
o = abi.solving_opts(solver=2)
db.system_solve(o); ctx.synchronize()
ctx.timer_begin(); db.system_solve(o); ms = ctx.timer_end()
res = db.get_results()
ph = db.phase_cycles(o)
tot = sum(ph.values())
print(f"{ms:.2f} ms / {n}; trials {res['trials'].sum()}")
print({k: f"{v / tot:.1%}" for k, v in ph.items()}, f"cycles per trial {tot / res['trials'].sum():.0f}")
