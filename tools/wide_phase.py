import sys; sys.path.insert(0, '.')
import fiksi_amd
from fiksi_amd import workloads
ctx = fiksi_amd.Context(0)
for n_tri in (16, 31):
    b = workloads.hinged_triangles(64, n_tri)
    db = ctx.upload(b)
    db.system_solve(); ctx.synchronize()
    ph = db.phase_cycles(); res = db.get_results()
    tot = sum(ph.values())
    print(f"hinged({n_tri}) {int(b['var_off'][1])} vars: cycles/system {tot//64}, per trial {tot/res['trials'].sum():.0f}", {k: f"{v/tot:.1%}" for k, v in ph.items()})
