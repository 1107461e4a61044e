"""Diagnostic driver for rocprofv3: upload cfg3, run the fused solve a few times."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = fiksi_amd.Context(0)
db = ctx.upload(workloads.ring16(n))
for _ in range(reps):
    db.system_solve()
for _ in range(reps):
    db.eval_residual_jacobian(0)
ctx.synchronize()
