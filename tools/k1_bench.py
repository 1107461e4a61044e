"""Diagnostic: time the Jacobian-assembly kernel (K1) alone on the cfg3 batch."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = fiksi_amd.Context(0)
b = workloads.ring16(n)
db = ctx.upload(b)
for _ in range(5): db.eval_residual_jacobian(0)
ctx.synchronize()
for reps in (20, 200):
    ctx.timer_begin()
    for _ in range(reps): db.eval_residual_jacobian(0)
    ms = ctx.timer_end() / reps
    by = workloads.k1_algorithmic_bytes(b, db.nnz)
    print(f"K1 x{reps}: {ms*1e3:.1f} us/launch, {by/ms/1e6:.0f} GB/s algorithmic = {by/ms/1e6/8000:.1%} of 8 TB/s")
ctx.timer_begin()
for _ in range(200): db.eval_residual(0)
ms = ctx.timer_end() / 200
print(f"K2 (residual only): {ms*1e3:.1f} us/launch")
