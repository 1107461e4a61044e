"""Diagnostic driver for rocprofv3: the Jacobian-assembly kernel (K1) alone on a batch of n ring16 sketches
(third argument `hinged`: the reference's bench sketch of 11 hinged triangles instead — distance rows only; `mixed`: ring16
sketches of two interleaved structures, workloads.ring16_two_structures)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = fiksi_amd.Context(0)
kind = sys.argv[3] if len(sys.argv) > 3 else "ring16"
b = {"hinged": lambda: workloads.hinged_triangles(n, 11), "mixed": lambda: workloads.ring16_two_structures(n, seed0=5_000_000),
     "ring16": lambda: workloads.ring16(n, seed0=5_000_000)}[kind]()
db = ctx.upload(b)
for _ in range(reps):
    db.eval_residual_jacobian(0)
ctx.synchronize()
