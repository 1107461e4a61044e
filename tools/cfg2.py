"""Diagnostic driver: solve BASELINE cfg2 (one 5000-point sketch) through the sparse path."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ctx = fiksi_amd.Context(0)
b = workloads.large_sketch(n)
t = time.time(); v, res = ctx.system_solve_batch(b); dt = time.time() - t
print(f"cfg2 n={n}: {dt:.3f} s, {res[0]}")
