"""Diagnostic driver: BASELINE cfg2 sketch solved with Decomposer::SinglePass (block by block)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import abi, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ctx = fiksi_amd.Context(0)
b = workloads.large_sketch(n)
blocks = abi.single_pass_blocks(b, 0)
print(f"{len(blocks)} blocks, largest {max(len(v) for _, _, v in blocks)} free variables")
for _ in range(2):
    t = time.time(); v, res = ctx.system_solve_batch(b, abi.solving_opts(decomposer=1)); dt = time.time() - t
    print(f"cfg2 SinglePass n={n}: {dt:.3f} s, {res[0]}")
