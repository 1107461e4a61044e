"""Diagnostic: fiksi_amd and torch (with its bundled HIP runtime and RCCL) in ONE process.
usage: python tools/torch_coexist.py [torch-first|fiksi-first]"""
import ctypes, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "torch-first"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
if order == "torch-first":
    import torch
    import fiksi_amd
else:
    import fiksi_amd
    import torch
import torch.distributed as dist
from fiksi_amd import workloads
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
ctx = fiksi_amd.Context(0)
v, res = ctx.system_solve_batch(workloads.ring16(1000))
t = torch.tensor([float(res["accepted"].sum())], dtype=torch.float64, device="cuda")
dist.all_reduce(t); dist.barrier()
print(order, "ok:", ctx.name(), int(t.item()), "accepted steps; torch sees", torch.cuda.get_device_name(0))
dist.destroy_process_group()
