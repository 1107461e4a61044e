"""fiksi_amd and PyTorch-ROCm (with its bundled HIP runtime and RCCL) in one process, in either import
order — the situation of every rank of `bench.py --gpus N`."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["torch-first", "fiksi-first"])
def test_one_hip_runtime_shared_with_torch(order, built):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541" if order == "torch-first" else "29542",
               RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "torch_coexist.py"), order], capture_output=True,
                         text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and f"{order} ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_runtime_choice_is_reported(fiksi):
    from fiksi_amd import _lib

    assert _lib.HIP_RUNTIME.split()[0] in ("torch", "system")
