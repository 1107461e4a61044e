"""Link rate of this box, both directions, pageable / page-locked / registered-in-place host memory (12.8 MB and 25.6 MB)."""
import ctypes as C
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
for mb in (12.8, 25.6):
    n = int(mb * 1e6 / 8)
    dev = torch.empty(n, dtype=torch.float64, device="cuda")
    page = torch.from_numpy(np.random.rand(n))
    pin = torch.empty(n, dtype=torch.float64).pin_memory()
    reg_np = np.random.rand(n)
    assert hip.hipHostRegister(C.c_void_p(reg_np.ctypes.data), C.c_size_t(reg_np.nbytes), 0) == 0
    reg = torch.from_numpy(reg_np)
    for name, host in (("pageable", page), ("page-locked (hipHostMalloc)", pin), ("registered in place", reg)):
        for direction in ("h2d", "d2h"):
            best = 1e9
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if direction == "h2d":
                    dev.copy_(host, non_blocking=True)
                else:
                    host.copy_(dev, non_blocking=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print(f"{mb} MB {name:30s} {direction}: {best * 1e3:.3f} ms = {mb / 1e3 / best:.1f} GB/s", flush=True)
    hip.hipHostUnregister(C.c_void_p(reg_np.ctypes.data))
