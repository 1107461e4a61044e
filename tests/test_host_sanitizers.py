"""The library's host-side logic (batch analysis, Jacobian structure, SinglePass decomposition, the RecursiveAssembly
plan, QR planning with its COLAMD, the System builder — ~3 000 lines of index arithmetic) under AddressSanitizer + UndefinedBehaviorSanitizer:
`make -C fiksi_amd/csrc asan` compiles the host sources (fx_analyze / fx_programs / fx_upload / fx_solve / fx_entry / fx_builder .cpp) with g++ against stubs of the HIP runtime
(fx_hip_shim.h: every device entry point answers FX_ERR_NO_DEVICE), and the CPU tests of those parts run against that
library in a child process with the sanitizer runtimes preloaded. GPU AddressSanitizer is not available on the pool;
the kernels' own indexing is covered by the parity tests."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_logic_under_asan_and_ubsan():
    csrc = os.path.join(ROOT, "fiksi_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-s", "asan"])
    lib = os.path.join(ROOT, "fiksi_amd", "libfiksi_host_asan.so")
    pre = []
    for name in ("libasan.so", "libubsan.so"):
        path = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
        if not os.path.isabs(path):
            pytest.skip(f"{name} not installed")
        pre.append(path)
    env = dict(os.environ, LD_PRELOAD=" ".join(pre), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               FIKSI_AMD_LIBRARY=lib, FIKSI_AMD_HIP_RUNTIME="system")
    tests = ["tests/test_host.py", "tests/test_single_pass.py", "tests/test_qr_plan.py", "tests/test_recursive_assembly.py",
             "tests/test_reference_suite.py"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = p.stdout + p.stderr
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_mutation_fuzz_of_the_entry_points_under_asan_and_ubsan():
    """tools/abi_fuzz.py, 300 iterations from a fixed seed, against the sanitizer build with its make-believe device
    (fx_hip_shim.h: allocations and copies are real, kernel launches answer "no device"): corrupted batches, column patterns,
    plan capacities and index arguments through fx_batch_validate, fx_jacobian_structure, fx_single_pass_blocks,
    fx_qr_symbolic, fxs_recursive_plan and — as far as their host analysis and uploads go — fx_batch_upload,
    fx_system_solve_batch, fx_system_solve_batch_multi, fx_system_prepare_batch, fx_cluster_solve_batch,
    fx_pose_transform_points, fx_unscale_vars_strided. Error codes only: no sanitizer report, no crash."""
    csrc = os.path.join(ROOT, "fiksi_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-s", "asan"])
    lib = os.path.join(ROOT, "fiksi_amd", "libfiksi_host_asan.so")
    pre = []
    for name in ("libasan.so", "libubsan.so"):
        path = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
        if not os.path.isabs(path):
            pytest.skip(f"{name} not installed")
        pre.append(path)
    env = dict(os.environ, LD_PRELOAD=" ".join(pre), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               FIKSI_AMD_LIBRARY=lib, FIKSI_AMD_HIP_RUNTIME="system", FIKSI_AMD_SHIM_FAKE_DEVICE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_fuzz.py"), "300", "20261004"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = p.stdout + p.stderr
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    assert "contexts: 3" in out and "'fx_qr_symbolic'" in out and "'fx_system_solve_batch_multi'" in out


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_no_allocation_failure_crosses_the_c_abi():
    """tools/alloc_fail_sweep.py against the sanitizer build: the n-th allocation of a call fails, for every n (sampled where a
    call makes thousands) — validation, Jacobian structure, SinglePass blocks, QR planning with its COLAMD, the System
    builder and its RecursiveAssembly plan, and through the make-believe device every upload path, the plans built on first
    use, the host-buffer call and the multi-device call with its threads. Each call answers FX_ERR_NOMEM (or goes on with
    fewer threads), nothing aborts, and no host or device block stays behind (the reference's System::solve
    returns: fiksi/src/lib.rs:464)."""
    csrc = os.path.join(ROOT, "fiksi_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-s", "asan"])
    lib = os.path.join(ROOT, "fiksi_amd", "libfiksi_host_asan.so")
    pre = []
    for name in ("libasan.so", "libubsan.so"):
        path = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
        if not os.path.isabs(path):
            pytest.skip(f"{name} not installed")
        pre.append(path)
    # (alloc_dealloc_mismatch: the build's counting operator new / delete sit on malloc / free, libstdc++'s own on ASan's)
    env = dict(os.environ, LD_PRELOAD=" ".join(pre), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:alloc_dealloc_mismatch=0",
               UBSAN_OPTIONS="print_stacktrace=1", FIKSI_AMD_LIBRARY=lib, FIKSI_AMD_HIP_RUNTIME="system", FIKSI_AMD_SHIM_FAKE_DEVICE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "alloc_fail_sweep.py"), "120"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1500)
    out = p.stdout + p.stderr
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "terminate called" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    assert "allocation-failure sweep: ok" in out and "make-believe device: True" in out
