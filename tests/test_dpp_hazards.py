"""The hand-written DPP instructions of fx_grouped.hip / fx_grouped_c.hip (fx_grouped_rows.h), of fx_grouped_tiny.hip (half-row
broadcasts) and of the multifrontal build in fx_sparse.hip (fx_front.h) are inline asm, which the compiler's hazard recogniser
does not look into: a DPP read needs two wait states after a VALU write of its source register. The ISA the
Makefile's flags produce is scanned for that pattern (tools/check_dpp_hazards.py) — CPU only, hipcc
cross-compiles."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("name, counts", [("fx_grouped", {"v_fmac_f64_dpp": 1000, "v_fmac_f32_dpp": 800}),
                                          ("fx_grouped_c", {"v_fmac_f64_dpp": 1000, "v_fmac_f32_dpp": 800}),  # (shares fx_grouped_rows.h)
                                          ("fx_grouped_tiny", {"v_mov_b64_dpp": 100}), ("fx_sparse", {"v_fmac_f64_dpp": 200})])
def test_no_valu_write_to_dpp_read_hazard_in_the_grouped_kernels(tmp_path, name, counts):
    src = os.path.join(ROOT, "fiksi_amd", "csrc", name + ".hip")
    out = tmp_path / (name + ".s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S",
           "-I", os.path.join(ROOT, "include"), src, "-o", str(out)]
    subprocess.run(cmd, check=True, cwd=str(tmp_path), timeout=900)
    text = out.read_text()
    for op, at_least in counts.items():  # the scan sees the instructions
        assert text.count(op) > at_least, (op, text.count(op))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


def test_the_scan_flags_a_hazard(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("\tv_mul_f64 v[4:5], v[0:1], v[2:3]\n\tv_fmac_f64_dpp v[8:9], -v[4:5], v[2:3] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    ok = tmp_path / "ok.s"
    ok.write_text("\tv_mul_f64 v[4:5], v[0:1], v[2:3]\n\ts_nop 1\n\tv_fmac_f64_dpp v[8:9], -v[4:5], v[2:3] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    # the write reaches the DPP read along a taken branch only (one wait state: the branch itself)
    jump = tmp_path / "jump.s"
    jump.write_text("\tv_mul_f64 v[4:5], v[0:1], v[2:3]\n\ts_cbranch_execz .L1\n\tv_mov_b32_e32 v9, v8\n\ts_nop 1\n.L1:\n"
                    "\tv_fmac_f64_dpp v[8:9], -v[4:5], v[2:3] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    tool = os.path.join(ROOT, "tools", "check_dpp_hazards.py")
    assert subprocess.run([sys.executable, tool, str(bad)], capture_output=True).returncode == 1
    assert subprocess.run([sys.executable, tool, str(jump)], capture_output=True).returncode == 1
    assert subprocess.run([sys.executable, tool, str(ok)], capture_output=True).returncode == 0
