"""bindings/ffi.rs (the Rust side of the boundary) against include/fiksi_amd.h without a Rust toolchain:
struct layouts by _Static_assert in a generated C file, constants, function arity, exported symbols."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ffi_rs_matches_the_header_and_the_library(fiksi):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ffi_layout.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all exported by libfiksi_amd.so" in r.stdout, r.stdout


def test_a_wrong_layout_is_caught(tmp_path):
    """The check has teeth: the same generator on a binding with two fields swapped fails to compile."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_ffi_layout as chk

    src = open(chk.RS).read().replace("pub accepted: u32,\n    pub trials: u32,\n    pub exit: u32,\n    pub ncomp: u32,\n    pub scale: f64,",
                                       "pub accepted: u32,\n    pub scale: f64,\n    pub trials: u32,\n    pub exit: u32,\n    pub ncomp: u32,")
    structs = chk.parse_structs(src)
    size, align, fields = chk.layout(structs, "fx_result", {})
    c = tmp_path / "bad.c"
    c.write_text('#include <stddef.h>\n#include "%s"\n' % chk.HDR +
                 "\n".join(f'_Static_assert(offsetof(fx_result, {f}) == {off}, "{f}");' for f, off, _ in fields) + "\n")
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", str(c)], capture_output=True, text=True)
    assert r.returncode != 0 and "static assertion failed" in (r.stdout + r.stderr)
