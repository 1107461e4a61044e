"""Registers, scratch and code bytes of every kernel in a HIP object file (or the library): unbundles the gfx950 code
object and reads its metadata note and symbol table.
    python3 tools/kernel_resources.py fiksi_amd/csrc/build/fx_grouped.o [filter]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as d:
        co = d + "/dev.co"
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
        if r.returncode:
            # a linked library: the fat binary sits in .hip_fatbin
            fb = d + "/fat.bin"
            subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
            subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
        syms = subprocess.check_output([LLVM + "llvm-readelf", "-sW", co], text=True)
    size = {}
    for ln in syms.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    rows = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = g("name")
        if flt and flt not in name:
            continue
        dem = subprocess.check_output(["c++filt", name], text=True).strip()
        dem = re.sub(r"\(.*", "", dem)
        rows.append((dem, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), size.get(name, 0)))
    for r in sorted(rows):
        print("%-70s vgpr %4s sgpr %4s scratch %5s code %7d B" % r)


if __name__ == "__main__":
    main()
