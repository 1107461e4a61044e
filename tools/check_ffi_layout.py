"""cargo-free check of bindings/ffi.rs against include/fiksi_amd.h (the image has no Rust toolchain).

  * every `#[repr(C)] pub struct` with fields is laid out by the C rules (field order, natural alignment) and a
    generated C file `_Static_assert`s each offset, the size and the alignment against the header's struct — compiled
    with gcc -fsyntax-only;
  * every `pub const` that mirrors a header enumerator / macro has the header's value (same generated file);
  * every `pub fn` of the extern block is declared in the header with the same number of arguments, and (when the
    built library is there) exported by it.
Exit code 0 = all good; prints what it checked.  `--emit-c PATH` keeps the generated C file."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, "bindings", "ffi.rs")
HDR = os.path.join(ROOT, "include", "fiksi_amd.h")

PRIM = {"u8": (1, 1), "i8": (1, 1), "u16": (2, 2), "i16": (2, 2), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4), "c_int": (4, 4),
        "u64": (8, 8), "i64": (8, 8), "f64": (8, 8), "usize": (8, 8)}


def parse_structs(src):
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\][^\n]*\n(?:\s*///[^\n]*\n)*\s*pub struct (\w+)\s*\{(.*?)\n\}", src, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = re.findall(r"pub (\w+):\s*([^,\n]+),", body)
        if fields:
            structs[name] = [(f, t.strip()) for f, t in fields]
    return structs


def layout(structs, name, memo):
    if name in memo:
        return memo[name]
    off, align, fields = 0, 1, []
    for f, t in structs[name]:
        if t.startswith("*"):
            sz, al = 8, 8
        elif t in PRIM:
            sz, al = PRIM[t]
        elif t in structs:
            sz, al, _ = layout(structs, t, memo)
        else:
            raise SystemExit(f"ffi.rs: unknown field type {t} in {name}")
        off = (off + al - 1) // al * al
        fields.append((f, off, sz))
        off += sz
        align = max(align, al)
    size = (off + align - 1) // align * align
    memo[name] = (size, align, fields)
    return memo[name]


TMAP = {
    'int': 'c_int', 'uint32_t': 'u32', 'int32_t': 'i32', 'uint64_t': 'u64', 'double': 'f64', 'float': 'f32', 'size_t': 'usize', 'char': 'c_char',
    'uint8_t': 'u8', 'uint16_t': 'u16', 'void': 'c_void',
}


def rust_type(t):
    t = re.sub(r'\s+', ' ', t.strip())
    m = re.match(r'^(const )?(\w+)( const)?\s*((?:\*\s*(?:const\s*)?)*)$', t)
    if not m:
        raise SystemExit('cannot parse type: ' + t)
    const, base, _, stars = m.groups()
    rust = TMAP.get(base, base)
    marks = re.findall(r'\*\s*(const)?', stars)
    for k in range(len(marks)):
        is_const = (const is not None) if k == 0 else (marks[k - 1] == 'const')
        rust = ('*const ' if is_const else '*mut ') + rust
    return rust


def regen():
    """Rewrites the `extern "C"` block of bindings/ffi.rs from the header's prototypes."""
    hdr = re.sub(r'/\*.*?\*/', '', open(HDR).read(), flags=re.S)
    protos = re.findall(r'^(int|void|const char\*|uint64_t) (fx_\w+)\(([^;]*?)\);', hdr, flags=re.M | re.S)
    out = []
    for ret, name, args in protos:
        args = ' '.join(args.split())
        params = []
        if args != 'void':
            for a in args.split(','):
                m = re.match(r'^(.*?)(\w+)(\[\d+\])?$', a.strip())
                ty, nm, arr = m.groups()
                ty = ty.strip() + (' *' if arr else '')
                if nm in ('type', 'ref', 'in', 'loop', 'fn', 'mod', 'use', 'box', 'self'):
                    nm += '_'
                params.append(f'{nm}: {rust_type(ty)}')
        r = {'int': ' -> c_int', 'void': '', 'const char*': ' -> *const c_char', 'uint64_t': ' -> u64'}[ret]
        out.append(f'    pub fn {name}({", ".join(params)}){r};')
    src = open(RS).read()
    a = src.index('extern "C" {\n') + len('extern "C" {\n')
    b = src.rindex('}')
    open(RS, 'w').write(src[:a] + '\n'.join(out) + '\n' + src[b:])
    print(f'bindings/ffi.rs: extern block regenerated ({len(out)} functions)')


def main():
    if '--regen' in sys.argv:
        regen()
    src = open(RS).read()
    hdr = open(HDR).read()
    structs = parse_structs(src)
    memo = {}
    lines = ['#include <stddef.h>', f'#include "{HDR}"']
    for name in structs:
        size, align, fields = layout(structs, name, memo)
        lines.append(f'_Static_assert(sizeof({name}) == {size}, "{name}: size");')
        lines.append(f'_Static_assert(_Alignof({name}) == {align}, "{name}: alignment");')
        for f, off, sz in fields:
            lines.append(f'_Static_assert(offsetof({name}, {f}) == {off}, "{name}.{f}: offset");')
            lines.append(f'_Static_assert(sizeof((({name}*)0)->{f}) == {sz}, "{name}.{f}: size");')
    consts = re.findall(r"pub const (FX_\w+): (\w+) = ([^;]+);", src)
    for cname, _, val in consts:
        lines.append(f'_Static_assert((long long)({cname}) == (long long)({val.strip()}), "{cname}");')
    # the header must not hold a struct the binding forgot (opaque handles aside)
    for hs in re.findall(r"typedef struct (\w+) \{", hdr):
        if hs not in structs:
            raise SystemExit(f"ffi.rs has no #[repr(C)] struct for {hs}")
    cfile = None
    if "--emit-c" in sys.argv:
        cfile = sys.argv[sys.argv.index("--emit-c") + 1]
    tmp = cfile or os.path.join(tempfile.mkdtemp(), "ffi_layout_check.c")
    with open(tmp, "w") as f:
        f.write("\n".join(lines) + "\n")
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", tmp], capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise SystemExit("layout of bindings/ffi.rs differs from include/fiksi_amd.h")
    # functions: declared in the header with the same arity, exported by the library
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"^[\w\* ]+?\b(fx_\w+)\(([^;]*?)\);", hdr_nc, flags=re.M | re.S)}
    fns = re.findall(r"pub fn (fx_\w+)\((.*?)\)(?: -> [^;]+)?;", src, flags=re.S)
    arity = lambda a: 0 if a.strip() in ("", "void") else a.count(",") + 1
    for name, args in fns:
        if name not in protos:
            raise SystemExit(f"ffi.rs declares {name}, the header does not")
        if arity(args) != arity(protos[name]):
            raise SystemExit(f"{name}: {arity(args)} arguments in ffi.rs, {arity(protos[name])} in the header")
    missing = sorted(set(protos) - {n for n, _ in fns})
    if missing:
        raise SystemExit("the header declares functions ffi.rs lacks: " + ", ".join(missing))
    lib = os.path.join(ROOT, "fiksi_amd", "libfiksi_amd.so")
    exported = None
    if os.path.exists(lib):
        out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
        lost = [n for n, _ in fns if n not in exported]
        if lost:
            raise SystemExit("not exported by libfiksi_amd.so: " + ", ".join(lost))
    print(f"bindings/ffi.rs: {len(structs)} structs ({sum(len(v) for v in structs.values())} fields), {len(consts)} constants, "
          f"{len(fns)} functions == include/fiksi_amd.h" + ("" if exported is None else "; all exported by libfiksi_amd.so"))


if __name__ == "__main__":
    main()
