"""bindings/rust/src/rtx_ffi.rs against include/rtx.h (CPU test; there is no Rust toolchain here, so the shim is checked as text).

For every `#[repr(C)] pub struct` of the Rust file the repr(C) layout (size, alignment, field offsets; x86-64: pointers 8 bytes)
is computed from the declared field types and compared with sizeof / offsetof of the struct of the same name in rtx.h as gcc sees it;
every `pub const` is compared with the header's enum value; the extern block must declare exactly the header's entry points, with the
same number of arguments."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, "bindings", "rust", "src", "rtx_ffi.rs")
HDR = os.path.join(ROOT, "include", "rtx.h")

PRIM = {"i32": (4, 4), "u32": (4, 4), "i64": (8, 8), "u64": (8, 8), "f64": (8, 8), "u8": (1, 1)}
C_NAME = {"RtxMaterial": "rtx_material", "RtxTexture": "rtx_texture", "RtxImage": "rtx_image", "RtxLight": "rtx_light",
          "RtxSceneDesc": "rtx_scene_desc", "RtxCamera": "rtx_camera", "RtxOptions": "rtx_options", "RtxStats": "rtx_stats",
          "RtxSceneInfo": "rtx_scene_info"}


def _size_align(ty):
    ty = ty.strip()
    if ty.startswith("*"):
        return 8, 8
    m = re.match(r"\[(.+);\s*(\d+)\]$", ty)
    if m:
        s, a = _size_align(m.group(1))
        return s * int(m.group(2)), a
    return PRIM[ty]


def _rust_structs(text):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, re.S):
        fields, off, amax = [], 0, 1
        for f in re.finditer(r"pub (\w+):\s*([^,\n]+),", m.group(2)):
            s, a = _size_align(f.group(2))
            off = (off + a - 1) // a * a
            fields.append((f.group(1), off, s))
            off += s
            amax = max(amax, a)
        out[m.group(1)] = (fields, (off + amax - 1) // amax * amax)
    return out


def test_rust_structs_match_the_header(tmp_path):
    text = open(RS).read()
    structs = _rust_structs(text)
    assert set(structs) == set(C_NAME), "every struct of rtx.h that crosses the ABI has a #[repr(C)] twin"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rtx.h"', 'int main(void) {']
    for rs, (fields, _) in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (rs, C_NAME[rs]))
        for name, _, _ in fields:
            lines.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));' % (rs, name, C_NAME[rs], name, C_NAME[rs], name))
    lines += ["return 0; }"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = {}
    for ln in subprocess.check_output([str(exe)]).decode().split("\n"):
        p = ln.split()
        if p:
            got[p[0]] = tuple(int(x) for x in p[1:])
    n_fields = 0
    for rs, (fields, size) in structs.items():
        assert got[rs] == (size,), "%s: size %d in Rust, %d in C" % (rs, size, got[rs][0])
        for name, off, sz in fields:
            assert got["%s.%s" % (rs, name)] == (off, sz), "%s.%s: Rust (%d, %d) vs C %s" % (rs, name, off, sz, got["%s.%s" % (rs, name)])
            n_fields += 1
    # ... and no field of the C structs is missing on the Rust side: the sizes match and the fields tile the struct up to padding
    assert n_fields >= 90


def test_rust_constants_match_the_header(tmp_path):
    text = open(RS).read()
    consts = re.findall(r"pub const (RTX_\w+): \w+ = (\d+);", text)
    assert len(consts) >= 20
    body = "\n".join('printf("%s %%d\\n", (int)%s);' % (n, n) for n, _ in consts)
    src = tmp_path / "c.c"
    src.write_text('#include <stdio.h>\n#include "rtx.h"\nint main(void){%s return 0;}' % body)
    exe = tmp_path / "c"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)]).decode().strip().split("\n"))
    for n, v in consts:
        assert int(got[n]) == int(v), n


def test_rust_extern_block_declares_the_header_symbols():
    text = open(RS).read()
    block = text[text.index('extern "C" {'):]
    rust = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rtx_\w+)\((.*?)\)\s*->", block, re.S)}
    hdr = open(HDR).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    c = {m.group(1): m.group(2) for m in re.finditer(r"\b(rtx_\w+)\s*\(([^;{]*?)\)\s*;", hdr, re.S)}
    assert set(rust) == set(c), (sorted(set(rust) ^ set(c)))

    def nargs(a):
        a = a.strip()
        return 0 if a in ("", "void") else a.count(",") + 1
    for name in c:
        assert nargs(rust[name]) == nargs(c[name]), name
