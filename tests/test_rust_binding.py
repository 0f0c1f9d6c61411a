"""rust/frw-sys/src/lib.rs against include/frw.h: the Rust binding cannot be compiled in this image (no cargo / rustc), so
its agreement with the C header is checked mechanically -- every function the header declares is declared in the
`extern "C"` block, nothing else is, and each has the same number of parameters."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_prototypes():
    text = open(os.path.join(ROOT, "include", "frw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(frw_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def _rust_prototypes():
    text = open(os.path.join(ROOT, "rust", "frw-sys", "src", "lib.rs")).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (frw_\w+)\s*\(([^;]*?)\)\s*(->\s*[^;]+)?;", block, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_frw_sys_declares_exactly_the_header():
    c, r = _c_prototypes(), _rust_prototypes()
    assert len(c) > 35
    assert sorted(c) == sorted(r)
    assert {k: v for k, v in c.items() if r[k] != v} == {}


def test_rust_structs_mirror_the_c_structs():
    """field order and array lengths of the three #[repr(C)] layout structs"""
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "frw.h")).read(), flags=re.S)
    rs = open(os.path.join(ROOT, "rust", "frw-sys", "src", "lib.rs")).read()
    for name in ("frw_layout", "frw_layout_dual", "frw_compact_layout"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s_t;" % (name, name), h, flags=re.S).group(1)
        c_fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, rest = decl.split(None, 1)
            for f in rest.split(","):
                m = re.match(r"\s*(\w+)(?:\[(\w+)\])?", f)
                n = m.group(2)
                n = {"FRW_NUM_SEGMENTS": "8", "FRW_NUM_SEGMENTS_DUAL": "15"}.get(n, n)
                c_fields.append((m.group(1), ctype, n))
        rbody = re.search(r"pub struct %s_t \{(.*?)\}" % name, rs, flags=re.S).group(1)
        r_fields = []
        for m in re.finditer(r"pub (\w+): (\[(\w+); (\d+)\]|\w+)", rbody):
            r_fields.append((m.group(1), {"i32": "int32_t", "u64": "uint64_t"}[m.group(3) or m.group(2)], m.group(4)))
        assert c_fields == r_fields, name
