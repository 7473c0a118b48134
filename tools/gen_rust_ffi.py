#!/usr/bin/env python3
"""Generates rust/zkp-accel/src/ffi.rs from include/zkp_accel.h: every exported function, every enum constant and every
struct of the C ABI, transcribed 1:1 (tests/test_rust_shim.py checks that the committed file is exactly what this script
prints and that the extern set equals the header's).

    python tools/gen_rust_ffi.py > rust/zkp-accel/src/ffi.rs
"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HDR = (ROOT / "include" / "zkp_accel.h").read_text()

SCALAR = {"int32_t": "i32", "int64_t": "i64", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "int": "c_int", "float": "f32",
          "double": "f64", "uint8_t": "u8", "void": "c_void", "char": "c_char", "zkp_curve_t": "c_int"}
OPAQUE = ["zkp_ctx", "zkp_groth16_pk", "zkp_groth16_pk_multi", "zkp_fs_rng", "zkp_marlin_index"]
STRUCTS = ["zkp_ctx_config", "zkp_csr", "zkp_groth16_pk_desc", "zkp_marlin_index_desc", "zkp_marlin_rand", "zkp_marlin_proof",
           "zkp_groth16_timing", "zkp_marlin_timing"]


def strip_comments(t):
    return re.sub(r"/\*.*?\*/", "", t, flags=re.S)


H = strip_comments(HDR)
DEFINES = {m.group(1): (int(m.group(2)), "u32" if m.group(3) else "usize") for m in re.finditer(r"#define\s+(ZKP_[A-Z0-9_]+)\s+(\d+)(u?)\b", H)}


def rust_type(ctype: str) -> str:
    """'const uint64_t* const*' -> '*const *const u64' etc."""
    t = ctype.strip()
    # split off pointer levels from the right
    levels = []
    while True:
        t = t.strip()
        if t.endswith("const"):
            t2 = t[:-5].strip()
            if t2.endswith("*"):
                levels.append("const_ptr_marker")       # 'T* const' : constness of the pointer variable itself, irrelevant
                t = t2
                continue
        if t.endswith("*"):
            levels.append("*")
            t = t[:-1]
            continue
        break
    levels = [l for l in levels if l == "*"]
    base_const = False
    toks = t.split()
    if "const" in toks:
        base_const = True
        toks.remove("const")
    if toks and toks[0] == "struct":
        toks = toks[1:]
    base = " ".join(toks)
    rb = SCALAR.get(base, base)
    out = rb
    # innermost pointer takes the base constness; outer pointers: const if the C declaration said '* const*' — we do not
    # track that per level, so outer levels follow the convention of the header: 'const T* const*' -> *const *const T
    n = len(levels)
    for i in range(n):
        inner = i == 0
        if inner:
            out = ("*const " if base_const else "*mut ") + out
        else:
            out = ("*const " if base_const else "*mut ") + out
    return out


def parse_params(s: str):
    s = s.strip()
    if s in ("", "void"):
        return []
    out = []
    for p in s.split(","):
        p = " ".join(p.split())
        m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*\w*\s*\])?$", p)
        ctype, name, arr = m.group(1), m.group(2), m.group(3)
        if arr:
            ctype = ctype.strip() + "*"
        if name in ("type", "in", "ref", "mod", "fn", "box", "match", "move", "self", "use"):
            name += "_"
        out.append((name, rust_type(ctype)))
    return out


def functions():
    fns = []
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?)\s*(zkp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", H, flags=re.S | re.M):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        fns.append((name, rust_type(ret), parse_params(params)))
    return fns


def enums():
    out = []
    for m in re.finditer(r"typedef enum\s*\{(.*?)\}\s*(zkp_[a-z_0-9]+)\s*;", H, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                nxt = int(v)
            else:
                k = item
            out.append((k, nxt, m.group(2)))
            nxt += 1
    return out


def eval_dim(expr: str) -> int:
    expr = expr.strip()
    for k, v in DEFINES.items():
        expr = expr.replace(k, str(v[0]))
    assert re.fullmatch(r"[0-9*+ ()]+", expr), expr
    return int(eval(expr))


def structs():
    out = []
    for m in re.finditer(r"typedef struct\s*\{(.*?)\}\s*(zkp_[a-z_0-9]+)\s*;", H, flags=re.S):
        name, fields = m.group(2), []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # 'const uint64_t* a; ' or 'zkp_csr at, bt, ct' or 'uint64_t comm[9 * 12]' or 'const uint64_t* x; const uint8_t* y' (split by ;)
            first, *rest = [x.strip() for x in decl.split(",")]
            m2 = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[(.*?)\])?$", first)
            ctype, fname, _, dim = m2.group(1).strip(), m2.group(2), m2.group(3), m2.group(4)
            names = [(fname, dim)]
            for r in rest:
                m3 = re.match(r"^(\*?)\s*([A-Za-z_][A-Za-z0-9_]*)\s*(\[(.*?)\])?$", r)
                names.append((m3.group(2), m3.group(4)))
            for fn_, dim_ in names:
                rt = rust_type(ctype)
                if dim_ is not None:
                    rt = f"[{rt}; {eval_dim(dim_)}]"
                fields.append((fn_, rt))
        out.append((name, fields))
    return out


def main():
    w = sys.stdout.write
    w("//! Raw `extern \"C\"` declarations: a 1:1 transcription of `include/zkp_accel.h` — EVERY exported function, enum constant\n"
      "//! and struct.  GENERATED by tools/gen_rust_ffi.py from the header; do not edit by hand (tests/test_rust_shim.py\n"
      "//! regenerates it and compares).\n"
      "#![allow(non_camel_case_types, clippy::too_many_arguments)]\n"
      "use std::os::raw::{c_char, c_int, c_void};\n\n")
    for o in OPAQUE:
        w(f"#[repr(C)]\npub struct {o} {{\n    _p: [u8; 0],\n}}\n")
    w("\n")
    for k, v, ty in enums():
        rty = "c_int" if ty in ("zkp_curve_t",) else "i32"
        w(f"pub const {k}: {rty} = {v};\n")
    for k, v in DEFINES.items():
        w(f"pub const {k}: {v[1]} = {v[0]};\n")
    w("\n")
    for name, fields in structs():
        copy = "#[derive(Clone, Copy)]\n" if name in ("zkp_csr", "zkp_groth16_timing", "zkp_ctx_config") else ""
        w(f"#[repr(C)]\n{copy}pub struct {name} {{\n")
        for fn_, rt in fields:
            w(f"    pub {fn_}: {rt},\n")
        w("}\n\n")
    w("extern \"C\" {\n")
    for name, ret, params in functions():
        ps = ", ".join(f"{n}: {t}" for n, t in params)
        w(f"    pub fn {name}({ps}) -> {ret};\n")
    w("}\n")


if __name__ == "__main__":
    main()
