"""CPU: the source-only Rust shim (rust/zkp-accel) binds symbols that exist: every `pub fn zkp_*` declared in
rust/zkp-accel/src/ffi.rs is declared in include/zkp_accel.h with the same number of parameters, the status / op
constants agree with the header's enums, and the repr(C) descriptor lists the header's fields in the header's order.
(The crate itself cannot be compiled here — no Rust toolchain — so this is the consistency check that can run.)"""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HDR = (ROOT / "include" / "zkp_accel.h").read_text()
FFI = (ROOT / "rust" / "zkp-accel" / "src" / "ffi.rs").read_text()


def _strip_c(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _c_decls():
    out = {}
    for m in re.finditer(r"\b(zkp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", _strip_c(HDR), flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_decls():
    out = {}
    body = re.sub(r"//.*", "", FFI)
    for m in re.finditer(r"pub fn (zkp_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", body, flags=re.S):
        args = m.group(2).strip().rstrip(",")
        out[m.group(1)] = 0 if not args else args.count(",") + 1
    return out


def test_every_rust_extern_exists_in_the_header_with_the_same_arity():
    c, r = _c_decls(), _rust_decls()
    assert len(r) >= 20
    for name, n in r.items():
        assert name in c, f"{name} bound in ffi.rs but not declared in zkp_accel.h"
        assert c[name] == n, (name, c[name], n)


def test_constants_and_descriptor_layout_agree():
    h = _strip_c(HDR)
    for name, val in re.findall(r"pub const (ZKP_[A-Z0-9_]+): (?:c_int|i32) = (-?\d+);", FFI):
        m = re.search(rf"\b{name}\s*=\s*(-?\d+)", h)
        assert m and int(m.group(1)) == int(val), name
    c_fields = re.findall(r"\b([a-z_0-9]+)\s*;", re.search(r"typedef struct \{\s*zkp_curve_t curve;(.*?)\} zkp_groth16_pk_desc;", h, flags=re.S).group(1))
    c_fields = ["curve"] + [f for f in c_fields]
    c_fields = [f for f in " ".join(c_fields).replace("at, bt, ct", "at bt ct").split()]
    r_struct = re.search(r"pub struct zkp_groth16_pk_desc \{(.*?)\n\}", FFI, flags=re.S).group(1)
    r_fields = re.findall(r"pub ([a-z_0-9]+):", r_struct)
    # the header declares `zkp_csr at, bt, ct;` on one line
    flat = []
    for line in re.search(r"typedef struct \{\s*(zkp_curve_t curve;.*?)\} zkp_groth16_pk_desc;", h, flags=re.S).group(1).split(";"):
        line = line.strip()
        if not line:
            continue
        names = [x.strip().lstrip("*").strip() for x in line.split(",")]
        names[0] = names[0].split()[-1].lstrip("*")
        flat += names
    assert flat == r_fields, (flat, r_fields)
