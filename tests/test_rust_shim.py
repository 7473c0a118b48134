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


def test_rust_externs_equal_the_header_exports_with_the_same_arity():
    """ffi.rs binds EVERY export of the header and nothing else (23 of 77 before)."""
    c, r = _c_decls(), _rust_decls()
    assert set(r) == set(c), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, n in r.items():
        assert c[name] == n, (name, c[name], n)


def test_ffi_rs_is_the_generator_output():
    """rust/zkp-accel/src/ffi.rs is generated from the header (tools/gen_rust_ffi.py): a header change without a
    regenerated binding fails here."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_rust_ffi.py")], capture_output=True, text=True, check=True).stdout
    assert out == FFI


def test_shim_wrappers_call_bound_symbols_only():
    """every `ffi::zkp_*` the safe layer (lib.rs, groth16.rs, marlin.rs) calls is declared in ffi.rs; the Marlin seam and the
    multi-device wrappers exist."""
    r = _rust_decls()
    used = set()
    for f in ("lib.rs", "groth16.rs", "marlin.rs"):
        used |= set(re.findall(r"ffi::(zkp_[a-z0-9_]+)\s*\(", (ROOT / "rust" / "zkp-accel" / "src" / f).read_text()))
    assert used <= set(r), sorted(used - set(r))
    for need in ("zkp_marlin_prove", "zkp_marlin_index_upload", "zkp_groth16_prove_multi", "zkp_groth16_prove_batch_multi",
                 "zkp_ctx_create_multi", "zkp_msm_g1_mont_dev"):
        assert need in used, need
    lib = (ROOT / "rust" / "zkp-accel" / "src" / "lib.rs").read_text()
    assert "debug_assert_eq!(std::mem::size_of" not in lib          # the element-size guard is a hard assert + sealed trait
    assert (ROOT / "rust" / "patches" / "marlin-accel.diff").exists()


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


def _crate_items():
    """{module path: set of public items}: '' = lib.rs, 'groth16' = groth16.rs, ..."""
    src = ROOT / "rust" / "zkp-accel" / "src"
    lib = (src / "lib.rs").read_text()
    mods = {"": lib}
    for m in re.findall(r"^pub mod ([a-z_0-9]+);", lib, flags=re.M):
        assert (src / f"{m}.rs").exists(), m
        mods[m] = (src / f"{m}.rs").read_text()
    items = {}
    for name, text in mods.items():
        text = re.sub(r"//.*", "", text)
        found = set(re.findall(r"^\s*pub(?:\(crate\))?\s+(?:unsafe\s+)?(?:fn|struct|enum|trait|type|const|static|mod)\s+([A-Za-z_0-9]+)", text, flags=re.M))
        for enum, body in re.findall(r"pub enum ([A-Za-z_0-9]+)\s*\{(.*?)\n\}", text, flags=re.S):
            for v in re.findall(r"^\s*([A-Z][A-Za-z0-9]*)\s*(?:[,({=]|$)", body, flags=re.M):
                found.add(f"{enum}::{v}")
        for ty, body in re.findall(r"^impl(?:<[^{]*?>)?\s+([A-Za-z_0-9]+)[^{]*\{(.*?)^\}", text, flags=re.S | re.M):
            for fn in re.findall(r"pub fn ([a-z_0-9]+)", body):
                found.add(f"{ty}::{fn}")
        items[name] = found
    return items


def _resolve(path, items):
    """zkp_accel::a::b::c -> is it a module, an item of a module, or Type::assoc of an item?"""
    segs = path.split("::")
    if segs[0] in items and segs[0] != "":
        mod, rest = segs[0], segs[1:]
    else:
        mod, rest = "", segs
    if not rest:
        return True                                  # a module itself
    return "::".join(rest) in items[mod] or (len(rest) == 1 and rest[0] in items[mod])


def test_every_zkp_accel_path_in_the_patches_resolves_to_a_crate_item():
    """rust/patches/*.diff may only name things rust/zkp-accel defines (the groth16 patch used to call an
    `accel_cache` module that existed nowhere).  Checked on the added lines AND on the commentary lines of the patches:
    `zkp_accel::<path>`, `use zkp_accel::<mod>::{..}`, and `<mod>::<item>` after `use zkp_accel::<mod>;`."""
    items = _crate_items()
    assert "accel_cache" in items and {"get_or_upload", "ctx", "evict", "try_ctx"} <= items["accel_cache"]
    for diff in sorted((ROOT / "rust" / "patches").glob("*.diff")):
        text = diff.read_text()
        lines = [l[1:] for l in text.splitlines() if l.startswith(("+", "#")) and not l.startswith("+++")]
        body = "\n".join(lines)
        imported_mods = set(re.findall(r"use zkp_accel::([a-z_0-9]+);", body))
        for mod, names in re.findall(r"use zkp_accel::([a-z_0-9]+)::\{([^}]*)\}", body):
            for n in [x.strip() for x in names.split(",") if x.strip()]:
                assert _resolve(f"{mod}::{n}", items), (diff.name, mod, n)
        for path in re.findall(r"zkp_accel::((?:[A-Za-z_0-9]+)(?:::[A-Za-z_0-9]+)*)", body):
            path = re.sub(r"::$", "", path)
            if path.split("::")[0] in items and len(path.split("::")) == 1:
                continue                              # `use zkp_accel::accel_cache;`
            assert _resolve(path, items), (diff.name, path)
        for mod in imported_mods:
            assert mod in items, (diff.name, mod)
            for n in re.findall(rf"(?<![A-Za-z_0-9:]){mod}::([a-z_0-9]+)\s*\(", body):
                assert n in items[mod], (diff.name, mod, n)
    # the seam's call shape: the closure handed to get_or_upload builds the key with DeviceProvingKey::upload(accel_cache::ctx(), ..)
    g = (ROOT / "rust" / "patches" / "groth16-accel.diff").read_text()
    assert "use zkp_accel::accel_cache;" in g and "accel_cache::get_or_upload(params" in g and "DeviceProvingKey::upload(accel_cache::ctx()" in g


def test_accel_cache_identifies_keys_by_content_not_only_by_address():
    """a `Parameters` reloaded at the same address must not hit the old device key.  Static shape of the fix:
    `get_or_upload` takes a `Fingerprint` (buffer spans, circuit shape, content digest), compares it on a hit, uploads under the
    entry's own lock with the table unlocked, recovers poisoned locks, and leaves the process environment alone."""
    src = (ROOT / "rust" / "zkp-accel" / "src" / "accel_cache.rs").read_text()
    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("//"))
    assert "pub struct Fingerprint" in code and "pub fn of<" in code
    for field in ("a_query", "h_query", "l_query", "num_inputs", "num_aux", "nnz", "digest"):
        assert re.search(rf"pub {field}:", code), field
    assert re.search(r"pub fn get_or_upload<[^>]*>\(params: &P, print: Fingerprint, upload: F\)", code)
    assert "s.print == print" in code                        # the hit test
    assert "set_var" not in code and "into_inner" in code
    # the table guard is dropped (end of the block that produced `slot`) before upload() runs
    a = code.index("let slot = {")
    assert a < code.index("\n    };", a) < code.index("upload()?")
    g = (ROOT / "rust" / "patches" / "groth16-accel.diff").read_text()
    assert "accel_cache::Fingerprint::of(&key" in g and "accel_cache::get_or_upload(params, print," in g
