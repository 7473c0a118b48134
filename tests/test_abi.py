"""CPU: the C-ABI library loads and exports exactly the symbols include/zkp_accel.h declares (no compute calls)."""
import ctypes
import re
from pathlib import Path

from ckb_zkp_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (ROOT / "include" / "zkp_accel.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(zkp_[a-z0-9_]+)\s*\(", text))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in zkp_accel.h but not exported by libzkp_accel.so"
    assert syms == set(_lib.SIGNATURES), (syms ^ set(_lib.SIGNATURES))


def test_status_strings_and_no_cpu_fallback():
    lib = _lib.load()
    assert lib.zkp_status_string(0) == b"ok"
    assert b"PolynomialDegreeTooLarge" in lib.zkp_status_string(-3)
    assert b"no CPU fallback" in lib.zkp_status_string(-5)
    assert lib.zkp_version().startswith(b"zkp_accel")
    # NULL context is rejected, never dereferenced
    assert lib.zkp_ctx_sync(None) == -1
    assert lib.zkp_ntt(None, 0, None, 3, 0) == -1


def test_desc_struct_layout_matches_header():
    """ctypes mirror of zkp_groth16_pk_desc must have the C layout (x86-64 SysV): 4 ints, 3 CSR triples, 5+15 words."""
    assert ctypes.sizeof(_lib.Csr) == 24
    assert ctypes.sizeof(_lib.Groth16PkDesc) == 16 + 3 * 24 + 5 * 8 + 5 * 24
    # zkp_groth16_timing: ask the C compiler (sizeof + offset of every member) and compare with the ctypes mirror
    import subprocess
    import tempfile
    fields = [f[0] for f in _lib.Groth16Timing._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"zkp_accel.h\"\nint main(void){printf(\"%zu\", sizeof(zkp_groth16_timing));" + \
        "".join(f'printf(" %zu", offsetof(zkp_groth16_timing, {f}));' for f in fields) + \
        "printf(\" %zu %zu\", sizeof(zkp_groth16_pk_desc), sizeof(zkp_csr));return 0;}"
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "t.c").write_text(prog)
        subprocess.run(["gcc", "-std=c99", f"-I{ROOT / 'include'}", str(Path(d) / "t.c"), "-o", str(Path(d) / "t")], check=True)
        nums = [int(x) for x in subprocess.run([str(Path(d) / "t")], capture_output=True, text=True, check=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(_lib.Groth16Timing)
    assert nums[1:1 + len(fields)] == [getattr(_lib.Groth16Timing, f).offset for f in fields]
    assert nums[-2:] == [ctypes.sizeof(_lib.Groth16PkDesc), ctypes.sizeof(_lib.Csr)]
    # zkp_marlin_timing likewise
    mf = [f[0] for f in _lib.MarlinTiming._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"zkp_accel.h\"\nint main(void){printf(\"%zu\", sizeof(zkp_marlin_timing));" + \
        "".join(f'printf(" %zu", offsetof(zkp_marlin_timing, {f}));' for f in mf) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "t.c").write_text(prog)
        subprocess.run(["gcc", "-std=c99", f"-I{ROOT / 'include'}", str(Path(d) / "t.c"), "-o", str(Path(d) / "t")], check=True)
        nums = [int(x) for x in subprocess.run([str(Path(d) / "t")], capture_output=True, text=True, check=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(_lib.MarlinTiming)
    assert nums[1:] == [getattr(_lib.MarlinTiming, f).offset for f in mf]


def test_product_does_not_import_oracle():
    """The product path must not reach into oracle/ (it is the checker, never the thing shipped)."""
    for f in (ROOT / "ckb_zkp_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".hpp", ".cpp", ".h", ".inc") and f.is_file():
            txt = f.read_text()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
            assert "libzkp_oracle" not in txt and "cpu_oracle" not in txt and "pyref" not in txt, f


def test_plain_c_program_links_against_the_header_and_library():
    """A C99 translation unit (gcc -std=c99 -pedantic -Werror) that includes include/zkp_accel.h links against
    libzkp_accel.so and runs: no ctypes, no C++ in between.  Without a GPU the context creation must fail LOUDLY with
    ZKP_ERR_DEVICE (there is no CPU fallback); with one it succeeds."""
    import subprocess
    from tests import c_driver
    exe = c_driver.build()
    r = subprocess.run([str(exe), "probe"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "version=zkp_accel" in r.stdout
    assert ("ctx=0 " in r.stdout) or ("ctx=-5 " in r.stdout and "no CPU fallback" in r.stdout), r.stdout
