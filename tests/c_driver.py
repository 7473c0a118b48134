"""Builds tests/c/abi_driver.c (plain C99) against include/zkp_accel.h and links it with the in-tree libzkp_accel.so."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "c" / "abi_driver.c"
OUT = ROOT / "tests" / "c" / "build" / "abi_driver"


def build() -> Path:
    from ckb_zkp_amd import _lib
    lib = Path(_lib.LIB_PATH)
    assert lib.exists(), "build libzkp_accel.so first (python -m ckb_zkp_amd.build)"
    OUT.parent.mkdir(exist_ok=True)
    if OUT.exists() and OUT.stat().st_mtime >= max(SRC.stat().st_mtime, lib.stat().st_mtime,
                                                   (ROOT / "include" / "zkp_accel.h").stat().st_mtime):
        return OUT
    import os
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O1", f"-I{ROOT / 'include'}", str(SRC), "-o", str(OUT),
           f"-L{lib.parent}", "-lzkp_accel", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link,/opt/rocm/lib"] + os.environ.get("ZKP_C_DRIVER_FLAGS", "").split()   # sanitizer run: -fsanitize=address,undefined
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}                             # (the compiler itself runs unsanitized)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    return OUT
