"""CPU: the sanitizer row of SURVEY §5 ("C++ host: ASan/UBSan build of CPU lib + tests") for the host C++ that has no device in it:
oracle/cpu (the checker and the timed cpu_baseline: 1 200 lines of threaded field / curve / NTT / Pippenger / Marlin code) and
csrc/fs_rng.cpp (merlin / STROBE / ChaCha20 / rejection sampling: the product's Fiat–Shamir RNG).  Both are rebuilt with
-fsanitize=address,undefined and driven by the existing parity tests in a child process that LD_PRELOADs the runtime; any
AddressSanitizer / UndefinedBehaviorSanitizer report fails the test.  (The HIP host code of libzkp_accel.so gets the same treatment
on the GPU box: tools/asan_run.sh, profiles/r06_asan.txt.)"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

DRIVER = r'''
import ctypes as C, sys
import pytest
from oracle import cpu_oracle
cpu_oracle.build(sanitize=True)                     # oracle/build/libzkp_oracle_asan.so becomes THE oracle of this process
assert cpu_oracle.LIB.name == "libzkp_oracle_asan.so"
from ckb_zkp_amd import _lib
lib = C.CDLL(sys.argv[1])                           # fs_rng.cpp alone, sanitized: only its entry points get signatures
for name, (res, args) in _lib.SIGNATURES.items():
    if name.startswith(("zkp_fs_rng_", "zkp_merlin_")):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
_lib._lib = lib
sys.exit(pytest.main(["-x", "-q", "-p", "no:cacheprovider", "tests/test_fs_rng.py", "tests/test_oracle_golden.py",
                      "tests/test_oracle_marlin_cpu.py::test_cpp_marlin_oracle_matches_python_oracle", "-k", "not mimc_swapped"]))
'''


def test_oracle_cpu_and_fs_rng_are_clean_under_asan_and_ubsan(tmp_path):
    import pytest
    from oracle import cpu_oracle
    if not os.path.exists(cpu_oracle.CLANGXX):
        pytest.skip(f"{cpu_oracle.CLANGXX} not present: no sanitizer toolchain on this host")
    try:
        rt = cpu_oracle.asan_runtime()
    except RuntimeError as e:
        pytest.skip(str(e))
    fs_so = tmp_path / "libfs_rng_asan.so"
    r = subprocess.run([cpu_oracle.CLANGXX, "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libasan",
                        "-fno-omit-frame-pointer", str(ROOT / "ckb_zkp_amd" / "csrc" / "fs_rng.cpp"), "-o", str(fs_so)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    log = tmp_path / "san"
    env = dict(os.environ, LD_PRELOAD=rt, PYTHONPATH=str(ROOT),
               ASAN_OPTIONS=f"detect_leaks=0:halt_on_error=1:log_path={log}.asan",
               UBSAN_OPTIONS=f"print_stacktrace=1:halt_on_error=1:log_path={log}.ubsan")
    out = subprocess.run([sys.executable, "-c", DRIVER, str(fs_so)], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    reports = sorted(tmp_path.glob("san.*"))
    text = "\n".join(p.read_text()[:3000] for p in reports)
    assert out.returncode == 0 and not reports, (out.stdout[-1500:], out.stderr[-1500:], text)
    assert " passed" in out.stdout and "failed" not in out.stdout
    for needle in ("AddressSanitizer", "runtime error:"):
        assert needle not in out.stderr and needle not in out.stdout, out.stderr[-3000:]
