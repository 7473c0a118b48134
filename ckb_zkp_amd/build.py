"""Builds ckb_zkp_amd/lib/libzkp_accel.so (hand-written HIP for gfx950) with hipcc, in-tree.

    python -m ckb_zkp_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
# A/B builds: ZKP_BUILD_TAG=<tag> ZKP_BUILD_DEFS="-DFOO -DBAR=1" python -m ckb_zkp_amd.build  ->  variants/<tag>/libzkp_accel.so
# (own object directory; select it at run time with ZKP_ACCEL_LIB=variants/<tag>/libzkp_accel.so).  The shipped library is
# the untagged build.
TAG = os.environ.get("ZKP_BUILD_TAG", "")
EXTRA_DEFS = os.environ.get("ZKP_BUILD_DEFS", "").split()
# Sanitizer build of the HOST code (tools/asan_run.sh): ZKP_BUILD_TAG=asan
#   ZKP_BUILD_DEFS="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g"  ZKP_BUILD_LDFLAGS="-fsanitize=address,undefined -shared-libasan"
# (-fno-gpu-sanitize: device code is compiled as always; the runtime is LD_PRELOADed into the Python process that loads the library)
EXTRA_LDFLAGS = os.environ.get("ZKP_BUILD_LDFLAGS", "").split()
LIBDIR = (ROOT.parent / "variants" / TAG) if TAG else ROOT / "lib"
OBJDIR = LIBDIR / "obj"
LIB = LIBDIR / "libzkp_accel.so"
CONFIGS = [(0, 1), (0, 2), (1, 1), (1, 2)]        # (curve, group): BN254 G1/G2, BLS12-381 G1/G2
# Full unrolling of the limb loops is part of the design (every index a compile-time constant, no scratch arrays).  The default
# pragma-unroll threshold (16 K instructions) silently left the 14-limb mul_add4 of the BLS12-381 G2 accumulator rolled, with
# its operands in scratch: 81 instead of 18 ms per 2^22 B-query MSM.  -Wpass-failed stays visible so that this cannot recur.
UNROLL = ["-mllvm", "-pragma-unroll-threshold=1000000"]
# (source, object name, extra flags).  ZKP_INLINE_MUL: the Montgomery multiplier is inlined into the hot loops
# (NTT butterflies, BN254 bucket accumulation); everywhere else one out-of-line copy per field is called.
UNITS = [("ntt.hip", "ntt.o", ["-DZKP_INLINE_MUL"]),
         ("poly.hip", "poly.o", ["-DZKP_INLINE_MUL"]),
         ("msm.hip", "msm.o", []),
         ("groth16.hip", "groth16.o", ["-DZKP_INLINE_MUL"]),
         ("capi.hip", "capi.o", []),
         ("bench_kern.hip", "bench_kern.o", ["-DZKP_INLINE_MUL"] + UNROLL),
         ("fs_rng.cpp", "fs_rng.o", []),
         ("marlin.hip", "marlin.o", [])]
for _c, _g in CONFIGS:
    _d = [f"-DZKP_CFG_CURVE={_c}", f"-DZKP_CFG_GROUP={_g}"]
    UNITS.append(("msm_group.hip", f"msm_group_c{_c}{_g}.o", _d + (["-DZKP_INLINE_MUL"] if (_c, _g) in ((0, 1), (0, 2), (1, 1)) else [])))
    _acc = []
    if _g == 1 and not os.environ.get("ZKP_BUILD_SATURATED_ACC"):
        _acc = ["-DZKP_ACC_UNSAT"]                    # G1: unsaturated-limb accumulator (unsat_dev.hpp)
    if (_c, _g) == (0, 2) and not os.environ.get("ZKP_BUILD_SATURATED_G2"):
        _acc = ["-DZKP_ACC_UNSAT_G2"]                 # BN254 G2: Fq2 on unsaturated limbs, lazily reduced schoolbook products
    if (_c, _g) == (1, 2) and not os.environ.get("ZKP_BUILD_SATURATED_G2"):
        _acc = ["-DZKP_ACC_UNSAT_G2"]                 # BLS12-381 G2 likewise (336 VGPRs, no spills: 14.4 vs 18.6 ms per 2^22 B-query accumulate)
    UNITS.append(("msm_acc.hip", f"msm_acc_c{_c}{_g}.o", _d + ["-DZKP_INLINE_MUL"] + _acc + UNROLL))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-result",
         "-ffp-contract=off"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(d).stat().st_mtime <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    LIBDIR.mkdir(parents=True, exist_ok=True)
    OBJDIR.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.hpp")) + sorted(CSRC.glob("*.inc")) + \
        [ROOT.parent / "include" / "zkp_accel.h"]
    hipcc = _hipcc()

    def compile_one(unit):
        src, oname, extra = unit
        s, o = CSRC / src, OBJDIR / oname
        if not force and _newer(o, [s, Path(__file__)] + headers):          # build.py: a change of flags rebuilds
            return o, 0.0
        import time
        t0 = time.time()
        unit_defs = os.environ.get("ZKP_BUILD_DEFS_" + Path(src).stem, "").split()     # e.g. ZKP_BUILD_DEFS_msm_group="-DFOO"
        cmd = [hipcc, *FLAGS, *extra, *EXTRA_DEFS, *unit_defs, "-c", str(s), "-o", str(o)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src} {extra}:\n{r.stdout}\n{r.stderr}")
        if verbose and "warning" in r.stderr:
            print(f"[build] {oname}: compiler warnings\n{r.stderr}", file=sys.stderr)
        return o, time.time() - t0

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, UNITS))
    objs = [o for o, _ in results]
    if verbose:
        for (o, dt), u in zip(results, UNITS):
            print(f"[build] {u[1]}: {'cached' if dt == 0 else '%.1fs' % dt}", file=sys.stderr)
    if force or not _newer(LIB, objs):
        r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *EXTRA_LDFLAGS, "-o", str(LIB), *map(str, objs)],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
