"""ORACLE (test infrastructure only) — ctypes wrapper of oracle/build/libzkp_oracle.so (oracle/cpu/zkp_oracle.cpp).
Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "cpu" / "zkp_oracle.cpp"
LIB = HERE / "build" / "libzkp_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    LIB.parent.mkdir(exist_ok=True)
    deps = [SRC, HERE / "cpu" / "field_constants64.inc", HERE.parent / "include" / "zkp_accel.h"]
    if not force and LIB.exists() and all(d.stat().st_mtime <= LIB.stat().st_mtime for d in deps):
        return LIB
    cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-shared", "-fPIC", "-pthread", str(SRC), "-o", str(LIB)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        # -march=native objects do not travel between hosts with different ISAs: fall back to a portable build
        raise RuntimeError(f"oracle build failed:\n{r.stderr}")
    return LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        try:
            build()
            _lib = C.CDLL(str(LIB))
            _lib.oracle_hardware_threads()
        except (OSError, RuntimeError):
            build(force=True)
            _lib = C.CDLL(str(LIB))
        _lib.oracle_hardware_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def hardware_threads() -> int:
    return load().oracle_hardware_threads()


def msm(curve_id: int, group: int, xy: np.ndarray, inf, scalars: np.ndarray, threads: int = 1) -> np.ndarray:
    xy = np.ascontiguousarray(xy, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    inf = None if inf is None else np.ascontiguousarray(inf, dtype=np.uint8)
    n = min(scalars.shape[0], xy.shape[0]) if scalars.size else 0
    fq = 4 if curve_id == 0 else 6
    out = np.zeros(3 * fq * group, dtype=np.uint64)
    rc = load().oracle_msm(curve_id, group, _p(xy), _p(inf), _p(scalars), C.c_size_t(n), threads, _p(out))
    assert rc == 0
    return out


def ntt(curve_id: int, data: np.ndarray, op: int, threads: int = 1) -> np.ndarray:
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    n = a.shape[0]
    rc = load().oracle_ntt(curve_id, _p(a), n.bit_length() - 1, op, threads)
    if rc != 0:
        raise ValueError(f"oracle_ntt rc={rc}")
    return a


def fixed_base_mul(curve_id: int, group: int, base_xy: np.ndarray, scalars: np.ndarray):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.shape[0]
    fq = 4 if curve_id == 0 else 6
    out = np.zeros((n, 2 * fq * group), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    rc = load().oracle_fixed_base_mul(curve_id, group, _p(np.ascontiguousarray(base_xy, dtype=np.uint64)), _p(scalars),
                                      C.c_size_t(n), _p(out), _p(inf))
    assert rc == 0
    return out, inf


def _desc(params, inst):
    """Build a zkp_groth16_pk_desc (same struct as the product ABI) from product-side Parameters + R1csInstance."""
    from ckb_zkp_amd._lib import Groth16PkDesc
    d = Groth16PkDesc()
    keep = []

    def P(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    d.curve, d.num_inputs, d.num_aux, d.num_constraints = params.curve.cid, inst.num_inputs, inst.num_aux, \
        inst.num_constraints()
    for name, which in (("at", "a"), ("bt", "b"), ("ct", "c")):
        rp, col, cf = inst.csr(which)
        m = getattr(d, name)
        m.row_ptr, m.col, m.coeff = P(rp, np.uint32), P(col, np.uint32), P(cf, np.uint64)
    for name in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
        setattr(d, name, P(getattr(params, name), np.uint64))
    for name, fld in (("a", "a_query"), ("b_g1", "b_g1_query"), ("b_g2", "b_g2_query"), ("h", "h_query"),
                      ("l", "l_query")):
        xy, inf = getattr(params, fld)
        setattr(d, f"{name}_query", P(xy, np.uint64))
        setattr(d, f"{name}_inf", P(inf, np.uint8) if len(inf) else None)
        setattr(d, f"{name}_len", len(inf))
    return d, keep


def witness_map(params, inst, z_mont: np.ndarray, threads: int = 1) -> np.ndarray:
    d, keep = _desc(params, inst)
    n = inst.num_constraints() + inst.num_inputs
    N = 1 << max(n - 1, 0).bit_length()
    h = np.zeros((N, 4), dtype=np.uint64)
    z = np.ascontiguousarray(z_mont, dtype=np.uint64)
    rc = load().oracle_witness_map(C.byref(d), _p(z), threads, _p(h))
    assert rc == 0
    return h


def groth16_prove(params, inst, z_mont, r_mont, s_mont, threads: int = 1):
    """-> (proof limbs, inf flags[3], phase_ms[8]) ; layout identical to zkp_groth16_prove."""
    d, keep = _desc(params, inst)
    fq = params.curve.fq_limbs
    out = np.zeros(8 * fq, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    ph = np.zeros(8, dtype=np.float64)
    z = np.ascontiguousarray(z_mont, dtype=np.uint64)
    rc = load().oracle_groth16_prove(C.byref(d), _p(z), _p(np.ascontiguousarray(r_mont, dtype=np.uint64)),
                                     _p(np.ascontiguousarray(s_mont, dtype=np.uint64)), threads, _p(out), _p(inf), _p(ph))
    assert rc == 0
    return out, inf, ph
