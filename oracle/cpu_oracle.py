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
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


def asan_runtime() -> str:
    """the shared AddressSanitizer runtime of the ROCm clang (to LD_PRELOAD into a Python process that loads a sanitized .so)"""
    r = subprocess.run([CLANGXX, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    if r.returncode != 0 or not Path(p).exists():
        import glob
        hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
        if not hits:
            raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
        p = hits[-1]
    return p


def build(force: bool = False, sanitize: bool = False) -> Path:
    """sanitize=True: an ASan + UBSan build (oracle/build/libzkp_oracle_asan.so, clang++ of the ROCm toolchain with its shared
    runtime, -O1) for tests/test_sanitizers.py — the process that loads it LD_PRELOADs the runtime (asan_runtime())."""
    global LIB
    if sanitize:
        LIB = HERE / "build" / "libzkp_oracle_asan.so"
    LIB.parent.mkdir(exist_ok=True)
    deps = [SRC, HERE / "cpu" / "field_constants64.inc", HERE / "cpu" / "marlin_oracle.inc", HERE.parent / "include" / "zkp_accel.h"]
    if not force and LIB.exists() and all(d.stat().st_mtime <= LIB.stat().st_mtime for d in deps):
        return LIB
    cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-shared", "-fPIC", "-pthread", str(SRC), "-o", str(LIB)]
    if sanitize:
        cmd = [CLANGXX, "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-pthread", "-fsanitize=address,undefined", "-shared-libasan",
               "-fno-omit-frame-pointer", str(SRC), "-o", str(LIB)]
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


def fr_dot(curve, a_mont: np.ndarray, b_mont: np.ndarray, threads: int = 0) -> int:
    """sum_i a_i b_i mod r as a canonical Python int; curve: oracle.pyref.fields Curve (or anything with .cid and .r)."""
    a = np.ascontiguousarray(a_mont, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b_mont, dtype=np.uint64).reshape(-1, 4)
    n = min(a.shape[0], b.shape[0])
    out = np.zeros(4, dtype=np.uint64)
    rc = load().oracle_fr_dot(curve.cid, _p(a), _p(b), C.c_size_t(n), threads or hardware_threads(), _p(out))
    assert rc == 0
    return int.from_bytes(out.tobytes(), "little") * pow(1 << 256, -1, curve.r) % curve.r


def _mont1(curve, x: int) -> np.ndarray:
    return np.frombuffer(((x % curve.r) * (1 << 256) % curve.r).to_bytes(32, "little"), dtype="<u8").copy()


def fr_powers(curve, base: int, n: int, first: int = 1, threads: int = 0) -> np.ndarray:
    """(n, 4) Montgomery: first * base^i"""
    out = np.zeros((n, 4), dtype=np.uint64)
    b, f = _mont1(curve, base), _mont1(curve, first)           # kept alive across the call
    rc = load().oracle_fr_powers(curve.cid, _p(b), _p(f), C.c_size_t(n), threads or hardware_threads(), _p(out))
    assert rc == 0
    return out


def lagrange_coeffs(curve, log_n: int, tau: int, threads: int = 0) -> np.ndarray:
    """u_k = L_k(tau) over the radix-2 domain of size 2^log_n, (N, 4) Montgomery: the inverse transform of (tau^j)_j
    (u_k = 1/N sum_j tau^j w^-jk = Z(tau) w^k / (N (tau - w^k)); r1cs_to_qap.rs:58-110 evaluates the same by batch inversion)"""
    return ntt(curve.cid, fr_powers(curve, tau, 1 << log_n, threads=threads), 1, threads=threads or hardware_threads())


def fixed_base_mul(curve_id: int, group: int, base_xy: np.ndarray, scalars: np.ndarray):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.shape[0]
    fq = 4 if curve_id == 0 else 6
    out = np.zeros((n, 2 * fq * group), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    base_xy = np.ascontiguousarray(base_xy, dtype=np.uint64)       # named: a temporary would be freed before the call
    rc = load().oracle_fixed_base_mul(curve_id, group, _p(base_xy), _p(scalars),
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


def groth16_prove(params, inst, z_mont, r_mont, s_mont, threads: int = 1, want_h: bool = False):
    """-> (proof limbs, inf flags[3], phase_ms[8]) ; layout identical to zkp_groth16_prove.  want_h: a fourth element, the quotient
    h = witness_map(z) ((N, 4) Montgomery) the proof was made from (saves the separate oracle witness_map pass of the full-size tests)."""
    d, keep = _desc(params, inst)
    fq = params.curve.fq_limbs
    out = np.zeros(8 * fq, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    ph = np.zeros(8, dtype=np.float64)
    z = np.ascontiguousarray(z_mont, dtype=np.uint64)
    r_mont, s_mont = np.ascontiguousarray(r_mont, dtype=np.uint64), np.ascontiguousarray(s_mont, dtype=np.uint64)
    if want_h:
        n = inst.num_constraints() + inst.num_inputs
        h = np.zeros((1 << max(n - 1, 0).bit_length(), 4), dtype=np.uint64)
        rc = load().oracle_groth16_prove_h(C.byref(d), _p(z), _p(r_mont), _p(s_mont), threads, _p(out), _p(inf), _p(ph), _p(h))
        assert rc == 0
        return out, inf, ph, h
    rc = load().oracle_groth16_prove(C.byref(d), _p(z), _p(r_mont), _p(s_mont), threads, _p(out), _p(inf), _p(ph))
    assert rc == 0
    return out, inf, ph


# ------------------------------------------------------------------ Marlin (oracle/cpu/marlin_oracle.inc)
class _Csr(C.Structure):                 # == zkp_csr (include/zkp_accel.h)
    _fields_ = [("row_ptr", C.c_void_p), ("col", C.c_void_p), ("coeff", C.c_void_p)]


class _MarlinDesc(C.Structure):
    _fields_ = [("curve", C.c_int32), ("num_inputs", C.c_uint32), ("num_aux", C.c_uint32), ("num_constraints", C.c_uint32),
                ("a", _Csr), ("b", _Csr), ("c", _Csr)]


class _MarlinRand(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w", "z_a", "z_b", "mask", "blind_w", "blind_z_a", "blind_z_b", "blind_g_1",
                                          "blind_shifted_g_1")]


MARLIN_LABELS = ["w", "z_a", "z_b", "mask", "t", "g_1", "h_1", "g_2", "h_2"]
MARLIN_INDEX_LABELS = [f"{m}_{k}" for m in "abc" for k in ("row", "col", "val", "row_col")]


class MarlinOracle:
    """C++ restatement of marlin::index + create_random_proof, driven phase by phase.  curve: oracle.pyref.fields Curve.
    inst: anything with num_inputs, num_aux, num_constraints() and csr(which) -> (row_ptr u32, col u32, coeff (nnz, 4) u64
    Montgomery) AS SYNTHESISED (make_matrices_square / balance_matrices / the column sort happen in C++).
    srs: ((g_xy, g_inf), (gamma_g_xy, gamma_g_inf)) affine Montgomery host arrays of the committer key (an input)."""

    def __init__(self, curve, inst, srs=None, threads: int = 0):
        lib = load()
        lib.oracle_marlin_new.restype = C.c_void_p
        lib.oracle_marlin_poly.restype = C.c_long
        self.lib, self.curve = lib, curve
        self.threads = threads or hardware_threads()
        d = _MarlinDesc()
        d.curve, d.num_inputs, d.num_aux, d.num_constraints = curve.cid, inst.num_inputs, inst.num_aux, inst.num_constraints()
        keep = []
        for name in "abc":
            ptr, col, cf = inst.csr(name)
            arrs = [np.ascontiguousarray(ptr, dtype=np.uint32), np.ascontiguousarray(col if len(col) else np.zeros(1), dtype=np.uint32),
                    np.ascontiguousarray(cf if len(cf) else np.zeros((1, 4)), dtype=np.uint64)]
            keep += arrs
            m = getattr(d, name)
            m.row_ptr, m.col, m.coeff = (a.ctypes.data for a in arrs)
        self.h = C.c_void_p(lib.oracle_marlin_new(C.byref(d), self.threads))
        if not self.h:
            raise RuntimeError("oracle_marlin_new failed")
        info = (C.c_uint64 * 8)()
        assert lib.oracle_marlin_info(self.h, info) == 0
        self.xs, self.hs, self.ks, self.bs, self.max_degree, self.num_non_zeros, self.n, self.pad_aux = (int(v) for v in info)
        self.fq = (curve.q.bit_length() + 63) // 64
        self._srs = None
        if srs is not None:
            self.set_srs(*srs)

    def set_options(self, threads=None, concurrent_commits=True):
        if threads:
            self.threads = threads
        assert self.lib.oracle_marlin_set_options(self.h, self.threads, int(concurrent_commits)) == 0

    def set_srs(self, g, gamma_g):
        self._srs = [np.ascontiguousarray(g[0], dtype=np.uint64), np.ascontiguousarray(g[1], dtype=np.uint8),
                     np.ascontiguousarray(gamma_g[0], dtype=np.uint64), np.ascontiguousarray(gamma_g[1], dtype=np.uint8)]
        a = self._srs
        assert self.lib.oracle_marlin_set_srs(self.h, _p(a[0]), _p(a[1]), C.c_size_t(len(a[1])), _p(a[2]), _p(a[3]),
                                              C.c_size_t(len(a[3]))) == 0

    def free(self):
        if self.h:
            self.lib.oracle_marlin_free(self.h)
            self.h = None

    # ---- conversions (python ints <-> Montgomery limbs), oracle-side
    def _mont(self, xs):
        r, R = self.curve.r, 1 << 256
        return np.frombuffer(b"".join(((int(x) % r) * R % r).to_bytes(32, "little") for x in xs), dtype="<u8").reshape(-1, 4).copy()

    def _ints(self, a):
        r = self.curve.r
        Ri = pow(1 << 256, -1, r)
        raw = np.ascontiguousarray(a, dtype="<u8").tobytes()
        return [int.from_bytes(raw[i:i + 32], "little") * Ri % r for i in range(0, len(raw), 32)]

    def _points(self, xy, inf, count):
        q, nb = self.curve.q, 8 * self.fq
        Ri = pow(1 << (64 * self.fq), -1, q)
        raw = np.ascontiguousarray(xy, dtype="<u8").tobytes()
        out = []
        for k in range(count):
            if inf[k]:
                out.append(None)
            else:
                o = 2 * nb * k
                out.append((int.from_bytes(raw[o:o + nb], "little") * Ri % q, int.from_bytes(raw[o + nb:o + 2 * nb], "little") * Ri % q))
        return out

    def _call_points(self, fn, count, *args):
        xy = np.zeros((count, 2 * self.fq), dtype=np.uint64)
        inf = np.zeros(count, dtype=np.uint8)
        rc = fn(self.h, *args, _p(xy), _p(inf))
        if rc != 0:
            raise RuntimeError(f"oracle marlin phase failed rc={rc}")
        return self._points(xy, inf, count)

    # ---- phases
    def index_commitments(self) -> dict:
        pts = self._call_points(self.lib.oracle_marlin_index_commit, 12)
        return {l: (p, None) for l, p in zip(MARLIN_INDEX_LABELS, pts)}

    def round1(self, x, w, rnd):
        """x: formatted inputs (ints, the leading one included); w: witness ints or (n_w, 4) Montgomery; rnd as oracle/pyref/marlin.py
        (mask: ints or (3|H|, 4) Montgomery)."""
        xm = self._mont(x)
        wm = np.ascontiguousarray(w, dtype=np.uint64) if isinstance(w, np.ndarray) else self._mont(w)
        mask = rnd["mask"]
        mk = np.ascontiguousarray(mask, dtype=np.uint64) if isinstance(mask, np.ndarray) else self._mont(mask)
        assert mk.shape[0] == 3 * self.hs
        keep = [xm, wm, mk]
        R = _MarlinRand()
        for k, v in (("w", rnd["w"]), ("z_a", rnd["z_a"]), ("z_b", rnd["z_b"]), ("blind_w", rnd["blind"]["w"]),
                     ("blind_z_a", rnd["blind"]["z_a"]), ("blind_z_b", rnd["blind"]["z_b"]), ("blind_g_1", rnd["blind"]["g_1"]),
                     ("blind_shifted_g_1", rnd["blind_shifted"]["g_1"])):
            a = self._mont(v)
            keep.append(a)
            setattr(R, k, a.ctypes.data)
        R.mask = mk.ctypes.data
        pts = self._call_points(self.lib.oracle_marlin_round1, 4, _p(xm), _p(wm), C.c_size_t(wm.shape[0]), C.byref(R))
        return [(p, None) for p in pts]

    def round2(self, alpha, eta_a, eta_b, eta_c):
        ch = self._mont([alpha, eta_a, eta_b, eta_c])
        t, g1, h1, g1s = self._call_points(self.lib.oracle_marlin_round2, 4, _p(ch))
        return [(t, None), (g1, g1s), (h1, None)]

    def round3(self, beta):
        b = self._mont([beta])
        g2, h2, g2s = self._call_points(self.lib.oracle_marlin_round3, 3, _p(b))
        return [(g2, g2s), (h2, None)]

    def evaluations(self, gamma):
        g = self._mont([gamma])
        out = np.zeros((21, 4), dtype=np.uint64)
        assert self.lib.oracle_marlin_evaluate(self.h, _p(g), _p(out)) == 0
        return self._ints(out)

    def open(self, xi):
        x = self._mont([xi])
        w_xy = np.zeros((2, 2 * self.fq), dtype=np.uint64)
        w_inf = np.zeros(2, dtype=np.uint8)
        rv = np.zeros((2, 4), dtype=np.uint64)
        has = np.zeros(2, dtype=np.uint8)
        k = self.lib.oracle_marlin_open(self.h, _p(x), _p(w_xy), _p(w_inf), _p(rv), _p(has))
        if k < 0:
            raise RuntimeError(f"oracle_marlin_open rc={k}")
        pts, rvs = self._points(w_xy, w_inf, k), self._ints(rv)
        return [(pts[i], rvs[i] if has[i] else None) for i in range(k)]

    def phase_seconds(self):
        out = np.zeros(8, dtype=np.float64)
        assert self.lib.oracle_marlin_phase_seconds(self.h, _p(out)) == 0
        return dict(zip(("round1_polys", "round1_commit", "round2_polys", "round2_commit", "round3_polys", "round3_commit",
                         "evaluations", "open"), (float(v) for v in out)))

    def poly(self, label):
        i = (MARLIN_LABELS + MARLIN_INDEX_LABELS).index(label)
        ln = self.lib.oracle_marlin_poly(self.h, i, None)
        out = np.zeros((max(ln, 1), 4), dtype=np.uint64)
        self.lib.oracle_marlin_poly(self.h, i, _p(out))
        return self._ints(out[:ln])

    # ---- marlin::create_random_proof (lib.rs:97-181) / create_proof with supplied verifier messages
    def create_proof(self, x, w, rnd, challenger):
        """challenger: oracle.pyref.marlin.FiatShamirChallenger (the reference's transcript) or FixedChallenger.
        -> dict(commitments, evaluations, opening_proofs, query, challenges, seconds) in oracle/pyref/marlin.py's shapes."""
        import time
        t0 = time.perf_counter()
        comms = {}
        c1 = self.round1(x, w, rnd)
        comms.update(zip(MARLIN_LABELS[0:4], c1))
        alpha, ea, eb, ec = challenger.first(c1)
        c2 = self.round2(alpha, ea, eb, ec)
        comms.update(zip(MARLIN_LABELS[4:7], c2))
        beta = challenger.second(c2)
        c3 = self.round3(beta)
        comms.update(zip(MARLIN_LABELS[7:9], c3))
        gamma = challenger.third(c3)
        evals = self.evaluations(gamma)
        xi = challenger.opening(evals)
        proofs = self.open(xi)
        dt = time.perf_counter() - t0
        query = sorted([(l, beta) for l in MARLIN_LABELS[:7]] + [(l, gamma) for l in MARLIN_LABELS[7:] + MARLIN_INDEX_LABELS])
        return dict(commitments=comms, evaluations=evals, opening_proofs=proofs, query=query,
                    challenges=dict(alpha=alpha, eta_a=ea, eta_b=eb, eta_c=ec, beta=beta, gamma=gamma, xi=xi), seconds=dt,
                    phase_seconds=self.phase_seconds())


class SynthesisedInstance:
    """CSR view of an oracle/pyref ConstraintSystem after synthesis (rows of (coeff, (kind, index)); kind 0 = input)."""

    def __init__(self, cs):
        self.curve, self.num_inputs, self.num_aux = cs.curve, cs.num_inputs, cs.num_aux
        self._rows = {"a": cs.at, "b": cs.bt, "c": cs.ct}
        self.x, self.w = list(cs.input_assignment), list(cs.aux_assignment)

    def num_constraints(self):
        return len(self._rows["a"])

    def csr(self, which):
        rows = self._rows[which]
        r, R = self.curve.r, 1 << 256
        ptr = np.zeros(len(rows) + 1, dtype=np.uint32)
        col, cf = [], []
        for i, row in enumerate(rows):
            for coeff, (kind, j) in row:
                col.append(j if kind == 0 else self.num_inputs + j)
                cf.append(((coeff % r) * R % r).to_bytes(32, "little"))
            ptr[i + 1] = len(col)
        return ptr, np.asarray(col, dtype=np.uint32), np.frombuffer(b"".join(cf), dtype="<u8").reshape(-1, 4).copy()


def srs_from_pyref(pp):
    """oracle/pyref/kzg10.setup parameters -> the host arrays MarlinOracle.set_srs takes"""
    curve = pp["curve"]
    q = curve.q
    fq = (q.bit_length() + 63) // 64
    R = 1 << (64 * fq)

    def pack(points):
        xy = np.zeros((len(points), 2 * fq), dtype=np.uint64)
        inf = np.zeros(len(points), dtype=np.uint8)
        for i, p in enumerate(points):
            if p is None:
                inf[i] = 1
            else:
                xy[i] = np.frombuffer((p[0] * R % q).to_bytes(8 * fq, "little") + (p[1] * R % q).to_bytes(8 * fq, "little"), dtype="<u8")
        return xy, inf

    return pack(pp["powers_of_g"]), pack(pp["powers_of_gamma_g"])
