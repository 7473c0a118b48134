"""Host-side encoders between Python integers and the ABI's limb layout (include/zkp_accel.h):
little-endian u64 limbs; field elements in Montgomery form, MSM scalars canonical."""
from __future__ import annotations

import numpy as np

from .params import CurveParams


def ints_to_limbs(xs, limbs: int) -> np.ndarray:
    """list[int] -> (len, limbs) uint64, little-endian limbs."""
    nb = limbs * 8
    buf = b"".join(int(x).to_bytes(nb, "little") for x in xs)
    return np.frombuffer(buf, dtype="<u8").reshape(-1, limbs).copy()


def limbs_to_ints(a: np.ndarray) -> list:
    a = np.ascontiguousarray(a, dtype="<u8")
    limbs = a.shape[-1]
    nb = limbs * 8
    raw = a.reshape(-1, limbs).tobytes()
    return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]


def fr_to_mont(xs, c: CurveParams) -> np.ndarray:
    R = 1 << (64 * c.fr_limbs)
    return ints_to_limbs([(x % c.r) * R % c.r for x in xs], c.fr_limbs)


def fr_from_mont(a: np.ndarray, c: CurveParams) -> list:
    Ri = pow(1 << (64 * c.fr_limbs), -1, c.r)
    return [x * Ri % c.r for x in limbs_to_ints(a)]


def fr_canonical(xs, c: CurveParams) -> np.ndarray:
    return ints_to_limbs([x % c.r for x in xs], c.fr_limbs)


def _fq_mont(v: int, c: CurveParams) -> int:
    return (v % c.q) * (1 << (64 * c.fq_limbs)) % c.q


def g1_to_mont(points, c: CurveParams):
    """list of (x, y) or None -> ((n, 2*fq_limbs) uint64, (n,) uint8 identity flags)."""
    flat, flags = [], np.zeros(len(points), dtype=np.uint8)
    for i, p in enumerate(points):
        if p is None:
            flags[i] = 1
            flat += [0, 0]
        else:
            flat += [_fq_mont(p[0], c), _fq_mont(p[1], c)]
    return ints_to_limbs(flat, c.fq_limbs).reshape(len(points), 2 * c.fq_limbs), flags


def g2_to_mont(points, c: CurveParams):
    flat, flags = [], np.zeros(len(points), dtype=np.uint8)
    for i, p in enumerate(points):
        if p is None:
            flags[i] = 1
            flat += [0, 0, 0, 0]
        else:
            (x0, x1), (y0, y1) = p
            flat += [_fq_mont(x0, c), _fq_mont(x1, c), _fq_mont(y0, c), _fq_mont(y1, c)]
    return ints_to_limbs(flat, c.fq_limbs).reshape(len(points), 4 * c.fq_limbs), flags


def _fq_list(a: np.ndarray, c: CurveParams) -> list:
    Ri = pow(1 << (64 * c.fq_limbs), -1, c.q)
    return [x * Ri % c.q for x in limbs_to_ints(a.reshape(-1, c.fq_limbs))]


def g1_from_mont(xy: np.ndarray, inf, c: CurveParams) -> list:
    v = _fq_list(xy, c)
    out = []
    for i in range(len(v) // 2):
        out.append(None if (inf is not None and inf[i]) else (v[2 * i], v[2 * i + 1]))
    return out


def g2_from_mont(xy: np.ndarray, inf, c: CurveParams) -> list:
    v = _fq_list(xy, c)
    out = []
    for i in range(len(v) // 4):
        out.append(None if (inf is not None and inf[i]) else ((v[4 * i], v[4 * i + 1]), (v[4 * i + 2], v[4 * i + 3])))
    return out
