"""Shared helpers for the parity tests: oracle <-> ABI conversions."""
import random

import numpy as np

from ckb_zkp_amd import codec
from ckb_zkp_amd.params import get_curve
from oracle.pyref import fields as ofields
from oracle.pyref.curves import Group

OC = {"bn254": ofields.BN254, "bls12_381": ofields.BLS12_381}


def jac_to_affine(ctx, curve, group, xyz):
    """device Jacobian limbs -> oracle-style affine point via zkp_g*_into_affine."""
    c = get_curve(curve)
    xy, inf = ctx.into_affine(c, group, xyz)
    if group == 1:
        return codec.g1_from_mont(xy, [inf], c)[0]
    return codec.g2_from_mont(xy, [inf], c)[0]


def jac_limbs_to_affine_oracle(curve, group, xyz):
    """Independent normalisation on the host with the oracle (does not trust the device's inversion)."""
    c = get_curve(curve)
    G = Group(OC[c.name], group)
    f = c.fq_limbs
    Ri = pow(1 << (64 * f), -1, c.q)
    v = [x * Ri % c.q for x in codec.limbs_to_ints(np.asarray(xyz).reshape(-1, f))]
    if group == 1:
        return G.to_affine((v[0], v[1], v[2]))
    return G.to_affine(((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))


def random_points(curve, group, n, seed):
    G = Group(OC[get_curve(curve).name], group)
    rnd = random.Random(seed)
    return [G.mul(G.gen, rnd.randrange(1, G.order)) for _ in range(n)]


def to_abi_points(curve, group, pts):
    c = get_curve(curve)
    return codec.g1_to_mont(pts, c) if group == 1 else codec.g2_to_mont(pts, c)
