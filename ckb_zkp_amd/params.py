"""Curve / field parameters used by the host side (product code — independent of oracle/).

BN254 ("BN256", alt_bn128 = ark-bn254) and BLS12-381 (ark-bls12-381): the two pairing engines the
reference's provers are instantiated with (/root/reference/groth16/tests/mini.rs:1).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class CurveParams:
    name: str
    cid: int            # zkp_curve_t
    r: int              # scalar field
    q: int              # base field
    fr_limbs: int       # u64 limbs
    fq_limbs: int
    fr_generator: int   # Fr::multiplicative_generator() (coset shift)
    two_adicity: int
    g1: tuple           # standard generator (x, y)
    g2: tuple           # ((x0, x1), (y0, y1))

    @property
    def scalar_bits(self) -> int:
        return self.r.bit_length()


BN254 = CurveParams(
    "bn254", 0,
    21888242871839275222246405745257275088548364400416034343698204186575808495617,
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    4, 4, 5, 28, (1, 2),
    ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
      11559732032986387107991004021392285783925812861821192530917403151452391805634),
     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
      4082367875863433681332203403145435568316851327593401208105741076214120093531)))

BLS12_381 = CurveParams(
    "bls12_381", 1,
    52435875175126190479447740508185965837690552500527637822603658699938581184513,
    0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    4, 6, 7, 32,
    (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
     0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
      0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
     (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
      0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)))

CURVES = {"bn254": BN254, "bn256": BN254, "bls12_381": BLS12_381, 0: BN254, 1: BLS12_381}


def get_curve(c) -> CurveParams:
    if isinstance(c, CurveParams):
        return c
    return CURVES[c]
