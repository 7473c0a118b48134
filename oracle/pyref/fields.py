"""ORACLE (test infrastructure only) — curve/field constants and big-int field helpers.

Restates the *mathematical contract* of ark-ff / ark-bn254 / ark-bls12-381 "0.2" (third-party crates
the reference depends on: /root/reference/groth16/Cargo.toml:20-28, curve/Cargo.toml:16-18; no lock
file, not vendored).  Nothing here is copied from arkworks; every constant is re-derived or checked by
`self_check()` (primality, two-adicity, generator orders, Montgomery INV).

Representation contract used across the repo:
  * Fr / Fq elements travel as little-endian u64 limbs of  a*R mod p  (Montgomery), R = 2^(64*limbs).
  * MSM scalars travel as little-endian u64 limbs of the CANONICAL integer (ark `into_repr()`).
"""
from __future__ import annotations

from dataclasses import dataclass


def _is_probable_prime(n: int) -> bool:
    if n < 2:
        return False
    small = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37]
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


@dataclass(frozen=True)
class Curve:
    name: str
    cid: int               # zkp_curve_t value in include/zkp_accel.h
    r: int                 # scalar field modulus (Fr)
    q: int                 # base field modulus (Fq)
    fr_limbs: int          # u64 limbs
    fq_limbs: int
    fr_generator: int      # multiplicative generator of Fr == ark `multiplicative_generator()` == coset shift
    two_adicity: int
    g1_b: int              # y^2 = x^3 + b
    g1_gen: tuple
    g2_b: tuple            # Fq2 element (c0, c1), Fq2 = Fq[u]/(u^2+1)
    g2_gen: tuple          # ((x0,x1),(y0,y1))
    # pairing parameters (pairing.py)
    x_param: int           # BN: u ; BLS: |x| (x is negative for BLS12-381)
    xi: tuple              # Fq6 non-residue (c0,c1) in Fq2: BN254 9+u ; BLS12-381 1+u
    twist_is_d: bool       # BN254: D-type twist ; BLS12-381: M-type

    @property
    def fr_R(self) -> int:
        return 1 << (64 * self.fr_limbs)

    @property
    def fq_R(self) -> int:
        return 1 << (64 * self.fq_limbs)

    @property
    def root_of_unity(self) -> int:
        """2^two_adicity-th primitive root: g^((r-1)/2^s) (== ark TWO_ADIC_ROOT_OF_UNITY)."""
        return pow(self.fr_generator, (self.r - 1) >> self.two_adicity, self.r)


BN254 = Curve(
    name="bn254", cid=0,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
    fr_limbs=4, fq_limbs=4, fr_generator=5, two_adicity=28,
    g1_b=3, g1_gen=(1, 2),
    g2_b=(19485874751759354771024239261021720505790618469301721065564631296452457478373,
          266929791119991161246907387137283842545076965332900288569378510910307636690),
    g2_gen=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
             11559732032986387107991004021392285783925812861821192530917403151452391805634),
            (8495653923123431417604973247489272438418190587263600148770280649306958101930,
             4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    x_param=4965661367192848881, xi=(9, 1), twist_is_d=True,
)

BLS12_381 = Curve(
    name="bls12_381", cid=1,
    r=52435875175126190479447740508185965837690552500527637822603658699938581184513,
    q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    fr_limbs=4, fq_limbs=6, fr_generator=7, two_adicity=32,
    g1_b=4,
    g1_gen=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
            0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    g2_b=(4, 4),
    g2_gen=((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
             0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
            (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
             0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)),
    x_param=0xd201000000010000, xi=(1, 1), twist_is_d=False,
)

CURVES = {"bn254": BN254, "bls12_381": BLS12_381, 0: BN254, 1: BLS12_381}


# ---------------------------------------------------------------- limb codecs
def to_limbs(x: int, n: int) -> list:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(limbs) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def to_mont(x: int, p: int, limbs: int) -> int:
    return (x << (64 * limbs)) % p


def from_mont(x: int, p: int, limbs: int) -> int:
    return x * pow(1 << (64 * limbs), -1, p) % p


def mont_inv64(p: int) -> int:
    """-p^{-1} mod 2^64 (ark `INV`)."""
    return (-pow(p, -1, 1 << 64)) % (1 << 64)


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1), tuples (c0,c1)
def f2_add(a, b, p):
    return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)


def f2_sub(a, b, p):
    return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)


def f2_neg(a, p):
    return ((-a[0]) % p, (-a[1]) % p)


def f2_mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_sqr(a, p):
    return ((a[0] + a[1]) * (a[0] - a[1]) % p, 2 * a[0] * a[1] % p)


def f2_inv(a, p):
    n = pow((a[0] * a[0] + a[1] * a[1]) % p, -1, p)
    return (a[0] * n % p, (-a[1]) * n % p)


def f2_scalar(a, k, p):
    return (a[0] * k % p, a[1] * k % p)


class FieldOps:
    """Uniform op-table so curve code is generic over Fq (ints) and Fq2 (tuples)."""

    def __init__(self, p: int, ext: int):
        self.p, self.ext = p, ext
        if ext == 1:
            self.zero, self.one = 0, 1
            self.add = lambda a, b: (a + b) % p
            self.sub = lambda a, b: (a - b) % p
            self.neg = lambda a: (-a) % p
            self.mul = lambda a, b: a * b % p
            self.sqr = lambda a: a * a % p
            self.inv = lambda a: pow(a, -1, p)
            self.small = lambda a, k: a * k % p
            self.is_zero = lambda a: a % p == 0
        else:
            self.zero, self.one = (0, 0), (1, 0)
            self.add = lambda a, b: f2_add(a, b, p)
            self.sub = lambda a, b: f2_sub(a, b, p)
            self.neg = lambda a: f2_neg(a, p)
            self.mul = lambda a, b: f2_mul(a, b, p)
            self.sqr = lambda a: f2_sqr(a, p)
            self.inv = lambda a: f2_inv(a, p)
            self.small = lambda a, k: f2_scalar(a, k, p)
            self.is_zero = lambda a: a[0] % p == 0 and a[1] % p == 0


def self_check() -> None:
    """Numerical re-derivation of every constant above (run by tests/test_oracle_fields.py)."""
    for c in (BN254, BLS12_381):
        assert _is_probable_prime(c.r) and _is_probable_prime(c.q), c.name
        assert (c.r - 1) % (1 << c.two_adicity) == 0 and ((c.r - 1) >> c.two_adicity) % 2 == 1
        w = c.root_of_unity
        assert pow(w, 1 << c.two_adicity, c.r) == 1 and pow(w, 1 << (c.two_adicity - 1), c.r) == c.r - 1
        # fr_generator is a generator: g^((r-1)/f) != 1 for the small prime factors we can see
        for f in (2, 3, 5, 7, 11, 13):
            if (c.r - 1) % f == 0:
                assert pow(c.fr_generator, (c.r - 1) // f, c.r) != 1
        x, y = c.g1_gen
        assert (y * y - x * x * x - c.g1_b) % c.q == 0
        X, Y = c.g2_gen
        lhs = f2_sqr(Y, c.q)
        rhs = f2_add(f2_mul(f2_sqr(X, c.q), X, c.q), c.g2_b, c.q)
        assert lhs == rhs, c.name
        # twist coefficient is b/xi (D-type) or b*xi (M-type)
        if c.twist_is_d:
            assert f2_mul(c.g2_b, c.xi, c.q) == (c.g1_b, 0)
        else:
            assert f2_scalar(c.xi, c.g1_b, c.q) == c.g2_b
        assert (c.q * mont_inv64(c.q) + 1) % (1 << 64) == 0
    assert BN254.root_of_unity == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert BLS12_381.root_of_unity == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    # BN parametrisation: q = 36u^4+36u^3+24u^2+6u+1, r = 36u^4+36u^3+18u^2+6u+1
    u = BN254.x_param
    assert BN254.q == 36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1
    assert BN254.r == 36 * u**4 + 36 * u**3 + 18 * u**2 + 6 * u + 1
    # BLS12 parametrisation with x = -x_param: r = x^4 - x^2 + 1, q = (x-1)^2 r / 3 + x
    x = -BLS12_381.x_param
    assert BLS12_381.r == x**4 - x**2 + 1
    assert BLS12_381.q == (x - 1) ** 2 * BLS12_381.r // 3 + x


if __name__ == "__main__":
    self_check()
    print("fields.py self_check OK")
