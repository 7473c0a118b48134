// Prime-field and quadratic-extension arithmetic for gfx950 (CDNA4), 32-bit limbs, Montgomery form.
//
// Replaces, on the device, the ark-ff 0.2 `Fp256`/`Fp384` arithmetic the reference reaches through
// ark-ec / ark-poly (call sites: /root/reference/groth16/src/prover.rs:187,190,220 and
// /root/reference/groth16/src/r1cs_to_qap.rs:144-169).  Wire format is identical to ark's
// `BigInteger256/384`: little-endian limbs of a*R mod p with R = 2^256 / 2^384 — a 4x u64 limb array and
// an 8x u32 limb array are the same bytes on a little-endian machine, so no conversion is needed.
//
// CDNA4 has no 64x64 multiplier: the widest integer multiply-add is v_mad_u64_u32 (32x32+64 -> 64), so
// the limb width is 32 bits and a 256-bit Montgomery product is 8x8 + 8x8 v_mad_u64_u32.  All moduli on
// this path have at least one spare top bit (254/255/381 bits), so the CIOS accumulator needs only N+1
// words ("no-carry" variant) and one conditional subtraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zkp {

#include "field_constants.inc"

#define ZKP_DEV __device__ __forceinline__
#define ZKP_HD __host__ __device__ __forceinline__
// Montgomery multiplication is ~600 VALU instructions (8 limbs) / ~1300 (12 limbs).  Fully inlining it into
// every EC formula makes G2 kernels hundreds of thousands of instructions long (tens of minutes of hipcc).
// Policy: translation units that hold a hot loop define ZKP_INLINE_MUL; everything else calls one
// out-of-line copy per field (operands by value -> passed in VGPRs, no scratch).
#ifdef ZKP_INLINE_MUL
#define ZKP_MUL_ATTR __device__ __forceinline__
#else
#define ZKP_MUL_ATTR __device__ __noinline__
#endif

template <class P>
struct Fp;
template <class P>
ZKP_MUL_ATTR Fp<P> fp_mul(Fp<P> a, Fp<P> b);
#include "addsub_gen.inc"

template <class P>
struct Fp {
  static constexpr int N = P::N;
  uint32_t v[N];

  ZKP_DEV static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
  }
  ZKP_DEV static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::ONE[i];
    return r;
  }
  ZKP_DEV static Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::R2[i];
    return r;
  }
  ZKP_DEV bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i];
    return o == 0;
  }
  ZKP_DEV bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
  ZKP_DEV bool operator!=(const Fp& b) const { return !(*this == b); }

  // modular add / sub / final reduction: 3 VALU instructions per limb as carry chains (tools/gen_addsub.py)
  ZKP_DEV static void mod_limbs(uint32_t* m) {
#pragma unroll
    for (int i = 0; i < N; i++) m[i] = P::MOD[i];
  }
  // r = a - p if a >= p else a   (input < 2p)
  ZKP_DEV static Fp reduce_once(const Fp& a) {
    Fp r = a;
    uint32_t m[N];
    mod_limbs(m);
    ReduceAsm<N>::run(r.v, m);
    return r;
  }
  // spare top bit => no carry out of the top word for a, b < p
  ZKP_DEV friend Fp operator+(const Fp& a, const Fp& b) {
    Fp r;
    uint32_t m[N];
    mod_limbs(m);
    AddModAsm<N>::run(r.v, a.v, b.v, m);
    return r;
  }
  ZKP_DEV friend Fp operator-(const Fp& a, const Fp& b) {
    Fp r;
    uint32_t m[N];
    mod_limbs(m);
    SubModAsm<N>::run(r.v, a.v, b.v, m);
    return r;
  }
  ZKP_DEV Fp neg() const { return is_zero() ? *this : (zero() - *this); }
  ZKP_DEV Fp dbl() const { return *this + *this; }

  ZKP_DEV friend Fp operator*(const Fp& a, const Fp& b) { return fp_mul<P>(a, b); }
  ZKP_DEV Fp sqr() const { return (*this) * (*this); }

  // a^e for a public exponent given as N 32-bit limbs (square-and-multiply, MSB first)
  ZKP_DEV Fp pow_limbs(const uint32_t* e) const {
    Fp r = one();
    for (int i = N - 1; i >= 0; i--) {
      for (int b = 31; b >= 0; b--) {
        r = r.sqr();
        if ((e[i] >> b) & 1) r = r * (*this);
      }
    }
    return r;
  }
  ZKP_DEV Fp pow_u64(uint64_t e) const {
    Fp r = one();
    Fp base = *this;
    while (e) {
      if (e & 1) r = r * base;
      base = base.sqr();
      e >>= 1;
    }
    return r;
  }
  // Fermat inverse (0 -> 0)
  ZKP_DEV Fp inv() const {
    uint32_t e[N];
#pragma unroll
    for (int i = 0; i < N; i++) e[i] = P::PM2[i];
    return pow_limbs(e);
  }
  // Montgomery -> canonical (ark `into_repr()`): multiply by 1
  ZKP_DEV Fp from_mont() const {
    Fp o = zero();
    o.v[0] = 1;
    return (*this) * o;
  }
  // canonical -> Montgomery
  ZKP_DEV Fp to_mont() const { return (*this) * r2(); }

  // 16-byte vector global loads / stores (N*4 bytes, N % 4 == 0)
  ZKP_DEV static Fp load(const void* p) {
    Fp r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
      uint4 x = q[i];
      r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
    return r;
  }
  // the same with the non-temporal hint (streaming gathers that should not displace other kernels' lines in L2)
  ZKP_DEV static Fp load_nt(const void* p) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    Fp r;
    const u32x4_t* q = reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
      u32x4_t x = __builtin_nontemporal_load(q + i);
      r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
    return r;
  }
  ZKP_DEV void store(void* p) const {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < N / 4; i++) q[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  }
};

// Montgomery product a*b*R^-1 mod p.
// Product scanning (column by column) with the running column sum held as a 64-bit accumulator plus a 32-bit
// overflow counter; every partial product is ONE v_mad_u64_u32 (32x32 + 64 -> 64, carry to VCC) followed by ONE
// v_addc_co_u32 that banks the carry (one asm statement per run of products of a column, tools/gen_mac.py) —
// 2 VALU instructions per 32x32 product and two v_mov per column, versus
// ~4.5 per product for the C (CIOS) formulation, whose 64-bit zero-extensions hipcc materialises as v_mov pairs
// (measured on MI355X: 118-129 G products/s vs 86-95 G for 256-bit fields, 59 vs 44 G for the 384-bit field;
// tools/ubench/mulasm.hip checks the two formulations against each other).  The modulus limbs are scalar
// operands (SGPR / constant bus).  All moduli here have a spare top bit, so the result is < 2p and one
// conditional subtraction finishes the reduction.
#include "mac_gen.inc"

// column K of the product scan (compile-time K so that each run of partial products is ONE asm statement)
template <class P, int K>
struct MontColumn {
  static ZKP_DEV void run(uint64_t& acc, uint32_t& ovf, const Fp<P>& a, const Fp<P>& b, uint32_t* m, Fp<P>& r) {
    constexpr int N = P::N;
    constexpr int i0 = K < N ? 0 : K - N + 1;           // first row with a partner in column K
    constexpr int i1 = K < N ? K : N - 1;
    MacVV<i1 - i0 + 1>::run(acc, ovf, &a.v[i0], &b.v[K - i0]);
    constexpr int mi1 = (K - 1 < N - 1) ? K - 1 : N - 1;  // m[i] is known for i < K
    if constexpr (mi1 >= i0) MacVS<P, mi1 - i0 + 1, K - i0>::run(acc, ovf, &m[i0]);
    if constexpr (K < N) {
      m[K] = (uint32_t)acc * P::INV;
      MacVS<P, 1, 0>::run(acc, ovf, &m[K]);             // low word of the column becomes 0
    } else {
      r.v[K - N] = (uint32_t)acc;
    }
    acc = (acc >> 32) | ((uint64_t)ovf << 32);
    ovf = 0;
    if constexpr (K + 1 < 2 * N - 1) MontColumn<P, K + 1>::run(acc, ovf, a, b, m, r);
  }
};

template <class P>
ZKP_MUL_ATTR Fp<P> fp_mul(Fp<P> a, Fp<P> b) {
  constexpr int N = P::N;
  uint32_t m[N];
  Fp<P> r;
  uint64_t acc = 0;
  uint32_t ovf = 0;
  MontColumn<P, 0>::run(acc, ovf, a, b, m, r);
  r.v[N - 1] = (uint32_t)acc;
  return Fp<P>::reduce_once(r);
}

// Fq2 = Fq[u]/(u^2+1)  (both BN254 and BLS12-381 use the non-residue -1), ark layout (c0, c1).
template <class P>
struct Fp2 {
  using B = Fp<P>;
  static constexpr int N = 2 * P::N;
  B c0, c1;
  ZKP_DEV static Fp2 zero() { return {B::zero(), B::zero()}; }
  ZKP_DEV static Fp2 one() { return {B::one(), B::zero()}; }
  ZKP_DEV bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZKP_DEV bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
  ZKP_DEV bool operator!=(const Fp2& o) const { return !(*this == o); }
  ZKP_DEV friend Fp2 operator+(const Fp2& a, const Fp2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  ZKP_DEV friend Fp2 operator-(const Fp2& a, const Fp2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  ZKP_DEV Fp2 neg() const { return {c0.neg(), c1.neg()}; }
  ZKP_DEV Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // Karatsuba: 3 base multiplications
  ZKP_DEV friend Fp2 operator*(const Fp2& a, const Fp2& b) {
    B v0 = a.c0 * b.c0, v1 = a.c1 * b.c1;
    B s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {v0 - v1, s - v0 - v1};
  }
  // complex squaring: 2 base multiplications
  ZKP_DEV Fp2 sqr() const {
    B t = c0 * c1;
    return {(c0 + c1) * (c0 - c1), t.dbl()};
  }
  ZKP_DEV Fp2 inv() const {
    B n = (c0.sqr() + c1.sqr()).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  ZKP_DEV static Fp2 load(const void* p) {
    const char* q = reinterpret_cast<const char*>(p);
    return {B::load(q), B::load(q + 4 * P::N)};
  }
  ZKP_DEV static Fp2 load_nt(const void* p) {
    const char* q = reinterpret_cast<const char*>(p);
    return {B::load_nt(q), B::load_nt(q + 4 * P::N)};
  }
  ZKP_DEV void store(void* p) const {
    char* q = reinterpret_cast<char*>(p);
    c0.store(q);
    c1.store(q + 4 * P::N);
  }
};

}  // namespace zkp
