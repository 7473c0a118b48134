// Unsaturated-limb field arithmetic for the bucket-accumulation kernel (the hot loop of the whole prover).
//
// field_dev.hpp multiplies on saturated 32-bit limbs: every 32x32 partial product is a v_mad_u64_u32 PLUS a v_addc_co_u32
// that banks the carry, and every column of the product scan costs two v_mov.  Here a field element is L limbs of B bits
// (9 x 29 for the 254-bit BN254 fields, 14 x 28 for the 381-bit BLS12-381 base field): a whole column (<= 2L partial
// products < 2^(2B)) fits the 64-bit accumulator, so a partial product is ONE instruction and the column hand-over is a
// shift — (L/N)^2 more products, but 0.6x the VALU instructions (measured: tools/ubench/unsat.hip, profiles/).
//
// Montgomery radix R' = 2^(L*B) (2^261 / 2^392), so p/R' <= 2^-7: a product of inputs < X*p and < Y*p is
// < (X*Y*p/R' + 1) * p < 2p for X*Y <= 128 WITHOUT a final subtraction, and additions / subtractions can stay
// unreduced ("value < K*p" bookkeeping is done by hand in xyzz_madd_u below).  A saturated Montgomery value
// X = x*2^(32N) (what the window tables and buckets hold in HBM) enters this form by a pure bit shift:
// (X << (L*B - 32N)) = x * R' (mod p), < 32p (< 256p); it leaves by one product with 2^(32N) mod p.
#pragma once
#include "ec_dev.hpp"
#include "field_dev.hpp"

namespace zkp {

template <class P>
struct UnsatCfg;
template <>
struct UnsatCfg<Bn254Fq> {
  static constexpr int L = 9, B = 29;
  static constexpr int MULCAP = 128;                        // floor(R' / p) = 169: products of (KA p) x (KB p) with KA KB <= 128 stay < 2p
};
template <>
struct UnsatCfg<Bls381Fq> {
  static constexpr int L = 14, B = 28;
  static constexpr int MULCAP = 2500;                       // floor(R' / p) = 2519
};

template <class P>
struct Fu {
  static constexpr int L = UnsatCfg<P>::L, B = UnsatCfg<P>::B, N = P::N;
  static constexpr uint32_t MASK = (1u << B) - 1;
  static constexpr int SHIFT = L * B - 32 * N;              // R' / R
  static constexpr int pbits() {                            // bit length of the modulus
    int top = N - 1;
    while (top > 0 && P::MOD[top] == 0) top--;
    int b = 0;
    for (uint32_t w = P::MOD[top]; w; w >>= 1) b++;
    return 32 * top + b;
  }
  static constexpr int PBITS = pbits();
  uint32_t v[L];

  static constexpr int KMAX = 48;                           // multiples of p kept as limb tables
  struct Tab {
    uint32_t mp[KMAX + 1][L];                               // limbs of M * p (top limb unmasked)
    uint32_t one[L];                                        // R' mod p = canonical "1" of this representation
    uint32_t ninv;                                          // -p^-1 mod 2^B
  };
  static constexpr uint32_t limb_of(const uint32_t* w, int nw, int i) {
    int bit = i * B, wi = bit >> 5, o = bit & 31;
    uint64_t lo = wi < nw ? w[wi] : 0, hi = wi + 1 < nw ? w[wi + 1] : 0;
    uint64_t x = (lo | (hi << 32)) >> o;
    return i == L - 1 ? (uint32_t)x : (uint32_t)(x & MASK);
  }
  static constexpr Tab make_tab() {
    Tab t{};
    for (int M = 0; M <= KMAX; M++) {
      uint32_t w[N + 2] = {};
      uint64_t c = 0;
      for (int k = 0; k < N; k++) {
        c += (uint64_t)P::MOD[k] * (uint32_t)M;
        w[k] = (uint32_t)c;
        c >>= 32;
      }
      w[N] = (uint32_t)c;
      for (int i = 0; i < L; i++) t.mp[M][i] = limb_of(w, N + 2, i);
    }
    // one = (2^(32N) mod p) * 2^SHIFT mod p, by SHIFT modular doublings
    uint32_t w[N + 2] = {};
    for (int k = 0; k < N; k++) w[k] = P::ONE[k];
    for (int s = 0; s < SHIFT; s++) {
      uint32_t carry = 0;
      for (int k = 0; k <= N; k++) {
        uint32_t nx = w[k] >> 31;
        w[k] = (w[k] << 1) | carry;
        carry = nx;
      }
      bool ge = true;                                       // w >= p ?
      for (int k = N; k >= 0; k--) {
        uint32_t pk = k < N ? P::MOD[k] : 0;
        if (w[k] != pk) {
          ge = w[k] > pk;
          break;
        }
      }
      if (ge) {
        uint64_t borrow = 0;
        for (int k = 0; k <= N; k++) {
          uint64_t pk = k < N ? P::MOD[k] : 0;
          uint64_t d = (uint64_t)w[k] - pk - borrow;
          w[k] = (uint32_t)d;
          borrow = (d >> 63) & 1;
        }
      }
    }
    for (int i = 0; i < L; i++) t.one[i] = limb_of(w, N + 2, i);
    uint32_t p0 = P::MOD[0], x = 1;
    for (int i = 0; i < 6; i++) x *= 2u - p0 * x;
    t.ninv = (0u - x) & MASK;
    return t;
  }
  static constexpr Tab TAB = make_tab();
  static constexpr uint32_t mp_limb(int M, int i) { return TAB.mp[M][i]; }
  static constexpr uint32_t ninv() { return TAB.ninv; }
  ZKP_DEV static Fu zero() {
    Fu r;
#pragma unroll
    for (int i = 0; i < L; i++) r.v[i] = 0;
    return r;
  }
  ZKP_DEV static Fu one() {
    Fu r;
#pragma unroll
    for (int i = 0; i < L; i++) r.v[i] = TAB.one[i];
    return r;
  }

  // saturated words (value < 2^(32N)) -> limbs, shifted left by `sh` bits
  ZKP_DEV static Fu from_words(const uint32_t* a, int sh) {
    Fu r;
#pragma unroll
    for (int i = 0; i < L; i++) {
      int bit = i * B - sh;
      int wi = bit >> 5, o = bit & 31;                      // arithmetic shift: wi = -1 for the bits shifted in
      uint32_t lo = (wi >= 0 && wi < N) ? a[wi] : 0, hi = (wi + 1 >= 0 && wi + 1 < N) ? a[wi + 1] : 0;
      uint64_t w = ((uint64_t)hi << 32) | lo;
      r.v[i] = (uint32_t)(w >> o) & MASK;
    }
    return r;
  }
  // x * 2^(32N) (saturated Montgomery, canonical) -> x * R' as the integer 2^SHIFT * X (< 2^SHIFT * p): usable as ONE
  // factor of a product whose other factor is < 4p
  ZKP_DEV static Fu from_sat(const Fp<P>& a) { return from_words(a.v, SHIFT); }
  // the same, brought below 2p by a product with one()
  ZKP_DEV static Fu from_sat_reduced(const Fp<P>& a) { return mul(from_sat(a), one()); }
  // normalised limbs, value < 2^(32N) -> saturated words
  ZKP_DEV void to_words(uint32_t* out) const {
#pragma unroll
    for (int w = 0; w < N; w++) {
      uint64_t acc = 0;
#pragma unroll
      for (int i = 0; i < L; i++) {
        int sh = i * B - 32 * w;
        if (sh > -B && sh < 32) acc |= sh >= 0 ? ((uint64_t)v[i] << sh) : ((uint64_t)v[i] >> (-sh));
      }
      out[w] = (uint32_t)acc;
    }
  }
  // x * R' (any value < 128p) -> canonical saturated Montgomery x * 2^(32N)
  ZKP_DEV Fp<P> to_sat() const {
    Fp<P> one = Fp<P>::one();                               // 2^(32N) mod p
    Fu t = mul(*this, from_words(one.v, 0));                // * 2^(32N) / R', < 2p
    Fp<P> r;
    t.to_words(r.v);
    return Fp<P>::reduce_once(r);
  }

  // Montgomery product, one 64-bit accumulator per column, no carry bank.  Limbs of a, b < 2^30.
  ZKP_DEV static Fu mul(const Fu& a, const Fu& b) {
    uint32_t m[L];
    Fu r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = 0; i < L; i++) {
        int j = k - i;
        if (j >= 0 && j < L) acc += (uint64_t)a.v[i] * b.v[j];
      }
#pragma unroll
      for (int i = 0; i < L; i++) {
        int j = k - i;
        if (j >= 0 && j < L && i < k) acc += (uint64_t)m[i] * mp_limb(1, j);
      }
      if (k < L) {
        m[k] = ((uint32_t)acc * ninv()) & MASK;
        acc += (uint64_t)m[k] * mp_limb(1, 0);
      } else {
        r.v[k - L] = (uint32_t)acc & MASK;
      }
      acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
  }
  // a*b + c*d with ONE Montgomery reduction (lazy reduction of a sum of two products): 2 L^2 + L^2 instead of 4 L^2 partial
  // products.  A column holds <= 2L products < 2^(2B) plus <= L reduction products: 27 * 2^58 < 2^63 for L = 9, B = 29
  // (BLS12-381: 42 * 2^56 < 2^62).  Value: (a*b + c*d) / R' + p < 2p when KA*KB + KC*KD <= MULCAP.
  ZKP_DEV static Fu mul_add(const Fu& a, const Fu& b, const Fu& c, const Fu& d) {
    uint32_t m[L];
    Fu r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = 0; i < L; i++) {
        int j = k - i;
        if (j >= 0 && j < L) {
          acc += (uint64_t)a.v[i] * b.v[j];
          acc += (uint64_t)c.v[i] * d.v[j];
        }
      }
#pragma unroll
      for (int i = 0; i < L; i++) {
        int j = k - i;
        if (j >= 0 && j < L && i < k) acc += (uint64_t)m[i] * mp_limb(1, j);
      }
      if (k < L) {
        m[k] = ((uint32_t)acc * ninv()) & MASK;
        acc += (uint64_t)m[k] * mp_limb(1, 0);
      } else {
        r.v[k - L] = (uint32_t)acc & MASK;
      }
      acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
  }
  // a*b + c*d + e*f + g*h with ONE reduction (an Fq2 combination such as R (Q - X3) - Y1 PPP): a column holds <= 4L products
  // < 2^(2B) plus <= L reduction products: 45 * 2^58 < 2^64 for L = 9, B = 29.  Value < 2p when the four K-products sum
  // to <= MULCAP.  (Only instantiated for the 254-bit fields.)
  ZKP_DEV static Fu mul_add4(const Fu& a, const Fu& b, const Fu& c, const Fu& d, const Fu& e, const Fu& f, const Fu& g,
                             const Fu& h) {
    static_assert(5 * L * ((uint64_t)1 << (2 * B - 32)) < ((uint64_t)1 << 32), "column accumulator would overflow");
    uint32_t m[L];
    Fu r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = 0; i < L; i++) {
        int j = k - i;
        if (j >= 0 && j < L) {
          acc += (uint64_t)a.v[i] * b.v[j];
          acc += (uint64_t)c.v[i] * d.v[j];
          acc += (uint64_t)e.v[i] * f.v[j];
          acc += (uint64_t)g.v[i] * h.v[j];
        }
      }
#pragma unroll
      for (int i = 0; i < L; i++) {
        int j = k - i;
        if (j >= 0 && j < L && i < k) acc += (uint64_t)m[i] * mp_limb(1, j);
      }
      if (k < L) {
        m[k] = ((uint32_t)acc * ninv()) & MASK;
        acc += (uint64_t)m[k] * mp_limb(1, 0);
      } else {
        r.v[k - L] = (uint32_t)acc & MASK;
      }
      acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
  }
  // a >= M*p ? a - M*p : a   (normalised limbs in, normalised limbs out): halves a value bound without a product
  template <int M>
  ZKP_DEV static Fu csub(const Fu& a) {
    Fu r;
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      int32_t t = (int32_t)a.v[i] - (int32_t)mp_limb(M, i) + carry;
      if (i < L - 1) {
        r.v[i] = (uint32_t)t & MASK;
        carry = t >> B;
      } else {
        r.v[i] = (uint32_t)t;
      }
    }
    const bool neg = (int32_t)r.v[L - 1] < 0;
#pragma unroll
    for (int i = 0; i < L; i++) r.v[i] = neg ? a.v[i] : r.v[i];
    return r;
  }
  // (a dedicated squaring — cross terms once against a doubled operand, 45 instead of 81 products — measured no
  //  difference in the accumulate kernel and was dropped)
  ZKP_DEV Fu sqr() const { return mul(*this, *this); }

  // a - b + M*p with normalised limbs; requires b < M*p (value) — result in (0, a + M*p)
  template <int M>
  ZKP_DEV static Fu sub(const Fu& a, const Fu& b) {
    Fu r;
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      int32_t t = (int32_t)a.v[i] + (int32_t)(mp_limb(M, i) - b.v[i]) + carry;
      if (i < L - 1) {
        r.v[i] = (uint32_t)t & MASK;
        carry = t >> B;
      } else {
        r.v[i] = (uint32_t)t;
      }
    }
    return r;
  }
  // a + b, normalised
  ZKP_DEV static Fu add(const Fu& a, const Fu& b) {
    Fu r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      uint32_t t = a.v[i] + b.v[i] + carry;
      if (i < L - 1) {
        r.v[i] = t & MASK;
        carry = t >> B;
      } else {
        r.v[i] = t;
      }
    }
    return r;
  }
  // 2a, normalised
  ZKP_DEV Fu dbl() const {
    Fu r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      uint32_t t = (v[i] << 1) + carry;
      if (i < L - 1) {
        r.v[i] = t & MASK;
        carry = t >> B;
      } else {
        r.v[i] = t;
      }
    }
    return r;
  }
  // cheap necessary condition for "value is a multiple of p" when 0 < value < KMAX*p: the two low limbs match k*p
  // (2^-58 false alarms per candidate: a false alarm costs a whole task on the slow exact path)
  template <int KMAX>
  ZKP_DEV bool maybe_multiple_of_p() const {
    bool hit = false;
#pragma unroll
    for (int k = 1; k < KMAX; k++) hit |= (v[0] == (mp_limb(k, 0) & MASK)) & (v[1] == (mp_limb(k, 1) & MASK));
    return hit;
  }
  // exact: normalised limbs are a unique representation, so "value in {p, 2p, .., (KMAX-1)p}" is a limb-wise comparison
  template <int KMAX>
  ZKP_DEV bool is_multiple_of_p() const {
    bool hit = false;
#pragma unroll
    for (int k = 1; k < KMAX; k++) {
      bool eq = true;
#pragma unroll
      for (int i = 0; i < L; i++) eq &= v[i] == mp_limb(k, i);
      hit |= eq;
    }
    return hit;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Value bounds as types.  UB<P, K> wraps an element whose VALUE is known to be < K * p (limbs normalised); every operation
// derives the bound of its result and static_asserts its precondition, so the "lazy reduction" bookkeeping of the mixed
// addition below is checked by the compiler instead of by comments:
//   product:   needs KA * KB <= MULCAP (p / R' <= 2^-7: the result is < (KA*KB*p/R' + 1) p < 2p)          -> K = 2
//   a - b:     computed as a + (KB*p - b), needs the table of multiples to reach KB                        -> K = KA + KB
//   2a:                                                                                                     -> K = 2 KA
//   capacity:  K * p must fit L*B bits with room for one more addition (K <= CAPK)
template <class P, int K>
struct UB {
  Fu<P> f;
};
template <class P>
struct UBLimits {
  // p / R' <= 2^-(L*B - bits(p)):  BN254 2^-7, BLS12-381 2^-11
  static constexpr int MULCAP = UnsatCfg<P>::MULCAP;
  static_assert(MULCAP < (1 << (Fu<P>::L * Fu<P>::B - Fu<P>::PBITS + 1)), "MULCAP must be below R' / p");
  static constexpr int CAPK = MULCAP;                      // K*p < 2^(L*B) with the same margin
};
template <class P, int KA, int KB>
ZKP_DEV UB<P, 2> ub_mul(const UB<P, KA>& a, const UB<P, KB>& b) {
  static_assert((long)KA * KB <= UBLimits<P>::MULCAP, "product of unreduced operands exceeds the Montgomery slack");
  return {Fu<P>::mul(a.f, b.f)};
}
// a*b + c*d, one reduction
template <class P, int KA, int KB, int KC, int KD>
ZKP_DEV UB<P, 2> ub_mul_add(const UB<P, KA>& a, const UB<P, KB>& b, const UB<P, KC>& c, const UB<P, KD>& d) {
  static_assert((long)KA * KB + (long)KC * KD <= UBLimits<P>::MULCAP, "sum of products exceeds the Montgomery slack");
  return {Fu<P>::mul_add(a.f, b.f, c.f, d.f)};
}
// K*p - a  (a < K*p), normalised limbs: the negation that turns a difference of products into a sum
template <int K, class P, int KA>
ZKP_DEV UB<P, K> ub_neg(const UB<P, KA>& a) {
  static_assert(KA <= K && K <= Fu<P>::KMAX, "no table entry for this multiple of p");
  Fu<P> kp;
#pragma unroll
  for (int i = 0; i < Fu<P>::L; i++) kp.v[i] = Fu<P>::mp_limb(0, i);    // zero
  return {Fu<P>::template sub<K>(kp, a.f)};
}
template <class P, int KA, int KB>
ZKP_DEV UB<P, KA + KB> ub_sub(const UB<P, KA>& a, const UB<P, KB>& b) {
  static_assert(KB <= Fu<P>::KMAX, "no table entry for this multiple of p");
  static_assert(KA + KB <= UBLimits<P>::CAPK, "value would outgrow the limbs");
  return {Fu<P>::template sub<KB>(a.f, b.f)};
}
template <class P, int KA, int KB>
ZKP_DEV UB<P, KA + KB> ub_add(const UB<P, KA>& a, const UB<P, KB>& b) {
  static_assert(KA + KB <= UBLimits<P>::CAPK, "value would outgrow the limbs");
  return {Fu<P>::add(a.f, b.f)};
}
template <class P, int KA>
ZKP_DEV UB<P, 2 * KA> ub_dbl(const UB<P, KA>& a) {
  static_assert(2 * KA <= UBLimits<P>::CAPK, "value would outgrow the limbs");
  return {a.f.dbl()};
}

// Bucket accumulator in unsaturated form; the stored bounds are part of the type.
template <class P>
struct XYZZu {
  UB<P, 8> x;
  UB<P, 4> y;
  UB<P, 2> zz, zzz;
  bool inf;
};

// acc += P where (ux, uy) = from_sat of a gathered affine point (not the identity): the integers 2^SHIFT * X, i.e.
// values < 2^SHIFT * p.  madd-2008-s (8M + 2S).  Returns false WITHOUT touching acc when P may equal +-acc (the
// difference of the x coordinates matches a multiple of p in its two low limbs): the caller hands the bucket to the exact
// saturated path.
template <class P>
ZKP_DEV bool xyzz_madd_u(XYZZu<P>& acc, const Fu<P>& ux_, const Fu<P>& uy_) {
  using U = Fu<P>;
  constexpr int KIN = 1 << U::SHIFT;                       // 32 (BN254) / 256 (BLS12-381)
  const UB<P, KIN> ux{ux_}, uy{uy_};
  const UB<P, 1> one{U::one()};
  if (acc.inf) {
    acc.x = {ub_mul(ux, one).f};                           // < 2p, stored as "< 8p"
    acc.y = {ub_mul(uy, one).f};
    acc.zz = {one.f};
    acc.zzz = {one.f};
    acc.inf = false;
    return true;
  }
  const auto u2 = ub_mul(ux, acc.zz);                      // < 2p
  const auto s2 = ub_mul(uy, acc.zzz);                     // < 2p
  const auto pd = ub_sub(u2, acc.x);                       // (0, 10p)
  const auto rd = ub_sub(s2, acc.y);                       // (0, 6p)
  if (pd.f.template maybe_multiple_of_p<10>()) return false;
  const auto pp = ub_mul(pd, pd);                          // < 2p
  const auto ppp = ub_mul(pd, pp);
  const auto q = ub_mul(acc.x, pp);
  const auto t = ub_sub(ub_mul(rd, rd), ppp);              // (0, 4p)
  const UB<P, 8> x3 = ub_sub(t, ub_dbl(q));                // (0, 8p): the type of acc.x
#ifdef ZKP_UNSAT_NO_LAZY_Y3
  const UB<P, 4> y3 = ub_sub(ub_mul(rd, ub_sub(q, x3)), ub_mul(acc.y, ppp));   // (0, 4p): the type of acc.y
#else
  // y3 = R (Q - X3) - Y1 PPP as ONE lazily reduced sum of two products: R (Q - X3) + Y1 (2p - PPP); bounds 6*10 + 4*2 = 68
  const UB<P, 4> y3 = {ub_mul_add(rd, ub_sub(q, x3), acc.y, ub_neg<2>(ppp)).f};   // < 2p, stored as "< 4p"
#endif
  acc.zz = ub_mul(acc.zz, pp);
  acc.zzz = ub_mul(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2 + 1) on the bounded types, for the BN254 G2 accumulator (round 2).  Products are SCHOOLBOOK with one
// lazily reduced sum of two products per component (c0 = a0 b0 + a1 (Kp - b1), c1 = a0 b1 + a1 b0: 2 * (2 * 81 + 81) = 486
// v_mad_u64_u32 — the same multiplier work as three separate products, but no additions of unreduced operands, so the
// 2^-7 slack of the 254-bit field suffices: KA*KB + KA*KB <= 128 holds for every product of the mixed addition once x3 is
// brought below 4p by one conditional subtraction).
template <class P, int K>
struct UB2 {
  UB<P, K> c0, c1;
};
template <class P, int KA, int KB>
ZKP_DEV UB2<P, 2> ub2_mul(const UB2<P, KA>& a, const UB2<P, KB>& b) {
  return {ub_mul_add(a.c0, b.c0, a.c1, ub_neg<KB>(b.c1)), ub_mul_add(a.c0, b.c1, a.c1, b.c0)};
}
template <class P, int KA>
ZKP_DEV UB2<P, 4> ub2_sqr(const UB2<P, KA>& a) {
  const UB<P, 2> c0 = ub_mul_add(a.c0, a.c0, a.c1, ub_neg<KA>(a.c1));
  const UB<P, 4> c1 = ub_dbl(ub_mul(a.c0, a.c1));
  return {UB<P, 4>{c0.f}, c1};
}
template <class P, int KA, int KB>
ZKP_DEV UB2<P, KA + KB> ub2_sub(const UB2<P, KA>& a, const UB2<P, KB>& b) {
  return {ub_sub(a.c0, b.c0), ub_sub(a.c1, b.c1)};
}
template <class P, int KA>
ZKP_DEV UB2<P, 2 * KA> ub2_dbl(const UB2<P, KA>& a) {
  return {ub_dbl(a.c0), ub_dbl(a.c1)};
}

template <class P>
struct XYZZu2 {
  UB2<P, 4> x;
  UB2<P, 2> y, zz, zzz;
  bool inf;
};

// acc += P for G2 (Fq2 coordinates); (ux, uy) = from_sat of the gathered affine point's components.  Same contract as
// xyzz_madd_u: returns false WITHOUT touching acc when P may equal +-acc.
template <class P>
ZKP_DEV bool xyzz_madd_u2(XYZZu2<P>& acc, const Fu<P>& ux0, const Fu<P>& ux1, const Fu<P>& uy0, const Fu<P>& uy1) {
  using U = Fu<P>;
  constexpr int KIN = 1 << U::SHIFT;                       // 32: the integers 2^SHIFT * X, < 32p
  static_assert(2 * KIN * 2 <= UBLimits<P>::MULCAP, "table operand times a reduced Fq2 element exceeds the slack");
  const UB2<P, KIN> ux{{ux0}, {ux1}}, uy{{uy0}, {uy1}};
  const UB<P, 1> one{U::one()};
  if (acc.inf) {
    acc.x = {UB<P, 4>{ub_mul(ux.c0, one).f}, UB<P, 4>{ub_mul(ux.c1, one).f}};
    acc.y = {ub_mul(uy.c0, one), ub_mul(uy.c1, one)};
    acc.zz = {UB<P, 2>{one.f}, UB<P, 2>{U::zero()}};
    acc.zzz = acc.zz;
    acc.inf = false;
    return true;
  }
  const auto u2 = ub2_mul(ux, acc.zz);                      // < 2p          (32*2 + 32*2 = 128)
  const auto s2 = ub2_mul(uy, acc.zzz);
  const auto pd = ub2_sub(u2, acc.x);                       // (0, 6p)
  const auto rd = ub2_sub(s2, acc.y);                       // (0, 4p)
  if (pd.c0.f.template maybe_multiple_of_p<6>() && pd.c1.f.template maybe_multiple_of_p<6>()) return false;
  const UB2<P, 2> pp = [&] {                                // pd^2: 36 + 36 = 72
    const UB<P, 2> c0 = ub_mul_add(pd.c0, pd.c0, pd.c1, ub_neg<6>(pd.c1));
    const UB<P, 2> c1 = ub_mul_add(pd.c0, pd.c1, pd.c0, pd.c1);        // 2 pd0 pd1 as one lazily reduced sum: < 2p
    return UB2<P, 2>{c0, c1};
  }();
  const auto ppp = ub2_mul(pd, pp);                         // 6*2 + 6*2 = 24
  const auto q = ub2_mul(acc.x, pp);                        // 4*2 + 4*2 = 16
  const UB2<P, 2> rr = [&] {                                // rd^2: 16 + 16 = 32
    const UB<P, 2> c0 = ub_mul_add(rd.c0, rd.c0, rd.c1, ub_neg<4>(rd.c1));
    const UB<P, 2> c1 = ub_mul_add(rd.c0, rd.c1, rd.c0, rd.c1);
    return UB2<P, 2>{c0, c1};
  }();
  const auto x3w = ub2_sub(ub2_sub(rr, ppp), ub2_dbl(q));   // (0, 8p)
  const UB2<P, 4> x3 = {UB<P, 4>{U::template csub<4>(x3w.c0.f)}, UB<P, 4>{U::template csub<4>(x3w.c1.f)}};   // < 4p
  const auto t = ub2_sub(q, x3);                            // (0, 6p)
  // y3 = rd * t - y * ppp, each component ONE lazily reduced sum of four products: 4*6 + 4*6 + 2*2 + 2*2 = 56
  static_assert(4 * 6 * 2 + 2 * 2 * 2 <= UBLimits<P>::MULCAP, "y3 exceeds the slack");
  const UB<P, 2> y30{U::mul_add4(rd.c0.f, t.c0.f, rd.c1.f, ub_neg<6>(t.c1).f, acc.y.c0.f, ub_neg<2>(ppp.c0).f, acc.y.c1.f, ppp.c1.f)};
  const UB<P, 2> y31{U::mul_add4(rd.c0.f, t.c1.f, rd.c1.f, t.c0.f, acc.y.c0.f, ub_neg<2>(ppp.c1).f, acc.y.c1.f, ub_neg<2>(ppp.c0).f)};
  acc.zz = ub2_mul(acc.zz, pp);                             // 2*2 + 2*2 = 8
  acc.zzz = ub2_mul(acc.zzz, ppp);
  acc.x = x3;
  acc.y = {y30, y31};
  return true;
}

// (An Fq2 layer on these types — Karatsuba products <4p, 6p>, complex squarings <2p, 4p>, accumulator bounds x <14p, 22p>,
//  y <8p, 12p> closed without reductions, window table stored as x * R' — was built for BLS12-381 G2, where p / R' = 2^-11
//  leaves room for the Karatsuba sums.  Bit-exact, but 391 VGPRs and only 5 % faster than the saturated G2 kernel (24.3 vs
//  25.5 ms for the 2^22 B2 MSM) and slower end to end (15.9 vs 16.4 proofs/s): removed, see DESIGN.md.)

}  // namespace zkp
