// Bucket points in the UNSATURATED layout, end to end (round 2): the bucket array, the per-task partial sums and every level
// of the reduction pyramid hold XYZZ points as L x 29/28-bit limbs with the value bounds of unsat_dev.hpp (x < 8p / 4p,
// y < 4p / 2p, zz, zzz < 2p) instead of canonical 32-bit-limb Montgomery values.  What this buys:
//   * the accumulate kernels store their VGPR accumulator as it is (no 4 / 8 conversion products per bucket);
//   * pair / segsum / combine / final kernels — 2 * 2^(c-1) full XYZZ additions per MSM, ~10 % of a proof's machine time —
//     run on the one-instruction-per-partial-product multiplier: add-2008-s costs 13.5 product-equivalents of 162
//     v_mad_u64_u32 (y3 is one lazily reduced sum of two products) instead of 14 saturated products of ~330 instructions.
// Only the final result of an MSM is converted back to the canonical saturated XYZZ / Jacobian form the rest of the
// library (proof assembly, C ABI) uses.
//
// The identity is all-zero limbs (zz == 0), so a zeroed bucket array is an array of identities.  The additions are
// complete: when the x-difference is a multiple of p (P = +-Q; a two-limb filter, then an exact limb-wise comparison — normalised
// limbs are unique) the y-difference decides between the doubling formulas and the identity, all in the unsaturated domain.
// (Round 2 first called the saturated formulas out of line for this: the callee's 214 / 264 VGPRs and 0.6 / 1.4 KB of scratch
// became the register budget of every pyramid kernel.)
//
// Replaces, like ec_dev.hpp, the `add_assign` / `double_in_place` calls inside ark-ec 0.2
// `VariableBaseMSM::multi_scalar_mul` (reference call sites: /root/reference/groth16/src/prover.rs:187,190,220).
#pragma once
#include "ec_dev.hpp"
#include "unsat_dev.hpp"

namespace zkp {

template <class P, int KA, int KB>
ZKP_DEV UB<P, KA + KB> ub_add3(const UB<P, KA>& a, const UB<P, KB>& b) {
  static_assert(KA + KB <= UBLimits<P>::CAPK, "value would outgrow the limbs");
  return {Fu<P>::add(a.f, b.f)};
}
template <class P>
ZKP_DEV bool fu_is_zero(const Fu<P>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < Fu<P>::L; i++) o |= a.v[i];
  return o == 0;
}
template <class P>
ZKP_DEV Fu<P> fu_load(const void* p) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
  Fu<P> r;
#pragma unroll
  for (int i = 0; i < Fu<P>::L; i++) r.v[i] = w[i];
  return r;
}
template <class P>
ZKP_DEV void fu_store(void* p, const Fu<P>& a) {
  uint32_t* w = reinterpret_cast<uint32_t*>(p);
#pragma unroll
  for (int i = 0; i < Fu<P>::L; i++) w[i] = a.v[i];
}

template <class F>
struct BkPoint;
template <class F>
__device__ __noinline__ void bk_same_x_mem(const char* a, char* out, bool same_y);
// P = +-Q inside an addition (same x): doubling or the identity.  Out of line — one copy of the doubling formulas per
// translation unit instead of one per call site of add() (the BLS12-381 G2 unit took 5 minutes to compile with it inline);
// the callee needs fewer registers than add() itself, so it does not raise the callers' register budget.
template <class F>
__device__ __noinline__ void bk_same_x(BkPoint<F>& a, bool same_y);

// ------------------------------------------------------------------------------------------------ G1
template <class P>
struct BkPoint<Fp<P>> {
  using F = Fp<P>;
  using U = Fu<P>;
  XYZZu<P> v;
  static constexpr int EB = 4 * U::L;                      // bytes of one element
  static constexpr int BYTES = 4 * EB;
  ZKP_DEV bool is_inf() const { return v.inf; }
  ZKP_DEV static BkPoint inf() {
    BkPoint r;
    r.v.x = {U::zero()};
    r.v.y = {U::zero()};
    r.v.zz = {U::zero()};
    r.v.zzz = {U::zero()};
    r.v.inf = true;
    return r;
  }
  ZKP_DEV static BkPoint load(const void* p) {
    const char* q = reinterpret_cast<const char*>(p);
    BkPoint r;
    r.v.zz = {fu_load<P>(q + 2 * EB)};
    r.v.inf = fu_is_zero(r.v.zz.f);
    r.v.x = {fu_load<P>(q)};
    r.v.y = {fu_load<P>(q + EB)};
    r.v.zzz = {fu_load<P>(q + 3 * EB)};
    return r;
  }
  ZKP_DEV void store(void* p) const {
    char* q = reinterpret_cast<char*>(p);
    const BkPoint z = inf();
    const XYZZu<P>& s = v.inf ? z.v : v;
    fu_store<P>(q, s.x.f);
    fu_store<P>(q + EB, s.y.f);
    fu_store<P>(q + 2 * EB, s.zz.f);
    fu_store<P>(q + 3 * EB, s.zzz.f);
  }
  ZKP_DEV static BkPoint from_sat(const XYZZ<F>& s) {
    if (s.is_inf()) return inf();
    BkPoint r;
    r.v.x = {U::from_sat_reduced(s.x)};                    // < 2p
    r.v.y = {U::from_sat_reduced(s.y)};
    r.v.zz = {U::from_sat_reduced(s.zz)};
    r.v.zzz = {U::from_sat_reduced(s.zzz)};
    r.v.inf = false;
    return r;
  }
  ZKP_DEV XYZZ<F> to_sat() const {
    if (v.inf) return XYZZ<F>::inf();
    return {v.x.f.to_sat(), v.y.f.to_sat(), v.zz.f.to_sat(), v.zzz.f.to_sat()};
  }
#if defined(ZKP_G1_REG_ADD)
  static constexpr bool MEM_ADD = false;                   // round-2 form: both operands in registers (115-120 VGPRs + 168 B scratch)
  ZKP_DEV static void dbl_mem(const char* a, char* out) { load(a).dbl().store(out); }
  ZKP_DEV static void add_mem(const char* a, const char* b, char* out) {
    BkPoint pa = load(a);
    pa.add(load(b));
    pa.store(out);
  }
#else
  // streamed operands like the G2 form below (see BkPoint<Fp2>::add_mem for the scheme)
  static constexpr bool MEM_ADD = true;
  ZKP_DEV static void copy_point(char* out, const char* in) {
    if (out == in) return;
#pragma unroll
    for (int k = 0; k < 4; k++) fu_store<P>(out + k * EB, fu_load<P>(in + k * EB));
  }
  ZKP_DEV static void dbl_mem(const char* a, char* out) {   // exceptional path only: through the out-of-line doubling (104 VGPRs)
    BkPoint p = load(a);
    bk_same_x<F>(p, true);
    p.store(out);
  }
  ZKP_DEV static void add_mem(const char* a, const char* b, char* out) {
    const UB<P, 2> zz1{fu_load<P>(a + 2 * EB)}, zz2{fu_load<P>(b + 2 * EB)};
    if (fu_is_zero(zz2.f)) {
      copy_point(out, a);
      return;
    }
    if (fu_is_zero(zz1.f)) {
      copy_point(out, b);
      return;
    }
    const UB<P, 8> x1{fu_load<P>(a)};
    const auto u1 = ub_mul(x1, zz2);
    const UB<P, 8> x2{fu_load<P>(b)};
    const auto pd = ub_sub(ub_mul(x2, zz1), u1);           // (0, 4p)
    const bool same_x = pd.f.template maybe_multiple_of_p<4>() && pd.f.template is_multiple_of_p<4>();
    if (!same_x) fu_store<P>(out + 2 * EB, ub_mul(zz1, zz2).f);     // zz1 zz2 parked in out.zz
    __asm__ volatile("" ::: "memory");
    const UB<P, 2> zzz1{fu_load<P>(a + 3 * EB)}, zzz2{fu_load<P>(b + 3 * EB)};
    const UB<P, 4> y1{fu_load<P>(a + EB)};
    const auto s1 = ub_mul(y1, zzz2);
    const UB<P, 4> y2{fu_load<P>(b + EB)};
    const auto rd = ub_sub(ub_mul(y2, zzz1), s1);          // (0, 4p)
    if (same_x) {
      bk_same_x_mem<F>(a, out, rd.f.template is_multiple_of_p<4>());
      return;
    }
    fu_store<P>(out + 3 * EB, ub_mul(zzz1, zzz2).f);
    __asm__ volatile("" ::: "memory");
    const auto pp = ub_sqr(pd);
    const auto ppp = ub_mul(pd, pp);
    fu_store<P>(out + 2 * EB, ub_mul(UB<P, 2>{fu_load<P>(out + 2 * EB)}, pp).f);
    fu_store<P>(out + 3 * EB, ub_mul(UB<P, 2>{fu_load<P>(out + 3 * EB)}, ppp).f);
    const auto q = ub_mul(u1, pp);
    const UB<P, 8> x3 = ub_sub_sub2(ub_sqr(rd), ppp, q);   // (0, 8p)
    fu_store<P>(out, x3.f);
    const UB<P, 2> y3 = ub_mul_add(rd, ub_sub_lazy(q, x3), s1, ub_neg_lazy(ppp));
    fu_store<P>(out + EB, y3.f);
  }
#endif
  // this += o   (add-2008-s)
  ZKP_DEV void add(const BkPoint& o) {
    if (o.v.inf) return;
    if (v.inf) {
      *this = o;
      return;
    }
    const auto u1 = ub_mul(v.x, o.v.zz);                   // 8*2
    const auto u2 = ub_mul(o.v.x, v.zz);
    const auto s1 = ub_mul(v.y, o.v.zzz);                  // 4*2
    const auto s2 = ub_mul(o.v.y, v.zzz);
    const auto pd = ub_sub(u2, u1);                        // (0, 4p)
    const auto rd = ub_sub(s2, s1);                        // (0, 4p)
    if (pd.f.template maybe_multiple_of_p<4>() && pd.f.template is_multiple_of_p<4>()) {   // same x: P = +-Q
      bk_same_x<F>(*this, rd.f.template is_multiple_of_p<4>());
      return;
    }
    const auto pp = ub_sqr(pd);                            // 16
    const auto ppp = ub_mul(pd, pp);
    const auto q = ub_mul(u1, pp);
    const UB<P, 8> x3 = ub_sub_sub2(ub_sqr(rd), ppp, q);   // R^2 - PPP - 2Q + 6p, one carry chain: (0, 8p)
    const UB<P, 2> y3 = ub_mul_add(rd, ub_sub_lazy(q, x3), s1, ub_neg_lazy(ppp));      // lazy-limb second factors: 4*11 + 2*3 = 50
    v.zz = ub_mul(ub_mul(v.zz, o.v.zz), pp);
    v.zzz = ub_mul(ub_mul(v.zzz, o.v.zzz), ppp);
    v.x = x3;
    v.y = {y3.f};
  }
  // 2 * this   (dbl-2008-s-1, a = 0; y == 0 cannot occur on these prime-order curves)
  ZKP_DEV BkPoint dbl() const {
    if (v.inf) return *this;
    const auto u = ub_dbl(v.y);                            // < 8p
    const auto vv = ub_sqr(u);                             // 64
    const auto w = ub_mul(u, vv);
    const auto s = ub_mul(v.x, vv);
    const auto x2 = ub_sqr(v.x);                           // 64
    const auto m = ub_add3(ub_dbl(x2), x2);                // < 6p
    const auto x3 = ub_sub(ub_sqr(m), ub_dbl(s));          // (0, 6p)
    const UB<P, 2> y3 = ub_mul_add(m, ub_sub_lazy(s, x3), w, ub_neg_lazy(v.y));         // 6*9 + 2*5 = 64
    BkPoint r;
    r.v.x = {x3.f};
    r.v.y = {y3.f};
    r.v.zz = ub_mul(vv, v.zz);
    r.v.zzz = ub_mul(w, v.zzz);
    r.v.inf = false;
    return r;
  }
};

// ------------------------------------------------------------------------------------------------ G2
template <class P>
struct BkPoint<Fp2<P>> {
  using F = Fp2<P>;
  using U = Fu<P>;
  XYZZu2<P> v;
  static constexpr int EB = 4 * U::L;
  static constexpr int BYTES = 8 * EB;
  ZKP_DEV bool is_inf() const { return v.inf; }
  ZKP_DEV static BkPoint inf() {
    BkPoint r;
    const UB<P, 4> z4{U::zero()};
    const UB<P, 2> z2{U::zero()};
    r.v.x = {z4, z4};
    r.v.y = {z2, z2};
    r.v.zz = {z2, z2};
    r.v.zzz = {z2, z2};
    r.v.inf = true;
    return r;
  }
  ZKP_DEV static BkPoint load(const void* p) {
    const char* q = reinterpret_cast<const char*>(p);
    BkPoint r;
    r.v.zz = {{fu_load<P>(q + 4 * EB)}, {fu_load<P>(q + 5 * EB)}};
    r.v.inf = fu_is_zero(r.v.zz.c0.f) && fu_is_zero(r.v.zz.c1.f);
    r.v.x = {{fu_load<P>(q)}, {fu_load<P>(q + EB)}};
    r.v.y = {{fu_load<P>(q + 2 * EB)}, {fu_load<P>(q + 3 * EB)}};
    r.v.zzz = {{fu_load<P>(q + 6 * EB)}, {fu_load<P>(q + 7 * EB)}};
    return r;
  }
  ZKP_DEV void store(void* p) const {
    char* q = reinterpret_cast<char*>(p);
    const BkPoint z = inf();
    const XYZZu2<P>& s = v.inf ? z.v : v;
    fu_store<P>(q, s.x.c0.f);
    fu_store<P>(q + EB, s.x.c1.f);
    fu_store<P>(q + 2 * EB, s.y.c0.f);
    fu_store<P>(q + 3 * EB, s.y.c1.f);
    fu_store<P>(q + 4 * EB, s.zz.c0.f);
    fu_store<P>(q + 5 * EB, s.zz.c1.f);
    fu_store<P>(q + 6 * EB, s.zzz.c0.f);
    fu_store<P>(q + 7 * EB, s.zzz.c1.f);
  }
  ZKP_DEV static BkPoint from_sat(const XYZZ<F>& s) {
    if (s.is_inf()) return inf();
    BkPoint r;
    r.v.x = {{U::from_sat_reduced(s.x.c0)}, {U::from_sat_reduced(s.x.c1)}};
    r.v.y = {{U::from_sat_reduced(s.y.c0)}, {U::from_sat_reduced(s.y.c1)}};
    r.v.zz = {{U::from_sat_reduced(s.zz.c0)}, {U::from_sat_reduced(s.zz.c1)}};
    r.v.zzz = {{U::from_sat_reduced(s.zzz.c0)}, {U::from_sat_reduced(s.zzz.c1)}};
    r.v.inf = false;
    return r;
  }
  ZKP_DEV XYZZ<F> to_sat() const {
    if (v.inf) return XYZZ<F>::inf();
    XYZZ<F> r;
    r.x = {v.x.c0.f.to_sat(), v.x.c1.f.to_sat()};
    r.y = {v.y.c0.f.to_sat(), v.y.c1.f.to_sat()};
    r.zz = {v.zz.c0.f.to_sat(), v.zz.c1.f.to_sat()};
    r.zzz = {v.zzz.c0.f.to_sat(), v.zzz.c1.f.to_sat()};
    return r;
  }
  // a^2 with both components lazily reduced sums (KA^2 + KA^2 <= MULCAP)
  template <int KA>
  ZKP_DEV static UB2<P, 2> sqr_lazy(const UB2<P, KA>& a) {
    return {ub_mul_add(a.c0, a.c0, a.c1, ub_neg<KA>(a.c1)), ub_mul_add(a.c0, a.c1, a.c0, a.c1)};
  }
  ZKP_DEV static UB2<P, 4> below4(const UB2<P, 8>& a) {
    return {UB<P, 4>{U::template csub<4>(a.c0.f)}, UB<P, 4>{U::template csub<4>(a.c1.f)}};
  }
  ZKP_DEV void add(const BkPoint& o) {
    if (o.v.inf) return;
    if (v.inf) {
      *this = o;
      return;
    }
    const auto u1 = ub2_mul(v.x, o.v.zz);                  // 4*2 + 4*2
    const auto u2 = ub2_mul(o.v.x, v.zz);
    const auto s1 = ub2_mul(v.y, o.v.zzz);
    const auto s2 = ub2_mul(o.v.y, v.zzz);
    const auto pd = ub2_sub(u2, u1);                       // (0, 4p)
    const auto rd = ub2_sub(s2, s1);                       // (0, 4p)
    if (pd.c0.f.template maybe_multiple_of_p<4>() && pd.c1.f.template maybe_multiple_of_p<4>() &&
        pd.c0.f.template is_multiple_of_p<4>() && pd.c1.f.template is_multiple_of_p<4>()) {           // same x: P = +-Q
      bk_same_x<F>(*this, rd.c0.f.template is_multiple_of_p<4>() && rd.c1.f.template is_multiple_of_p<4>());
      return;
    }
    const auto pp = sqr_lazy(pd);                          // 16 + 16
    const auto ppp = ub2_mul(pd, pp);                      // 4*2 + 4*2
    const auto q = ub2_mul(u1, pp);
    const auto rr = sqr_lazy(rd);
    const auto x3 = below4(ub2_sub(ub2_sub(rr, ppp), ub2_dbl(q)));              // (0, 8p) -> < 4p
    const auto t = ub2_sub(q, x3);                         // (0, 6p)
    static_assert(4 * 6 * 2 + 2 * 2 * 2 <= UBLimits<P>::MULCAP, "y3 exceeds the slack");
    const UB<P, 2> y30{U::mul_add4(rd.c0.f, t.c0.f, rd.c1.f, ub_neg<6>(t.c1).f, s1.c0.f, ub_neg<2>(ppp.c0).f, s1.c1.f, ppp.c1.f)};
    const UB<P, 2> y31{U::mul_add4(rd.c0.f, t.c1.f, rd.c1.f, t.c0.f, s1.c0.f, ub_neg<2>(ppp.c1).f, s1.c1.f, ub_neg<2>(ppp.c0).f)};
    v.zz = ub2_mul(ub2_mul(v.zz, o.v.zz), pp);
    v.zzz = ub2_mul(ub2_mul(v.zzz, o.v.zzz), ppp);
    v.x = x3;
    v.y = {y30, y31};
  }
  // out = a + b with the operands STREAMED from memory (global or LDS; out may be a): at no time are both points in registers —
  // the eight Fq2 coordinates of two G2 points are 144 VGPRs before the first product, and add() above compiled to 254 VGPRs + 312 B
  // of scratch (one or two waves per SIMD) in every reduction kernel.  zz1 zz2 and zzz1 zzz2 are parked in out.zz / out.zzz until
  // pp / ppp exist.  Round 3: the G2 reduction cost 0.70 ms of a 7.3 ms proof for 0.23 ms of multiplier work.
  static constexpr bool MEM_ADD = true;
#define ZKP_SCHED_FENCE() __asm__ volatile("" ::: "memory")
  ZKP_DEV static UB2<P, 2> ld2(const char* q, int slot) { return {{fu_load<P>(q + (2 * slot) * EB)}, {fu_load<P>(q + (2 * slot + 1) * EB)}}; }
  ZKP_DEV static void st2(char* q, int slot, const Fu<P>& c0, const Fu<P>& c1) {
    fu_store<P>(q + (2 * slot) * EB, c0);
    fu_store<P>(q + (2 * slot + 1) * EB, c1);
  }
  ZKP_DEV static void copy_point(char* out, const char* in) {
    if (out == in) return;
#pragma unroll
    for (int k = 0; k < 8; k++) fu_store<P>(out + k * EB, fu_load<P>(in + k * EB));
  }
  template <int KA>
  ZKP_DEV static UB2<P, 2> sqr_c(const UB2<P, KA>& a) {    // complex squaring: (a0 + a1)(a0 - a1 + KA p), (2 a0) a1
    return {ub_mul(ub_add(a.c0, a.c1), ub_sub(a.c0, a.c1)), ub_mul(ub_dbl(a.c0), a.c1)};
  }
  // out = 2 a, operands streamed like add_mem (out may be a; a is not the identity)
  ZKP_DEV static void dbl_mem(const char* a, char* out) {
    UB2<P, 2> vv;
    {
      const auto u = ub2_dbl(ld2(a, 1));                   // 2y < 4p
      vv = sqr_c(u);
      const auto zz3 = ub2_mul(vv, ld2(a, 2));
      st2(out, 2, zz3.c0.f, zz3.c1.f);
    }
    ZKP_SCHED_FENCE();
    UB2<P, 6> m;
    UB2<P, 6> t;
    {
      const UB2<P, 4> x = {UB<P, 4>{fu_load<P>(a)}, UB<P, 4>{fu_load<P>(a + EB)}};
      const auto s = ub2_mul(x, vv);
      const auto x2 = sqr_c(x);
      m = {ub_add3(ub_dbl(x2.c0), x2.c0), ub_add3(ub_dbl(x2.c1), x2.c1)};
      const auto mm = sqr_c(m);                            // 12 * 12 = 144
      const auto x3w = ub2_sub(mm, ub2_dbl(s));            // (0, 6p)
      const UB2<P, 4> x3 = {UB<P, 4>{U::template csub<4>(x3w.c0.f)}, UB<P, 4>{U::template csub<4>(x3w.c1.f)}};
      t = ub2_sub(s, x3);                                  // (0, 6p)
      st2(out, 0, x3.c0.f, x3.c1.f);
    }
    ZKP_SCHED_FENCE();
    const UB2<P, 2> y = ld2(a, 1);                         // out.y is written last: out may be a
    const auto w = ub2_mul(ub2_dbl(y), vv);                // (2y)^3
    {
      const auto zzz3 = ub2_mul(w, ld2(a, 3));
      st2(out, 3, zzz3.c0.f, zzz3.c1.f);
    }
    static_assert(6 * 6 + 6 * 7 + 2 * 2 * 2 <= UBLimits<P>::MULCAP, "y3 exceeds the slack");
    const auto ny0 = ub_neg<2>(y.c0);
    const Fu<P> y30 = U::mul_add4(m.c0.f, t.c0.f, m.c1.f, ub_neg_lazy(t.c1).f, w.c0.f, ny0.f, w.c1.f, y.c1.f);
    const Fu<P> y31 = U::mul_add4(m.c0.f, t.c1.f, m.c1.f, t.c0.f, w.c0.f, ub_neg_lazy(y.c1).f, w.c1.f, ny0.f);
    st2(out, 1, y30, y31);
  }
  ZKP_DEV static void add_mem(const char* a, const char* b, char* out) {
    const UB2<P, 2> zz1 = ld2(a, 2), zz2 = ld2(b, 2);
    const bool inf1 = fu_is_zero(zz1.c0.f) && fu_is_zero(zz1.c1.f), inf2 = fu_is_zero(zz2.c0.f) && fu_is_zero(zz2.c1.f);
    if (inf2) {
      copy_point(out, a);
      return;
    }
    if (inf1) {
      copy_point(out, b);
      return;
    }
    UB2<P, 4> pd;
    UB2<P, 2> u1;
    {
      const UB2<P, 4> x1 = {UB<P, 4>{fu_load<P>(a)}, UB<P, 4>{fu_load<P>(a + EB)}};
      u1 = ub2_mul(x1, zz2);
      const UB2<P, 4> x2 = {UB<P, 4>{fu_load<P>(b)}, UB<P, 4>{fu_load<P>(b + EB)}};
      pd = ub2_sub(ub2_mul(x2, zz1), u1);                  // (0, 4p)
    }
    const bool same_x = pd.c0.f.template maybe_multiple_of_p<4>() && pd.c1.f.template maybe_multiple_of_p<4>() &&
                        pd.c0.f.template is_multiple_of_p<4>() && pd.c1.f.template is_multiple_of_p<4>();
    if (!same_x) {                                         // (the exceptional path below still needs a's coordinates in place)
      const UB2<P, 2> zzp = ub2_mul(zz1, zz2);
      st2(out, 2, zzp.c0.f, zzp.c1.f);                     // parked: a's and b's zz are consumed
    }
    ZKP_SCHED_FENCE();                                     // keep the next stage's loads below this point (register pressure)
    UB2<P, 2> s1;
    UB2<P, 4> rd;
    {
      const UB2<P, 2> zzz1 = ld2(a, 3), zzz2 = ld2(b, 3);
      const UB2<P, 2> y1 = ld2(a, 1);
      s1 = ub2_mul(y1, zzz2);
      const UB2<P, 2> y2 = ld2(b, 1);
      rd = ub2_sub(ub2_mul(y2, zzz1), s1);                 // (0, 4p)
      if (same_x) {                                        // P = +-Q (crafted inputs only): doubling or the identity, out of line
        bk_same_x_mem<F>(a, out, rd.c0.f.template is_multiple_of_p<4>() && rd.c1.f.template is_multiple_of_p<4>());
        return;
      }
      const UB2<P, 2> zzzp = ub2_mul(zzz1, zzz2);
      st2(out, 3, zzzp.c0.f, zzzp.c1.f);                   // parked likewise
      st2(out, 1, s1.c0.f, s1.c1.f);                       // s1 waits in out.y until y3 needs it
    }
    ZKP_SCHED_FENCE();
    const UB2<P, 2> pp = {ub_mul(ub_add(pd.c0, pd.c1), ub_sub(pd.c0, pd.c1)), ub_mul(ub_dbl(pd.c0), pd.c1)};   // complex squaring: 8*8, 8*4
    const auto ppp = ub2_mul(pd, pp);
    {
      const UB2<P, 2> zzp = ld2(out, 2);
      const auto zz3 = ub2_mul(zzp, pp);
      st2(out, 2, zz3.c0.f, zz3.c1.f);
      const UB2<P, 2> zzzp = ld2(out, 3);
      const auto zzz3 = ub2_mul(zzzp, ppp);
      st2(out, 3, zzz3.c0.f, zzz3.c1.f);
    }
    ZKP_SCHED_FENCE();
    const auto q = ub2_mul(u1, pp);
    const UB2<P, 2> rr = {ub_mul(ub_add(rd.c0, rd.c1), ub_sub(rd.c0, rd.c1)), ub_mul(ub_dbl(rd.c0), rd.c1)};
    const UB2<P, 4> x3 = {UB<P, 4>{U::template csub<4>(ub_sub_sub2(rr.c0, ppp.c0, q.c0).f)},
                          UB<P, 4>{U::template csub<4>(ub_sub_sub2(rr.c1, ppp.c1, q.c1).f)}};
    st2(out, 0, x3.c0.f, x3.c1.f);
    ZKP_SCHED_FENCE();
    const auto t = ub2_sub(q, x3);                         // (0, 6p)
    static_assert(4 * 6 + 4 * 7 + 2 * 2 * 2 <= UBLimits<P>::MULCAP, "y3 exceeds the slack");
    const auto nppp0 = ub_neg<2>(ppp.c0);
    const UB2<P, 2> s1b = ld2(out, 1);
    const Fu<P> y30 = U::mul_add4(rd.c0.f, t.c0.f, rd.c1.f, ub_neg_lazy(t.c1).f, s1b.c0.f, nppp0.f, s1b.c1.f, ppp.c1.f);
    const Fu<P> y31 = U::mul_add4(rd.c0.f, t.c1.f, rd.c1.f, t.c0.f, s1b.c0.f, ub_neg_lazy(ppp.c1).f, s1b.c1.f, nppp0.f);
    st2(out, 1, y30, y31);
  }
  ZKP_DEV BkPoint dbl() const {
    if (v.inf) return *this;
    const auto u = ub2_dbl(v.y);                           // < 4p
    const auto vv = sqr_lazy(u);                           // 16 + 16
    const auto w = ub2_mul(u, vv);                         // 4*2 + 4*2
    const auto s = ub2_mul(v.x, vv);
    const auto x2 = sqr_lazy(v.x);
    const UB2<P, 6> m = {ub_add3(ub_dbl(x2.c0), x2.c0), ub_add3(ub_dbl(x2.c1), x2.c1)};
    const auto mm = sqr_lazy(m);                           // 36 + 36
    const auto x3w = ub2_sub(mm, ub2_dbl(s));              // (0, 6p)
    const UB2<P, 4> x3 = {UB<P, 4>{U::template csub<4>(x3w.c0.f)}, UB<P, 4>{U::template csub<4>(x3w.c1.f)}};
    const auto t = ub2_sub(s, x3);                         // (0, 6p)
    static_assert(6 * 6 * 2 + 2 * 2 * 2 <= UBLimits<P>::MULCAP, "y3 exceeds the slack");
    const UB<P, 2> y30{U::mul_add4(m.c0.f, t.c0.f, m.c1.f, ub_neg<6>(t.c1).f, w.c0.f, ub_neg<2>(v.y.c0).f, w.c1.f, v.y.c1.f)};
    const UB<P, 2> y31{U::mul_add4(m.c0.f, t.c1.f, m.c1.f, t.c0.f, w.c0.f, ub_neg<2>(v.y.c1).f, w.c1.f, ub_neg<2>(v.y.c0).f)};
    BkPoint r;
    r.v.x = x3;
    r.v.y = {y30, y31};
    r.v.zz = ub2_mul(vv, v.zz);
    r.v.zzz = ub2_mul(w, v.zzz);
    r.v.inf = false;
    return r;
  }
};

template <class F>
__device__ __noinline__ void bk_same_x(BkPoint<F>& a, bool same_y) {
  a = same_y ? a.dbl() : BkPoint<F>::inf();
}
template <class F>
__device__ __noinline__ void bk_same_x_mem(const char* a, char* out, bool same_y) {
  if (same_y) BkPoint<F>::dbl_mem(a, out);
  else BkPoint<F>::inf().store(out);
}

}  // namespace zkp
