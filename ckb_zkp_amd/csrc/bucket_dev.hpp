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

}  // namespace zkp
