// Quad-cooperative point arithmetic for the latency-bound tails of the bucket reduction (round 4).
//
// The top of the reduction pyramid, the LDS trees of the segmented sums and the final weighting are chains of DEPENDENT point
// additions with fewer additions per step than the machine has lanes: one lane per addition runs add-2008-s as 13 products + 2
// squarings one after the other (~3000 VALU instructions, ~5 us).  Here FOUR adjacent lanes (a DPP quad) share one addition: the
// formulas have four levels of independent products
//     { x1 zz2, x2 zz1, y1 zzz2, y2 zzz1 | zz1 zz2, zzz1 zzz2 }  ->  { P^2 }  ->  { P^3, U1 P^2, R^2, (zz1 zz2) P^2 }
//     ->  { R (Q - X3) - S1 P^3  (one lazily reduced sum of two products),  (zzz1 zzz2) P^3 }
// so a quad needs 5 product latencies + one lazily reduced sum instead of 15, and a doubling (dbl-2008-s-1) 3 instead of 9.  Lanes
// exchange field elements with `v_mov_b32 ... quad_perm` (one instruction per limb for all four lanes), load their first operands
// straight from memory by role, and execute ONE instruction stream: every lane multiplies in every round, some results are unused
// (the waste is lanes, not time: these steps have lanes to spare).  Same formulas, same value bounds as BkPoint::add_mem / dbl
// (bucket_dev.hpp), so the stored points are the same field elements; P = +-Q and identity operands fall back to the exact
// single-lane path.  G1 only (an Fq2 variant would split every Fq2 product over the quad; not built).
//
// Serves the pyramid behind ark-ec 0.2 `VariableBaseMSM::multi_scalar_mul` (reference call sites:
// /root/reference/groth16/src/prover.rs:187,190,220; marlin/src/pc/kzg10.rs:109,118,137,146).
#pragma once
#include "bucket_dev.hpp"

namespace zkp {

template <int K, class P>
ZKP_DEV Fu<P> quad_bcast(const Fu<P>& x) {                 // the value lane K of every quad holds, in all four lanes
  Fu<P> r;
#pragma unroll
  for (int i = 0; i < Fu<P>::L; i++)
    r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.v[i], K * 0x55, 0xf, 0xf, true);
  return r;
}
template <class P>
ZKP_DEV Fu<P> quad_sel(int role, const Fu<P>& a0, const Fu<P>& a1, const Fu<P>& a2, const Fu<P>& a3) {
  Fu<P> r;
#pragma unroll
  for (int i = 0; i < Fu<P>::L; i++) {
    const uint32_t lo = (role & 1) ? a1.v[i] : a0.v[i], hi = (role & 1) ? a3.v[i] : a2.v[i];
    r.v[i] = (role & 2) ? hi : lo;
  }
  return r;
}

// out = a + b (BkPoint<Fp<P>> in memory: x | y | zz | zzz, L limbs each), computed by the four lanes of a quad; role = lane & 3.
// All four lanes must be active and pass the same pointers.  out may alias a or b.
template <class P>
ZKP_DEV void quad_add_mem(const char* a, const char* b, char* out, int role) {
  using U = Fu<P>;
  using B = BkPoint<Fp<P>>;
  constexpr int EB = B::EB;
  const U zz1 = fu_load<P>(a + 2 * EB), zz2 = fu_load<P>(b + 2 * EB);
  const bool inf1 = fu_is_zero(zz1), inf2 = fu_is_zero(zz2);       // uniform over the quad
  if (inf1 || inf2) {
    if (role == 0) B::copy_point(out, inf2 ? a : b);
    return;
  }
  // round 1 — role 0: x1 zz2 (= u1) | 1: x2 zz1 (= u2) | 2: y1 zzz2 (= s1) | 3: y2 zzz1 (= s2)         bounds 8*2, 4*2
  const char* mine = (role & 1) ? b : a;
  const char* other = (role & 1) ? a : b;
  const U t1 = U::mul(fu_load<P>(mine + (role >> 1) * EB), fu_load<P>(other + (2 + (role >> 1)) * EB));
  // round 1b — role 2: zzz1 zzz2 | others: zz1 zz2 (role 3 keeps it)                                     bounds 2*2
  const int f2 = role == 2 ? 3 : 2;
  const U t2 = U::mul(fu_load<P>(a + f2 * EB), fu_load<P>(b + f2 * EB));
  const U u1 = quad_bcast<0>(t1), u2 = quad_bcast<1>(t1), s1 = quad_bcast<2>(t1), s2 = quad_bcast<3>(t1);
  const UB<P, 4> pd = ub_sub(UB<P, 2>{u2}, UB<P, 2>{u1});          // (0, 4p), every lane
  const UB<P, 4> rd = ub_sub(UB<P, 2>{s2}, UB<P, 2>{s1});
  if (pd.f.template maybe_multiple_of_p<4>() && pd.f.template is_multiple_of_p<4>()) {     // P = +-Q: uniform over the quad
    if (role == 0) B::add_mem(a, b, out);                          // the exact path (doubling or the identity), one lane
    return;
  }
  const UB<P, 2> pp = ub_sqr(pd);                                  // round 2, every lane
  // round 3 — role 0: P^3 = pd pp | 1: Q = u1 pp | 2: R^2 | 3: zz3 = (zz1 zz2) pp                       bounds <= 4*4
  const U t3 = U::mul(quad_sel<P>(role, pd.f, u1, rd.f, t2), quad_sel<P>(role, pp.f, pp.f, rd.f, pp.f));
  const UB<P, 2> ppp{quad_bcast<0>(t3)}, q{quad_bcast<1>(t3)}, rr{quad_bcast<2>(t3)};
  const UB<P, 8> x3 = ub_sub_sub2(rr, ppp, q);                     // R^2 - P^3 - 2Q + 6p: (0, 8p), every lane
  // round 4 — role 0 (and 1, 3, unused): y3 = rd (Q - X3) + s1 (-P^3), lazy second factors, 4*11 + 2*3 | role 2: zzz3 = (zzz1 zzz2) P^3
  const bool r2 = role == 2;
  const U la = ub_sub_lazy(q, x3).f, lb = ub_neg_lazy(ppp).f, zero = U::zero();
  U a4, b4, c4, d4;
#pragma unroll
  for (int i = 0; i < U::L; i++) {
    a4.v[i] = r2 ? t2.v[i] : rd.f.v[i];
    b4.v[i] = r2 ? ppp.f.v[i] : la.v[i];
    c4.v[i] = r2 ? zero.v[i] : s1.v[i];
    d4.v[i] = r2 ? zero.v[i] : lb.v[i];
  }
  const U t4 = U::mul_add(a4, b4, c4, d4);
  if (role == 0) fu_store<P>(out + EB, t4);                        // y3 < 2p
  else if (role == 1) fu_store<P>(out, x3.f);
  else if (role == 2) fu_store<P>(out + 3 * EB, t4);               // zzz3
  else fu_store<P>(out + 2 * EB, t3);                              // zz3
}

// out = 2 a  (dbl-2008-s-1, a = 0 curve), by the four lanes of a quad.  out may alias a.
template <class P>
ZKP_DEV void quad_dbl_mem(const char* a, char* out, int role) {
  using U = Fu<P>;
  using B = BkPoint<Fp<P>>;
  constexpr int EB = B::EB;
  const UB<P, 2> zz{fu_load<P>(a + 2 * EB)};
  if (fu_is_zero(zz.f)) {                                          // uniform over the quad
    if (role == 0) B::copy_point(out, a);
    return;
  }
  const UB<P, 8> x{fu_load<P>(a)};
  const UB<P, 4> y{fu_load<P>(a + EB)};
  const UB<P, 2> zzz{fu_load<P>(a + 3 * EB)};
  const UB<P, 8> u = ub_dbl(y);                                    // < 8p
  // round 1 — role 1: x^2 | others: V = u^2                                                             bounds 64
  const U t1 = U::mul(role == 1 ? x.f : u.f, role == 1 ? x.f : u.f);
  const UB<P, 2> vv{quad_bcast<0>(t1)}, x2{quad_bcast<1>(t1)};
  const UB<P, 6> m = ub_add3(ub_dbl(x2), x2);                      // 3 x^2 < 6p
  // round 2 — role 0: W = u V | 1: S = x V | 2: M^2 | 3: zz3 = V zz                                     bounds 8*2, 8*2, 36, 2*2
  const U t2 = U::mul(quad_sel<P>(role, u.f, x.f, m.f, vv.f), quad_sel<P>(role, vv.f, vv.f, m.f, zz.f));
  const UB<P, 2> w{quad_bcast<0>(t2)}, s{quad_bcast<1>(t2)}, mm{quad_bcast<2>(t2)};
  const UB<P, 6> x3 = ub_sub(mm, ub_dbl(s));                       // (0, 6p)
  // round 3 — role 0 (1, 2 unused): y3 = M (S - X3) + W (-y), lazy second factors, 6*9 + 2*5 | role 3: zzz3 = W zzz
  const bool r3 = role == 3;
  const U la = ub_sub_lazy(s, x3).f, lb = ub_neg_lazy(y).f, zero = U::zero();
  U a4, b4, c4, d4;
#pragma unroll
  for (int i = 0; i < U::L; i++) {
    a4.v[i] = r3 ? w.f.v[i] : m.f.v[i];
    b4.v[i] = r3 ? zzz.f.v[i] : la.v[i];
    c4.v[i] = r3 ? zero.v[i] : w.f.v[i];
    d4.v[i] = r3 ? zero.v[i] : lb.v[i];
  }
  const U t4 = U::mul_add(a4, b4, c4, d4);
  if (role == 0) fu_store<P>(out + EB, t4);                        // y3
  else if (role == 1) fu_store<P>(out, x3.f);
  else if (role == 3) {
    fu_store<P>(out + 2 * EB, t2);                                 // zz3
    fu_store<P>(out + 3 * EB, t4);                                 // zzz3
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// G2: the quad is two Fq2 products wide — lane (2 prod + comp) computes component `comp` of product `prod`, one lazily reduced
// sum of two Fq products (the schoolbook rows of ub2_mul, bucket_dev.hpp).  add-2008-s over Fq2 then takes five such rounds
// ({u1, u2}, {s1, s2}, {zz1 zz2, zzz1 zzz2}, {P^3, Q}, {zz3, zzz3}), one round of plain products (the complex squarings P^2 and R^2,
// four products for four lanes) and the two components of y3 (sums of four products) instead of 13 Fq2 products, two squarings
// and y3 one after the other: ~11 instead of ~48 Fq-product latencies.
template <class P, int KB>
ZKP_DEV Fu<P> quad_fq2_comp(int comp, const Fu<P>& a0, const Fu<P>& a1, const Fu<P>& b0, const Fu<P>& b1) {
  using U = Fu<P>;
  const U nb1 = U::template neg_lazy<KB + 1>(b1);
  U f0, f1;
#pragma unroll
  for (int i = 0; i < U::L; i++) {
    f0.v[i] = comp ? b1.v[i] : b0.v[i];
    f1.v[i] = comp ? b0.v[i] : nb1.v[i];
  }
  return U::mul_add(a0, f0, a1, f1);                                // comp 0: a0 b0 - a1 b1 ; comp 1: a0 b1 + a1 b0
}
template <class P>
ZKP_DEV Fu<P> quad_pick(bool second, const Fu<P>& a, const Fu<P>& b) {
  Fu<P> r;
#pragma unroll
  for (int i = 0; i < Fu<P>::L; i++) r.v[i] = second ? b.v[i] : a.v[i];
  return r;
}

// out = a + b for BkPoint<Fp2<P>> in memory (x.c0 | x.c1 | y.c0 | y.c1 | zz.c0 | zz.c1 | zzz.c0 | zzz.c1).  out may alias a or b.
template <class P>
ZKP_DEV void quad_add_mem2(const char* a, const char* b, char* out, int role) {
  using U = Fu<P>;
  using B = BkPoint<Fp2<P>>;
  constexpr int EB = B::EB;
  const int prod = role >> 1, comp = role & 1;
  auto ld = [](const char* p, int k) { return fu_load<P>(p + k * EB); };
  const U zz1a = ld(a, 4), zz1b = ld(a, 5), zz2a = ld(b, 4), zz2b = ld(b, 5);
  const bool inf1 = fu_is_zero(zz1a) && fu_is_zero(zz1b), inf2 = fu_is_zero(zz2a) && fu_is_zero(zz2b);    // uniform over the quad
  if (inf1 || inf2) {
    if (role == 0) B::copy_point(out, inf2 ? a : b);
    return;
  }
  const char* mine = prod ? b : a;
  const char* other = prod ? a : b;
  // round 1 — product 0: u1 = x1 zz2 | product 1: u2 = x2 zz1                                               bounds 4*2 + 4*3
  const U t1 = quad_fq2_comp<P, 2>(comp, ld(mine, 0), ld(mine, 1), quad_pick<P>(prod, zz2a, zz1a), quad_pick<P>(prod, zz2b, zz1b));
  const U u10 = quad_bcast<0>(t1), u11 = quad_bcast<1>(t1);
  const UB<P, 4> pd0 = ub_sub(UB<P, 2>{quad_bcast<2>(t1)}, UB<P, 2>{u10}), pd1 = ub_sub(UB<P, 2>{quad_bcast<3>(t1)}, UB<P, 2>{u11});
  // round 2 — s1 = y1 zzz2 | s2 = y2 zzz1                                                                    bounds 2*2 + 2*3
  const U t2 = quad_fq2_comp<P, 2>(comp, ld(mine, 2), ld(mine, 3), ld(other, 6), ld(other, 7));
  const U s10 = quad_bcast<0>(t2), s11 = quad_bcast<1>(t2);
  const UB<P, 4> rd0 = ub_sub(UB<P, 2>{quad_bcast<2>(t2)}, UB<P, 2>{s10}), rd1 = ub_sub(UB<P, 2>{quad_bcast<3>(t2)}, UB<P, 2>{s11});
  if (pd0.f.template maybe_multiple_of_p<4>() && pd1.f.template maybe_multiple_of_p<4>() && pd0.f.template is_multiple_of_p<4>() &&
      pd1.f.template is_multiple_of_p<4>()) {                       // P = +-Q: uniform over the quad; the exact path, one lane
    if (role == 0) B::add_mem(a, b, out);
    return;
  }
  // round 3 — zz1 zz2 | zzz1 zzz2
  const U t3 = quad_fq2_comp<P, 2>(comp, ld(a, 4 + 2 * prod), ld(a, 5 + 2 * prod), ld(b, 4 + 2 * prod), ld(b, 5 + 2 * prod));
  const U zz120 = quad_bcast<0>(t3), zz121 = quad_bcast<1>(t3), zzz120 = quad_bcast<2>(t3), zzz121 = quad_bcast<3>(t3);
  // round 4 — the complex squarings, four plain products: P^2 (lanes 0, 1) and R^2 (lanes 2, 3); (v0 + v1)(v0 - v1) | (2 v0) v1, 8*8
  const UB<P, 4> v0{quad_pick<P>(prod, pd0.f, rd0.f)}, v1{quad_pick<P>(prod, pd1.f, rd1.f)};
  const U t4 = U::mul(quad_pick<P>(comp, ub_add(v0, v1).f, ub_dbl(v0).f), quad_pick<P>(comp, ub_sub(v0, v1).f, v1.f));
  const UB<P, 2> pp0{quad_bcast<0>(t4)}, pp1{quad_bcast<1>(t4)}, rr0{quad_bcast<2>(t4)}, rr1{quad_bcast<3>(t4)};
  // round 5 — P^3 = pd pp | Q = u1 pp                                                                        bounds 4*2 + 4*3
  const U t5 = quad_fq2_comp<P, 2>(comp, quad_pick<P>(prod, pd0.f, u10), quad_pick<P>(prod, pd1.f, u11), pp0.f, pp1.f);
  const UB<P, 2> ppp0{quad_bcast<0>(t5)}, ppp1{quad_bcast<1>(t5)}, q0{quad_bcast<2>(t5)}, q1{quad_bcast<3>(t5)};
  const UB<P, 4> x30{U::template csub<4>(ub_sub_sub2(rr0, ppp0, q0).f)}, x31{U::template csub<4>(ub_sub_sub2(rr1, ppp1, q1).f)};
  const UB<P, 6> tt0 = ub_sub(q0, x30), tt1 = ub_sub(q1, x31);
  // round 6 — zz3 = (zz1 zz2) pp | zzz3 = (zzz1 zzz2) P^3
  const U t6 = quad_fq2_comp<P, 2>(comp, quad_pick<P>(prod, zz120, zzz120), quad_pick<P>(prod, zz121, zzz121),
                                   quad_pick<P>(prod, pp0.f, ppp0.f), quad_pick<P>(prod, pp1.f, ppp1.f));
  // round 7 — y3, one component per lane (lanes 2, 3 repeat 0, 1): sums of four products with ONE lazy factor each
  const U nppp0 = ub_neg<2>(ppp0).f;
  const U fb = quad_pick<P>(comp, tt0.f, tt1.f), fd = quad_pick<P>(comp, ub_neg_lazy(tt1).f, tt0.f);
  const U ff = quad_pick<P>(comp, nppp0, ub_neg_lazy(ppp1).f), fh = quad_pick<P>(comp, ppp1.f, nppp0);
  const U t7 = U::mul_add4(rd0.f, fb, rd1.f, fd, s10, ff, s11, fh);
  if (prod == 0) {
    fu_store<P>(out + comp * EB, comp ? x31.f : x30.f);
    fu_store<P>(out + (2 + comp) * EB, t7);
    fu_store<P>(out + (4 + comp) * EB, t6);
  } else {
    fu_store<P>(out + (6 + comp) * EB, t6);
  }
}

template <class F>
struct QuadCoop {
  static constexpr bool ON = false;
  static constexpr int GROUP = 0;
  using P = void;
};
#ifndef ZKP_NO_QUAD_COOP
template <class P_>
struct QuadCoop<Fp<P_>> {
  static constexpr bool ON = BkPoint<Fp<P_>>::MEM_ADD;
  static constexpr int GROUP = 1;
  using P = P_;
};
// G2 on 9-limb fields only (BN254): the 14-limb BLS12-381 quad kernel would spill
template <class P_>
struct QuadCoop<Fp2<P_>> {
  static constexpr bool ON = Fu<P_>::L <= 9;
  static constexpr int GROUP = 2;
  using P = P_;
};
#endif

// one entry point for both groups
template <class F>
ZKP_DEV void quad_add_any(const char* a, const char* b, char* out, int role) {
  if constexpr (QuadCoop<F>::GROUP == 1) quad_add_mem<typename QuadCoop<F>::P>(a, b, out, role);
  else quad_add_mem2<typename QuadCoop<F>::P>(a, b, out, role);
}

}  // namespace zkp
