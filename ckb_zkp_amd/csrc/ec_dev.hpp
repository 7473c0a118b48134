// Short-Weierstrass (a = 0) group arithmetic over F = Fq (G1) or Fq2 (G2) for the MSM kernels.
//
// Replaces the `add_assign_mixed` / `add_assign` / `double_in_place` calls that ark-ec 0.2's
// `VariableBaseMSM::multi_scalar_mul` makes on `GroupProjective` (call sites in the reference:
// /root/reference/groth16/src/prover.rs:187,190,220).  ark uses Jacobian (X,Y,Z); the kernels use the
// extended-Jacobian "XYZZ" form (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), whose mixed addition costs
// 8M + 2S instead of 7M + 4S and needs no field doubling chains.  A group element has one canonical
// affine form, so the choice of projective system cannot change the (bit-exact) result.
//
// All exceptional cases are handled (the reference's keys DO contain identity points — variables that
// never occur in A or B, generator.rs:219-232 — and bucket sums can collide):
//   acc = inf, P = inf, acc == P (-> doubling), acc == -P (-> inf).
// Affine identity is encoded as (0, 0), which is on neither curve (b != 0).
#pragma once
#include "field_dev.hpp"

namespace zkp {

template <class F>
struct Affine {
  F x, y;
  ZKP_DEV bool is_inf() const { return x.is_zero() && y.is_zero(); }
  ZKP_DEV static Affine inf() { return {F::zero(), F::zero()}; }
  ZKP_DEV static Affine load(const void* p) {
    const char* q = reinterpret_cast<const char*>(p);
    return {F::load(q), F::load(q + 4 * F::N)};
  }
  ZKP_DEV static Affine load_nt(const void* p) {
    const char* q = reinterpret_cast<const char*>(p);
    return {F::load_nt(q), F::load_nt(q + 4 * F::N)};
  }
  ZKP_DEV void store(void* p) const {
    char* q = reinterpret_cast<char*>(p);
    x.store(q);
    y.store(q + 4 * F::N);
  }
  static constexpr int BYTES = 8 * F::N;
};

template <class F>
struct XYZZ;
// exceptional-case paths (P + P inside add/madd) are rare: keep ONE out-of-line copy so that inlining the
// multiplier into the hot formulas does not triple their code size
template <class F>
__device__ __noinline__ XYZZ<F> xyzz_dbl_slow(XYZZ<F> p);
template <class F>
__device__ __noinline__ XYZZ<F> xyzz_dbl_affine_slow(Affine<F> p);

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  static constexpr int BYTES = 16 * F::N;
  ZKP_DEV bool is_inf() const { return zz.is_zero(); }
  ZKP_DEV static XYZZ inf() { return {F::one(), F::one(), F::zero(), F::zero()}; }
  ZKP_DEV static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one(), F::one()};
  }
  ZKP_DEV XYZZ neg() const { return {x, y.neg(), zz, zzz}; }
  ZKP_DEV static XYZZ load(const void* p) {
    const char* q = reinterpret_cast<const char*>(p);
    return {F::load(q), F::load(q + 4 * F::N), F::load(q + 8 * F::N), F::load(q + 12 * F::N)};
  }
  ZKP_DEV void store(void* p) const {
    char* q = reinterpret_cast<char*>(p);
    x.store(q);
    y.store(q + 4 * F::N);
    zz.store(q + 8 * F::N);
    zzz.store(q + 12 * F::N);
  }

  // 2*(affine p) -> XYZZ   (mdbl-2008-s-1, a = 0)
  ZKP_DEV static XYZZ dbl_affine(const Affine<F>& p) {
    if (p.is_inf() || p.y.is_zero()) return inf();
    F u = p.y.dbl();
    F v = u.sqr();
    F w = u * v;
    F s = p.x * v;
    F x2 = p.x.sqr();
    F m = x2.dbl() + x2;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * p.y;
    return {x3, y3, v, w};
  }
  // dbl-2008-s-1, a = 0
  ZKP_DEV XYZZ dbl() const {
    if (is_inf() || y.is_zero()) return inf();
    F u = y.dbl();
    F v = u.sqr();
    F w = u * v;
    F s = x * v;
    F x2 = x.sqr();
    F m = x2.dbl() + x2;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * y;
    return {x3, y3, v * zz, w * zzz};
  }
  // this += affine p   (madd-2008-s; 8M + 2S)
  ZKP_DEV void madd(const Affine<F>& p) {
    if (p.is_inf()) return;
    if (is_inf()) {
      *this = {p.x, p.y, F::one(), F::one()};
      return;
    }
    F u2 = p.x * zz;
    F s2 = p.y * zzz;
    F pp_ = u2 - x;
    F r = s2 - y;
    if (pp_.is_zero()) {
      if (r.is_zero()) *this = xyzz_dbl_affine_slow<F>(p);
      else *this = inf();
      return;
    }
    F pp = pp_.sqr();
    F ppp = pp_ * pp;
    F q = x * pp;
    F x3 = r.sqr() - ppp - q.dbl();
    F y3 = r * (q - x3) - y * ppp;
    x = x3;
    y = y3;
    zz = zz * pp;
    zzz = zzz * ppp;
  }
  // madd without the exceptional formulas: returns false (this untouched) when p = +-this, so that a hot loop can hand
  // the rare case to an exact out-of-band path instead of carrying the doubling code through its register budget
  ZKP_DEV bool madd_fast(const Affine<F>& p) {
    if (p.is_inf()) return true;
    if (is_inf()) {
      *this = {p.x, p.y, F::one(), F::one()};
      return true;
    }
    F u2 = p.x * zz;
    F s2 = p.y * zzz;
    F pp_ = u2 - x;
    if (pp_.is_zero()) return false;
    F r = s2 - y;
    F pp = pp_.sqr();
    F ppp = pp_ * pp;
    F q = x * pp;
    F x3 = r.sqr() - ppp - q.dbl();
    F y3 = r * (q - x3) - y * ppp;
    x = x3;
    y = y3;
    zz = zz * pp;
    zzz = zzz * ppp;
    return true;
  }
  // this += o   (add-2008-s; 12M + 2S)
  ZKP_DEV void add(const XYZZ& o) {
    if (o.is_inf()) return;
    if (is_inf()) {
      *this = o;
      return;
    }
    F u1 = x * o.zz;
    F u2 = o.x * zz;
    F s1 = y * o.zzz;
    F s2 = o.y * zzz;
    F pp_ = u2 - u1;
    F r = s2 - s1;
    if (pp_.is_zero()) {
      if (r.is_zero()) *this = xyzz_dbl_slow<F>(*this);
      else *this = inf();
      return;
    }
    F pp = pp_.sqr();
    F ppp = pp_ * pp;
    F q = u1 * pp;
    F x3 = r.sqr() - ppp - q.dbl();
    F y3 = r * (q - x3) - s1 * ppp;
    x = x3;
    y = y3;
    zz = zz * o.zz * pp;
    zzz = zzz * o.zzz * ppp;
  }
  // -> affine (one inversion)
  ZKP_DEV Affine<F> to_affine() const {
    if (is_inf()) return Affine<F>::inf();
    F i = (zz * zzz).inv();       // 1/(ZZ*ZZZ)
    F izz = i * zzz;              // 1/ZZ
    F izzz = i * zz;              // 1/ZZZ
    return {x * izz, y * izzz};
  }
  // -> ark Jacobian (X, Y, Z) without inversion: Z = ZZ*ZZZ, X = x*ZZ*ZZZ^2, Y = y*ZZ^3*ZZZ^2; identity = (0,1,0)
  ZKP_DEV void store_jacobian(void* p) const {
    char* q = reinterpret_cast<char*>(p);
    if (is_inf()) {
      F::zero().store(q);
      F::one().store(q + 4 * F::N);
      F::zero().store(q + 8 * F::N);
      return;
    }
    F z = zz * zzz;
    F t = zz * zzz.sqr();         // ZZ*ZZZ^2
    F X = x * t;
    F Y = y * t * zz.sqr();
    X.store(q);
    Y.store(q + 4 * F::N);
    z.store(q + 8 * F::N);
  }
};

template <class F>
__device__ __noinline__ XYZZ<F> xyzz_dbl_slow(XYZZ<F> p) { return p.dbl(); }
template <class F>
__device__ __noinline__ XYZZ<F> xyzz_dbl_affine_slow(Affine<F> p) { return XYZZ<F>::dbl_affine(p); }

}  // namespace zkp
