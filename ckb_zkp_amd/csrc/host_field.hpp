// Host-side field / curve arithmetic (CPU, 32-bit limbs, Montgomery): the scalars the Marlin orchestration needs and the
// Jacobian / XYZZ -> affine tails of commitments and proofs (one Fermat inversion per batch, Montgomery's trick).  Not a compute
// path: a proof's MSMs and NTTs run on the device; this replaces single-lane device launches whose only content was one inversion.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/zkp_accel.h"

namespace zkp {
namespace hostf {

namespace consts {
#include "field_constants.inc"
}

// ------------------------------------------------------------------------------------------------ host field arithmetic
// Montgomery arithmetic on 32-bit limbs for the scalars the orchestration needs (challenge powers, vanishing-polynomial
// values, blinding combinations) and for Montgomery -> canonical conversion of commitments fed to the transcript.
struct HostField {
  int N;
  const uint32_t *mod, *one, *r2, *pm2;
  uint32_t inv;
  using E = std::array<uint32_t, 12>;
  bool geq(const uint32_t* a) const {
    for (int i = N - 1; i >= 0; i--)
      if (a[i] != mod[i]) return a[i] > mod[i];
    return true;
  }
  E zero() const { return E{}; }
  E one_() const {
    E r{};
    memcpy(r.data(), one, 4 * N);
    return r;
  }
  E mul(const E& a, const E& b) const {
    uint32_t t[14] = {0};
    for (int i = 0; i < N; i++) {
      uint64_t c = 0;
      for (int j = 0; j < N; j++) {
        c += (uint64_t)a[j] * b[i] + t[j];
        t[j] = (uint32_t)c;
        c >>= 32;
      }
      c += t[N];
      t[N] = (uint32_t)c;
      t[N + 1] = (uint32_t)(c >> 32);
      const uint32_t m = t[0] * inv;
      c = ((uint64_t)m * mod[0] + t[0]) >> 32;
      for (int j = 1; j < N; j++) {
        c += (uint64_t)m * mod[j] + t[j];
        t[j - 1] = (uint32_t)c;
        c >>= 32;
      }
      c += t[N];
      t[N - 1] = (uint32_t)c;
      t[N] = t[N + 1] + (uint32_t)(c >> 32);
    }
    if (t[N] || geq(t)) {
      uint64_t br = 0;
      for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)t[i] - mod[i] - br;
        t[i] = (uint32_t)d;
        br = (d >> 63) & 1;
      }
    }
    E r{};
    memcpy(r.data(), t, 4 * N);
    return r;
  }
  E add(const E& a, const E& b) const {
    E r{};
    uint64_t c = 0;
    for (int i = 0; i < N; i++) {
      c += (uint64_t)a[i] + b[i];
      r[i] = (uint32_t)c;
      c >>= 32;
    }
    if (c || geq(r.data())) {
      uint64_t br = 0;
      for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)r[i] - mod[i] - br;
        r[i] = (uint32_t)d;
        br = (d >> 63) & 1;
      }
    }
    return r;
  }
  bool is_zero(const E& a) const {
    for (int i = 0; i < N; i++)
      if (a[i]) return false;
    return true;
  }
  E neg(const E& a) const {
    if (is_zero(a)) return a;
    E r{};
    uint64_t br = 0;
    for (int i = 0; i < N; i++) {
      uint64_t d = (uint64_t)mod[i] - a[i] - br;
      r[i] = (uint32_t)d;
      br = (d >> 63) & 1;
    }
    return r;
  }
  E sub(const E& a, const E& b) const { return add(a, neg(b)); }
  E from_u64(uint64_t v) const {                      // canonical small integer -> Montgomery
    E r{};
    r[0] = (uint32_t)v;
    r[1] = (uint32_t)(v >> 32);
    E r2e{};
    memcpy(r2e.data(), r2, 4 * N);
    return mul(r, r2e);
  }
  E from_canonical(const uint32_t* limbs) const {
    E r{}, r2e{};
    memcpy(r.data(), limbs, 4 * N);
    memcpy(r2e.data(), r2, 4 * N);
    return mul(r, r2e);
  }
  E to_canonical(const E& a) const {                   // Montgomery -> canonical integer (into_repr)
    E o{};
    o[0] = 1;
    return mul(a, o);
  }
  E pow2k(E a, int k) const {                          // a^(2^k)
    for (int i = 0; i < k; i++) a = mul(a, a);
    return a;
  }
  E inverse(const E& a) const {                        // Fermat: a^(p-2)
    E r = one_();
    for (int bit = 32 * N - 1; bit >= 0; bit--) {
      r = mul(r, r);
      if ((pm2[bit >> 5] >> (bit & 31)) & 1) r = mul(r, a);
    }
    return r;
  }
  int cmp_canonical(const E& a, const E& b) const {    // a, b Montgomery; compares the canonical integers (Fp's Ord)
    E x = to_canonical(a), y = to_canonical(b);
    for (int i = N - 1; i >= 0; i--)
      if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
  }
};
inline HostField fr_field(int curve) {
  if (curve == ZKP_BN254)
    return {8, consts::Bn254Fr::MOD, consts::Bn254Fr::ONE, consts::Bn254Fr::R2, consts::Bn254Fr::PM2, consts::Bn254Fr::INV};
  return {8, consts::Bls381Fr::MOD, consts::Bls381Fr::ONE, consts::Bls381Fr::R2, consts::Bls381Fr::PM2, consts::Bls381Fr::INV};
}
inline HostField fq_field(int curve) {
  if (curve == ZKP_BN254)
    return {8, consts::Bn254Fq::MOD, consts::Bn254Fq::ONE, consts::Bn254Fq::R2, consts::Bn254Fq::PM2, consts::Bn254Fq::INV};
  return {12, consts::Bls381Fq::MOD, consts::Bls381Fq::ONE, consts::Bls381Fq::R2, consts::Bls381Fq::PM2, consts::Bls381Fq::INV};
}
using FrE = HostField::E;

// ---- host-side tail of a commitment: Jacobian (+ Jacobian) -> affine on the CPU (round 4).  The Jacobian MSM results are on the host
// anyway; a device launch for "add the blinding MSM, invert Z" is ONE lane running a 254-step Fermat chain (0.25-0.3 ms per launch, alone
// on the device, once per AHP round and twice per opening).  Here: a = 0 short Weierstrass formulas on HostField (ark's add_assign /
// double_in_place semantics: identity operands, equal operands, opposite operands) and ONE inversion for all points of a round
// (Montgomery's trick).  Coordinates are Fq Montgomery words, N per coordinate, exactly what the device kernels read / write.
struct HostJac {
  HostField::E x, y, z;
};
inline HostJac host_jac_load(const HostField& F, const uint64_t* p) {
  HostJac j{};
  memcpy(j.x.data(), p, 4 * F.N);
  memcpy(j.y.data(), reinterpret_cast<const uint32_t*>(p) + F.N, 4 * F.N);
  memcpy(j.z.data(), reinterpret_cast<const uint32_t*>(p) + 2 * F.N, 4 * F.N);
  return j;
}
inline HostJac host_jac_dbl(const HostField& F, const HostJac& p) {          // dbl-2009-l
  if (F.is_zero(p.z)) return p;
  const auto A = F.mul(p.x, p.x), B = F.mul(p.y, p.y), C = F.mul(B, B);
  auto t = F.add(p.x, B);
  t = F.sub(F.sub(F.mul(t, t), A), C);
  const auto D = F.add(t, t), E = F.add(F.add(A, A), A), Fq = F.mul(E, E);
  HostJac r;
  r.x = F.sub(Fq, F.add(D, D));
  auto c8 = F.add(C, C);
  c8 = F.add(c8, c8);
  c8 = F.add(c8, c8);
  r.y = F.sub(F.mul(E, F.sub(D, r.x)), c8);
  const auto yz = F.mul(p.y, p.z);
  r.z = F.add(yz, yz);
  return r;
}
inline HostJac host_jac_add(const HostField& F, const HostJac& p, const HostJac& q) {   // add-2007-bl
  if (F.is_zero(p.z)) return q;
  if (F.is_zero(q.z)) return p;
  const auto z1z1 = F.mul(p.z, p.z), z2z2 = F.mul(q.z, q.z);
  const auto u1 = F.mul(p.x, z2z2), u2 = F.mul(q.x, z1z1);
  const auto s1 = F.mul(F.mul(p.y, q.z), z2z2), s2 = F.mul(F.mul(q.y, p.z), z1z1);
  if (u1 == u2) {
    if (s1 == s2) return host_jac_dbl(F, p);
    HostJac inf{};
    inf.x = F.one_();
    inf.y = F.one_();
    return inf;
  }
  const auto h = F.sub(u2, u1);
  auto i = F.add(h, h);
  i = F.mul(i, i);
  const auto j = F.mul(h, i);
  auto rr = F.sub(s2, s1);
  rr = F.add(rr, rr);
  const auto v = F.mul(u1, i);
  HostJac r;
  r.x = F.sub(F.sub(F.mul(rr, rr), j), F.add(v, v));
  auto s1j = F.mul(s1, j);
  s1j = F.add(s1j, s1j);
  r.y = F.sub(F.mul(rr, F.sub(v, r.x)), s1j);
  auto zz = F.add(p.z, q.z);
  zz = F.sub(F.sub(F.mul(zz, zz), z1z1), z2z2);
  r.z = F.mul(zz, h);
  return r;
}
// k points -> affine words (N per coordinate, zeros for the identity) + identity flags; one field inversion in total
inline void host_into_affine(const HostField& F, const std::vector<HostJac>& pts, uint64_t* xy_out, size_t stride64, uint8_t* inf_out) {
  const size_t k = pts.size();
  std::vector<HostField::E> pref(k);
  HostField::E acc = F.one_();
  for (size_t i = 0; i < k; i++) {
    pref[i] = acc;
    if (!F.is_zero(pts[i].z)) acc = F.mul(acc, pts[i].z);
  }
  HostField::E inv = F.inverse(acc);
  for (size_t i = k; i-- > 0;) {
    uint32_t* o = reinterpret_cast<uint32_t*>(xy_out + i * stride64);
    memset(o, 0, 8 * F.N);
    inf_out[i] = F.is_zero(pts[i].z) ? 1 : 0;
    if (inf_out[i]) continue;
    const auto zi = F.mul(inv, pref[i]);
    inv = F.mul(inv, pts[i].z);
    const auto zi2 = F.mul(zi, zi);
    const auto ax = F.mul(pts[i].x, zi2), ay = F.mul(pts[i].y, F.mul(zi2, zi));
    memcpy(o, ax.data(), 4 * F.N);
    memcpy(o + F.N, ay.data(), 4 * F.N);
  }
}

// ---- Fq2 = Fq[u] / (u^2 + 1) (BN254 and BLS12-381 alike) on HostField elements
struct HostFq2 {
  HostField::E c0, c1;
};
inline HostFq2 fq2_mul(const HostField& F, const HostFq2& a, const HostFq2& b) {
  return {F.sub(F.mul(a.c0, b.c0), F.mul(a.c1, b.c1)), F.add(F.mul(a.c0, b.c1), F.mul(a.c1, b.c0))};
}
inline HostFq2 fq2_scale(const HostField& F, const HostFq2& a, const HostField::E& k) { return {F.mul(a.c0, k), F.mul(a.c1, k)}; }
inline bool fq2_is_zero(const HostField& F, const HostFq2& a) { return F.is_zero(a.c0) && F.is_zero(a.c1); }

// The three points of a Groth16 proof leave the device as XYZZ (x = X / ZZ, y = Y / ZZZ; ZZ == 0 <=> identity; Montgomery words,
// N per Fq coordinate, G2 coordinates (c0, c1)) and become affine here: 1 / (ZZ ZZZ) for A and C, conj / norm for B, ONE inversion.
//   a_xyzz, c_xyzz: 4 N words each; b_xyzz: 8 N words.  out: A (2N) | B (4N) | C (2N) words — ark's (x, y), zeros for the identity.
inline void groth16_points_into_affine(const HostField& F, const uint32_t* a_xyzz, const uint32_t* b_xyzz, const uint32_t* c_xyzz,
                                       uint32_t* out, uint8_t* inf_out) {
  using E = HostField::E;
  const int N = F.N;
  auto ld = [&](const uint32_t* p) {
    E e{};
    memcpy(e.data(), p, 4 * N);
    return e;
  };
  // denominators: d = ZZ * ZZZ
  const uint32_t* g1[2] = {a_xyzz, c_xyzz};
  E d1[2], zz1[2], zzz1[2];
  bool inf1[2];
  for (int k = 0; k < 2; k++) {
    zz1[k] = ld(g1[k] + 2 * N);
    zzz1[k] = ld(g1[k] + 3 * N);
    inf1[k] = F.is_zero(zz1[k]);
    d1[k] = inf1[k] ? F.one_() : F.mul(zz1[k], zzz1[k]);
  }
  const HostFq2 bzz{ld(b_xyzz + 4 * N), ld(b_xyzz + 5 * N)}, bzzz{ld(b_xyzz + 6 * N), ld(b_xyzz + 7 * N)};
  const bool infb = fq2_is_zero(F, bzz);
  const HostFq2 db = infb ? HostFq2{F.one_(), F.zero()} : fq2_mul(F, bzz, bzzz);
  const E nb = F.add(F.mul(db.c0, db.c0), F.mul(db.c1, db.c1));            // norm: (c0 + c1 u)(c0 - c1 u)
  // one inversion for d1[0], d1[1], nb
  const E p01 = F.mul(d1[0], d1[1]);
  E inv = F.inverse(F.mul(p01, nb));
  const E inb = F.mul(inv, p01);                                            // 1 / nb
  inv = F.mul(inv, nb);                                                     // 1 / (d0 d1)
  const E id1[2] = {F.mul(inv, d1[1]), F.mul(inv, d1[0])};
  memset(out, 0, 4 * 8 * N);
  for (int k = 0; k < 2; k++) {
    uint32_t* o = out + (k == 0 ? 0 : 6 * N);
    inf_out[k == 0 ? 0 : 2] = inf1[k] ? 1 : 0;
    if (inf1[k]) continue;
    const E x = F.mul(ld(g1[k]), F.mul(id1[k], zzz1[k])), y = F.mul(ld(g1[k] + N), F.mul(id1[k], zz1[k]));
    memcpy(o, x.data(), 4 * N);
    memcpy(o + N, y.data(), 4 * N);
  }
  inf_out[1] = infb ? 1 : 0;
  if (!infb) {
    const HostFq2 idb{F.mul(db.c0, inb), F.neg(F.mul(db.c1, inb))};       // 1 / db
    const HostFq2 izz = fq2_mul(F, idb, bzzz), izzz = fq2_mul(F, idb, bzz);
    const HostFq2 x = fq2_mul(F, HostFq2{ld(b_xyzz), ld(b_xyzz + N)}, izz);
    const HostFq2 y = fq2_mul(F, HostFq2{ld(b_xyzz + 2 * N), ld(b_xyzz + 3 * N)}, izzz);
    uint32_t* o = out + 2 * N;
    memcpy(o, x.c0.data(), 4 * N);
    memcpy(o + N, x.c1.data(), 4 * N);
    memcpy(o + 2 * N, y.c0.data(), 4 * N);
    memcpy(o + 3 * N, y.c1.data(), 4 * N);
  }
}

}  // namespace hostf
}  // namespace zkp
