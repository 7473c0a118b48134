// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or used as a fallback for the product.
// Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg ("kind": "port").
//
// C++17 CPU restatement of the algorithms the reference's Groth16 prover executes:
//   * ark-ff 0.2  Fp256/Fp384 Montgomery arithmetic (64-bit limbs, __int128)       [third-party, not vendored]
//   * ark-ec 0.2  short_weierstrass_jacobian add_assign_mixed / add_assign / double_in_place and
//                 msm::VariableBaseMSM::multi_scalar_mul (window rule c = 3 | ln_without_floats(n)+2,
//                 zero-skip, one fast-path in window 0, 2^c-1 Jacobian buckets, running sum, Horner;
//                 one task per window == rayon `parallel` feature)                  [third-party, not vendored]
//   * ark-poly 0.2 Radix2EvaluationDomain fft/ifft/coset_fft/coset_ifft (in place, natural order)
//   * /root/reference/groth16/src/r1cs_to_qap.rs:16-52,113-172   evaluate_constraint, witness_map
//   * /root/reference/groth16/src/prover.rs:124-211,213-228     create_proof, calculate_coeff
// PARITY UNPINNED by reference tests (the reference holds no golden vectors for this path and cannot be
// built here: no cargo/rustc, crates not on disk).  Pinned instead by oracle/pyref (big-int ground truth,
// pairing-verified) on every fixture in tests/golden/ — see oracle/README.md.
//
// Build: g++ -O3 -march=native -std=c++17 -shared -fPIC -pthread oracle/cpu/zkp_oracle.cpp -o oracle/build/libzkp_oracle.so
#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/zkp_accel.h"

typedef unsigned __int128 u128;

#include "field_constants64.inc"

// ------------------------------------------------------------------------------------------------ Fp
template <class P>
struct Fp {
  static constexpr int N = P::N;
  uint64_t v[N];
  static Fp zero() { Fp r; for (int i = 0; i < N; i++) r.v[i] = 0; return r; }
  static Fp one() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::ONE[i]; return r; }
  static Fp from_limbs(const uint64_t* p) { Fp r; memcpy(r.v, p, 8 * N); return r; }
  void to_limbs(uint64_t* p) const { memcpy(p, v, 8 * N); }
  bool is_zero() const { uint64_t o = 0; for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
  bool operator==(const Fp& b) const { return memcmp(v, b.v, 8 * N) == 0; }
  bool operator!=(const Fp& b) const { return !(*this == b); }
  static bool geq_mod(const uint64_t* a) {
    for (int i = N - 1; i >= 0; i--) { if (a[i] > P::MOD[i]) return true; if (a[i] < P::MOD[i]) return false; }
    return true;
  }
  static void sub_mod(uint64_t* a) {
    uint64_t borrow = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - P::MOD[i] - borrow; a[i] = (uint64_t)d; borrow = (uint64_t)(d >> 127); }
  }
  Fp operator+(const Fp& b) const {
    Fp r; uint64_t c = 0;
    for (int i = 0; i < N; i++) { u128 s = (u128)v[i] + b.v[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    if (geq_mod(r.v)) sub_mod(r.v);
    return r;
  }
  Fp operator-(const Fp& b) const {
    Fp r; uint64_t borrow = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)v[i] - b.v[i] - borrow; r.v[i] = (uint64_t)d; borrow = (uint64_t)(d >> 127); }
    if (borrow) { uint64_t c = 0; for (int i = 0; i < N; i++) { u128 s = (u128)r.v[i] + P::MOD[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
    return r;
  }
  Fp neg() const { return is_zero() ? *this : zero() - *this; }
  Fp dbl() const { return *this + *this; }
  Fp operator*(const Fp& b) const {   // CIOS, no-carry (all moduli here have a spare top bit)
    uint64_t t[N + 1];
    for (int i = 0; i <= N; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
      uint64_t c = 0;
      for (int j = 0; j < N; j++) { u128 x = (u128)v[j] * b.v[i] + t[j] + c; t[j] = (uint64_t)x; c = (uint64_t)(x >> 64); }
      t[N] += c;
      uint64_t m = t[0] * P::INV;
      c = (uint64_t)(((u128)m * P::MOD[0] + t[0]) >> 64);
      for (int j = 1; j < N; j++) { u128 x = (u128)m * P::MOD[j] + t[j] + c; t[j - 1] = (uint64_t)x; c = (uint64_t)(x >> 64); }
      u128 x = (u128)t[N] + c; t[N - 1] = (uint64_t)x; t[N] = (uint64_t)(x >> 64);
    }
    Fp r; for (int i = 0; i < N; i++) r.v[i] = t[i];
    if (geq_mod(r.v)) sub_mod(r.v);
    return r;
  }
  Fp sqr() const { return *this * *this; }
  Fp pow(const uint64_t* e, int n) const {
    Fp r = one();
    for (int i = n - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { r = r.sqr(); if ((e[i] >> b) & 1) r = r * *this; }
    return r;
  }
  Fp pow64(uint64_t e) const { return pow(&e, 1); }
  Fp inv() const {
    uint64_t e[N]; for (int i = 0; i < N; i++) e[i] = P::MOD[i];
    e[0] -= 2;   // moduli are odd and > 2 in limb 0 => no borrow
    return pow(e, N);
  }
  Fp from_mont() const { Fp o = zero(); o.v[0] = 1; return *this * o; }     // into_repr()
  Fp to_mont() const { Fp r2; for (int i = 0; i < N; i++) r2.v[i] = P::R2[i]; return *this * r2; }
};

template <class P>
struct Fp2 {
  using B = Fp<P>;
  static constexpr int N = 2 * P::N;
  B c0, c1;
  static Fp2 zero() { return {B::zero(), B::zero()}; }
  static Fp2 one() { return {B::one(), B::zero()}; }
  static Fp2 from_limbs(const uint64_t* p) { return {B::from_limbs(p), B::from_limbs(p + P::N)}; }
  void to_limbs(uint64_t* p) const { c0.to_limbs(p); c1.to_limbs(p + P::N); }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool operator!=(const Fp2& o) const { return !(*this == o); }
  Fp2 operator+(const Fp2& o) const { return {c0 + o.c0, c1 + o.c1}; }
  Fp2 operator-(const Fp2& o) const { return {c0 - o.c0, c1 - o.c1}; }
  Fp2 neg() const { return {c0.neg(), c1.neg()}; }
  Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  Fp2 operator*(const Fp2& o) const {
    B v0 = c0 * o.c0, v1 = c1 * o.c1;
    return {v0 - v1, (c0 + c1) * (o.c0 + o.c1) - v0 - v1};
  }
  Fp2 sqr() const { B t = c0 * c1; return {(c0 + c1) * (c0 - c1), t.dbl()}; }
  Fp2 inv() const { B n = (c0.sqr() + c1.sqr()).inv(); return {c0 * n, (c1 * n).neg()}; }
};

// ------------------------------------------------------------------------------------------------ curve
template <class F>
struct Aff { F x, y; bool inf; };

template <class F>
struct Jac {
  F x, y, z;
  static Jac zero() { return {F::zero(), F::one(), F::zero()}; }
  bool is_zero() const { return z.is_zero(); }
  void dbl_in_place() {            // dbl-2009-l (a = 0)
    if (is_zero()) return;
    F a = x.sqr(), b = y.sqr(), c = b.sqr();
    F d = ((x + b).sqr() - a - c).dbl();
    F e = a + a.dbl();
    F f = e.sqr();
    F z3 = (z * y).dbl();
    F x3 = f - d.dbl();
    F y3 = e * (d - x3) - c.dbl().dbl().dbl();
    x = x3; y = y3; z = z3;
  }
  void add_mixed(const Aff<F>& o) {   // madd-2007-bl ; ark add_assign_mixed semantics
    if (o.inf) return;
    if (is_zero()) { x = o.x; y = o.y; z = F::one(); return; }
    F z1z1 = z.sqr();
    F u2 = o.x * z1z1;
    F s2 = (o.y * z) * z1z1;
    if (x == u2 && y == s2) { dbl_in_place(); return; }
    F h = u2 - x;
    F hh = h.sqr();
    F i = hh.dbl().dbl();
    F j = h * i;
    F r = (s2 - y).dbl();
    F v = x * i;
    F x3 = r.sqr() - j - v.dbl();
    F y3 = r * (v - x3) - (y * j).dbl();
    F z3 = (z + h).sqr() - z1z1 - hh;
    x = x3; y = y3; z = z3;
  }
  void add(const Jac& o) {            // add-2007-bl
    if (is_zero()) { *this = o; return; }
    if (o.is_zero()) return;
    F z1z1 = z.sqr(), z2z2 = o.z.sqr();
    F u1 = x * z2z2, u2 = o.x * z1z1;
    F s1 = y * o.z * z2z2, s2 = o.y * z * z1z1;
    if (u1 == u2 && s1 == s2) { dbl_in_place(); return; }
    F h = u2 - u1;
    F i = h.dbl().sqr();
    F j = h * i;
    F r = (s2 - s1).dbl();
    F v = u1 * i;
    F x3 = r.sqr() - j - v.dbl();
    F y3 = r * (v - x3) - (s1 * j).dbl();
    F z3 = ((z + o.z).sqr() - z1z1 - z2z2) * h;
    x = x3; y = y3; z = z3;
  }
  Jac neg() const { return {x, y.neg(), z}; }
  Aff<F> into_affine() const {
    if (is_zero()) return {F::zero(), F::zero(), true};
    F zi = z.inv(), zi2 = zi.sqr();
    return {x * zi2, y * zi2 * zi, false};
  }
};

template <class F>
static Jac<F> scalar_mul(const Jac<F>& p, const uint64_t* k, int limbs) {
  Jac<F> r = Jac<F>::zero();
  for (int i = limbs - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { r.dbl_in_place(); if ((k[i] >> b) & 1) r.add(p); }
  return r;
}

static int ark_log2(size_t x) { if (x <= 1) return 0; int l = 0; size_t v = x - 1; while (v) { v >>= 1; l++; } return l; }
static int ark_window(size_t n) { return n < 32 ? 3 : ark_log2(n) * 69 / 100 + 2; }

static bool big_is_zero(const uint64_t* s) { return (s[0] | s[1] | s[2] | s[3]) == 0; }
static bool big_is_one(const uint64_t* s) { return s[0] == 1 && (s[1] | s[2] | s[3]) == 0; }
static uint64_t big_window(const uint64_t* s, int start, int c) {   // (s >> start) % 2^c  (ark: divn + limb0 % 2^c)
  if (start >= 256) return 0;
  int limb = start >> 6, sh = start & 63;
  u128 two = s[limb];
  if (limb + 1 < 4) two |= (u128)s[limb + 1] << 64;
  return (uint64_t)(two >> sh) & ((1ULL << c) - 1);
}

// ark-ec 0.2 VariableBaseMSM::multi_scalar_mul.  scalars: canonical 4x u64.  threads: window-parallel.
template <class F>
static Jac<F> msm_pippenger(const Aff<F>* bases, const uint64_t* scalars, size_t n, int num_bits, int threads) {
  const int c = ark_window(n);
  std::vector<int> starts;
  for (int w = 0; w < num_bits; w += c) starts.push_back(w);
  std::vector<Jac<F>> window_sums(starts.size());
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      size_t wi = next.fetch_add(1);
      if (wi >= starts.size()) break;
      const int w_start = starts[wi];
      Jac<F> res = Jac<F>::zero();
      std::vector<Jac<F>> buckets(((size_t)1 << c) - 1, Jac<F>::zero());
      for (size_t i = 0; i < n; i++) {
        const uint64_t* s = scalars + 4 * i;
        if (big_is_zero(s)) continue;
        if (big_is_one(s)) { if (w_start == 0) res.add_mixed(bases[i]); continue; }
        uint64_t d = big_window(s, w_start, c);
        if (d != 0) buckets[d - 1].add_mixed(bases[i]);
      }
      Jac<F> running = Jac<F>::zero();
      for (size_t b = buckets.size(); b-- > 0;) { running.add(buckets[b]); res.add(running); }
      window_sums[wi] = res;
    }
  };
  int nt = std::max(1, std::min<int>(threads, (int)starts.size()));
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; t++) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  Jac<F> total = Jac<F>::zero();
  for (size_t wi = window_sums.size(); wi-- > 1;) {
    total.add(window_sums[wi]);
    for (int k = 0; k < c; k++) total.dbl_in_place();
  }
  total.add(window_sums[0]);
  return total;
}

// ------------------------------------------------------------------------------------------------ NTT
template <class P>
struct Domain {
  using F = Fp<P>;
  int log_n; size_t n;
  F w, w_inv, n_inv, g, g_inv;
  explicit Domain(int lg) : log_n(lg), n((size_t)1 << lg) {
    w = F::from_limbs(P::ROOT);
    for (int i = 0; i < P::TWO_ADICITY - lg; i++) w = w.sqr();
    w_inv = w.inv();
    F nn = F::zero(); nn.v[lg / 64] = 1ULL << (lg % 64);
    n_inv = nn.to_mont().inv();
    g = F::from_limbs(P::GEN);
    g_inv = g.inv();
  }
  // ark-poly serial_radix2_fft: bit-reverse, then log n DIT stages with running twiddle.
  // Threading (stand-in for ark's rayon `parallel` feature): T = 2^t threads; the first log n - t stages touch
  // only a thread's own contiguous n/T chunk (one spawn for all of them); the last t stages split the j-range.
  static void stage_range(F* a, size_t k, size_t m, size_t j0, size_t j1, const F& w_m) {
    F wj = j0 ? w_m.pow64(j0) : F::one();
    for (size_t j = j0; j < j1; j++) {
      F t = a[k + j + m] * wj;
      a[k + j + m] = a[k + j] - t;
      a[k + j] = a[k + j] + t;
      wj = wj * w_m;
    }
  }
  void transform(F* a, const F& omega, int threads) const {
    int t = 0;
    while ((2 << t) <= threads && t + 1 <= 6 && log_n - (t + 1) >= 10) t++;   // <= 64 threads, chunks >= 1024
    const size_t T = (size_t)1 << t;
    auto par = [&](const std::function<void(size_t)>& fn) {
      if (T == 1) { fn(0); return; }
      std::vector<std::thread> pool;
      for (size_t i = 1; i < T; i++) pool.emplace_back(fn, i);
      fn(0);
      for (auto& th : pool) th.join();
    };
    // bit reversal
    par([&](size_t ti) {
      size_t per = n / T;
      for (size_t k = ti * per; k < (ti + 1) * per; k++) {
        size_t rk = 0; for (int b = 0; b < log_n; b++) rk |= ((k >> b) & 1) << (log_n - 1 - b);
        if (k < rk) std::swap(a[k], a[rk]);
      }
    });
    const int local_stages = log_n - t;
    std::vector<F> wm(log_n);
    for (int s = 0; s < log_n; s++) wm[s] = omega.pow64(n >> (s + 1));
    par([&](size_t ti) {
      size_t per = n / T, base = ti * per;
      for (int s = 0; s < local_stages; s++) {
        size_t m = (size_t)1 << s;
        for (size_t k = base; k < base + per; k += 2 * m) stage_range(a, k, m, 0, m, wm[s]);
      }
    });
    for (int s = local_stages; s < log_n; s++) {
      size_t m = (size_t)1 << s;
      size_t blocks = n / (2 * m);               // < T
      size_t split = T / blocks;                 // threads per block
      par([&](size_t ti) {
        size_t blk = ti / split, part = ti % split;
        size_t per = m / split;
        stage_range(a, blk * 2 * m, m, part * per, (part + 1) * per, wm[s]);
      });
    }
  }
  void distribute_powers(F* a, const F& gg, const F& c0) const { F p = c0; for (size_t i = 0; i < n; i++) { a[i] = a[i] * p; p = p * gg; } }
  void fft(F* a, int th) const { transform(a, w, th); }
  void ifft(F* a, int th) const { transform(a, w_inv, th); for (size_t i = 0; i < n; i++) a[i] = a[i] * n_inv; }
  void coset_fft(F* a, int th) const { distribute_powers(a, g, F::one()); fft(a, th); }
  void coset_ifft(F* a, int th) const { ifft(a, th); distribute_powers(a, g_inv, F::one()); }
  F vanishing_on_coset_inv() const { F t = g; for (int i = 0; i < log_n; i++) t = t.sqr(); return (t - F::one()).inv(); }
};

// ------------------------------------------------------------------------------------------------ Groth16
template <class FrP>
static void eval_rows(const zkp_csr& m, const Fp<FrP>* z, size_t nc, Fp<FrP>* out, int threads) {
  using F = Fp<FrP>;
  auto body = [&](size_t i0, size_t i1) {
    const F one = F::one();
    for (size_t i = i0; i < i1; i++) {
      F acc = F::zero();
      for (uint32_t k = m.row_ptr[i]; k < m.row_ptr[i + 1]; k++) {
        F cf = F::from_limbs(m.coeff + 4 * (size_t)k);
        const F& v = z[m.col[k]];
        acc = (cf == one) ? acc + v : acc + v * cf;       // r1cs_to_qap.rs:39-43
      }
      out[i] = acc;
    }
  };
  std::vector<std::thread> pool;
  size_t per = (nc + threads - 1) / std::max(1, threads);
  for (int t = 0; t < threads; t++) { size_t i0 = t * per, i1 = std::min(nc, i0 + per); if (i0 < i1) pool.emplace_back(body, i0, i1); }
  for (auto& t : pool) t.join();
}

// r1cs_to_qap.rs:113-172
template <class FrP>
static std::vector<Fp<FrP>> witness_map(const zkp_groth16_pk_desc* d, const Fp<FrP>* z, int threads) {
  using F = Fp<FrP>;
  size_t nc = d->num_constraints, ni = d->num_inputs;
  int lg = ark_log2(nc + ni);
  Domain<FrP> dom(lg);
  size_t N = dom.n;
  std::vector<F> a(N, F::zero()), b(N, F::zero());
  eval_rows<FrP>(d->at, z, nc, a.data(), threads);
  eval_rows<FrP>(d->bt, z, nc, b.data(), threads);
  for (size_t i = 0; i < ni; i++) a[nc + i] = z[i];
  dom.ifft(a.data(), threads); dom.ifft(b.data(), threads);
  dom.coset_fft(a.data(), threads); dom.coset_fft(b.data(), threads);
  for (size_t i = 0; i < N; i++) a[i] = a[i] * b[i];
  std::vector<F>().swap(b);
  std::vector<F> c(N, F::zero());
  eval_rows<FrP>(d->ct, z, nc, c.data(), threads);
  dom.ifft(c.data(), threads); dom.coset_fft(c.data(), threads);
  F zi = dom.vanishing_on_coset_inv();
  for (size_t i = 0; i < N; i++) a[i] = (a[i] - c[i]) * zi;
  dom.coset_ifft(a.data(), threads);
  return a;
}

template <class F>
static std::vector<Aff<F>> load_points(const uint64_t* xy, const uint8_t* inf, size_t n) {
  std::vector<Aff<F>> out(n);
  const int L = F::N;   // u64 limbs per coordinate
  for (size_t i = 0; i < n; i++) {
    out[i].inf = inf && inf[i];
    out[i].x = F::from_limbs(xy + (2 * i) * L);
    out[i].y = F::from_limbs(xy + (2 * i + 1) * L);
  }
  return out;
}

template <class F>
static void store_affine(const Aff<F>& a, uint64_t* out, uint8_t* inf) {
  if (a.inf) { memset(out, 0, 16 * F::N); *inf = 1; return; }
  a.x.to_limbs(out); a.y.to_limbs(out + F::N); *inf = 0;
}

// prover.rs:124-211 (after synthesis).  r, s, z: Montgomery Fr.
template <class FrP, class FqP>
static void create_proof(const zkp_groth16_pk_desc* d, const uint64_t* z_limbs, const uint64_t* r_l, const uint64_t* s_l,
                         int threads, uint64_t* proof_out, uint8_t* inf_out, double* phase_ms, uint64_t* h_out = nullptr) {
  using Fr = Fp<FrP>; using G1F = Fp<FqP>; using G2F = Fp2<FqP>;
  const size_t ni = d->num_inputs, na = d->num_aux, nz = ni + na;
  std::vector<Fr> z(nz);
  for (size_t i = 0; i < nz; i++) z[i] = Fr::from_limbs(z_limbs + 4 * i);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<Fr> h = witness_map<FrP>(d, z.data(), threads);
  auto t1 = std::chrono::steady_clock::now();
  if (h_out) memcpy(h_out, h.data(), h.size() * 32);          // the quotient this proof was made from (Montgomery), for the caller's own checks
  // into_repr (prover.rs:150-161)
  std::vector<uint64_t> assignment(4 * (nz - 1)), hrep(4 * h.size());
  for (size_t i = 1; i < nz; i++) z[i].from_mont().to_limbs(&assignment[4 * (i - 1)]);
  for (size_t i = 0; i < h.size(); i++) h[i].from_mont().to_limbs(&hrep[4 * i]);
  const uint64_t* aux_rep = assignment.data() + 4 * (ni - 1);
  Fr r = Fr::from_limbs(r_l), s = Fr::from_limbs(s_l);
  Fr rr = r.from_mont(), sr = s.from_mont();
  auto a_q = load_points<G1F>(d->a_query, d->a_inf, d->a_len);
  auto b1_q = load_points<G1F>(d->b_g1_query, d->b_g1_inf, d->b_g1_len);
  auto b2_q = load_points<G2F>(d->b_g2_query, d->b_g2_inf, d->b_g2_len);
  auto h_q = load_points<G1F>(d->h_query, d->h_inf, d->h_len);
  auto l_q = load_points<G1F>(d->l_query, d->l_inf, d->l_len);
  Aff<G1F> alpha = load_points<G1F>(d->alpha_g1, nullptr, 1)[0], beta1 = load_points<G1F>(d->beta_g1, nullptr, 1)[0],
           delta1 = load_points<G1F>(d->delta_g1, nullptr, 1)[0];
  Aff<G2F> beta2 = load_points<G2F>(d->beta_g2, nullptr, 1)[0], delta2 = load_points<G2F>(d->delta_g2, nullptr, 1)[0];
  const int bits = FrP::BITS;
  auto t2 = std::chrono::steady_clock::now();
  auto to_j1 = [](const Aff<G1F>& p) { Jac<G1F> j = Jac<G1F>::zero(); j.add_mixed(p); return j; };
  auto to_j2 = [](const Aff<G2F>& p) { Jac<G2F> j = Jac<G2F>::zero(); j.add_mixed(p); return j; };
  // calculate_coeff (prover.rs:213-228)
  auto coeff1 = [&](Jac<G1F> initial, const std::vector<Aff<G1F>>& q, const Aff<G1F>& vk) {
    Jac<G1F> acc = msm_pippenger<G1F>(q.data() + 1, assignment.data(), std::min(q.size() - 1, nz - 1), bits, threads);
    Jac<G1F> res = initial; res.add_mixed(q[0]); res.add(acc); res.add_mixed(vk); return res;
  };
  Jac<G1F> g_a = coeff1(scalar_mul(to_j1(delta1), rr.v, 4), a_q, alpha);
  auto t3 = std::chrono::steady_clock::now();
  Jac<G1F> g1_b = Jac<G1F>::zero();
  if (!r.is_zero()) g1_b = coeff1(scalar_mul(to_j1(delta1), sr.v, 4), b1_q, beta1);
  auto t4 = std::chrono::steady_clock::now();
  Jac<G2F> g2_b;
  {
    Jac<G2F> acc = msm_pippenger<G2F>(b2_q.data() + 1, assignment.data(), std::min(b2_q.size() - 1, nz - 1), bits, threads);
    g2_b = scalar_mul(to_j2(delta2), sr.v, 4); g2_b.add_mixed(b2_q[0]); g2_b.add(acc); g2_b.add_mixed(beta2);
  }
  auto t5 = std::chrono::steady_clock::now();
  Jac<G1F> h_acc = msm_pippenger<G1F>(h_q.data(), hrep.data(), std::min(h_q.size(), h.size()), bits, threads);
  auto t6 = std::chrono::steady_clock::now();
  Jac<G1F> l_acc = msm_pippenger<G1F>(l_q.data(), aux_rep, std::min(l_q.size(), na), bits, threads);
  auto t7 = std::chrono::steady_clock::now();
  Jac<G1F> g_c = scalar_mul(g_a, sr.v, 4);
  g_c.add(scalar_mul(g1_b, rr.v, 4));
  Jac<G1F> rsd = scalar_mul(scalar_mul(to_j1(delta1), rr.v, 4), sr.v, 4);
  g_c.add(rsd.neg());
  g_c.add(l_acc);
  g_c.add(h_acc);
  const int f = FqP::N;
  store_affine(g_a.into_affine(), proof_out, inf_out + 0);
  store_affine(g2_b.into_affine(), proof_out + 2 * f, inf_out + 1);
  store_affine(g_c.into_affine(), proof_out + 6 * f, inf_out + 2);
  auto t8 = std::chrono::steady_clock::now();
  if (phase_ms) {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    phase_ms[0] = ms(t0, t1); phase_ms[1] = ms(t2, t3); phase_ms[2] = ms(t3, t4); phase_ms[3] = ms(t4, t5);
    phase_ms[4] = ms(t5, t6); phase_ms[5] = ms(t6, t7); phase_ms[6] = ms(t7, t8); phase_ms[7] = ms(t0, t8);
  }
}

template <class F>
static void msm_entry(const uint64_t* xy, const uint8_t* inf, const uint64_t* scalars, size_t n, int bits, int threads,
                      uint64_t* out_xyz) {
  auto pts = load_points<F>(xy, inf, n);
  Jac<F> r = msm_pippenger<F>(pts.data(), scalars, n, bits, threads);
  r.x.to_limbs(out_xyz); r.y.to_limbs(out_xyz + F::N); r.z.to_limbs(out_xyz + 2 * F::N);
}

template <class P>
static void ntt_entry(uint64_t* data, int log_n, int op, int threads) {
  Domain<P> d(log_n);
  auto* a = reinterpret_cast<Fp<P>*>(data);
  switch (op) {
    case ZKP_NTT_FFT: d.fft(a, threads); break;
    case ZKP_NTT_IFFT: d.ifft(a, threads); break;
    case ZKP_NTT_COSET_FFT: d.coset_fft(a, threads); break;
    case ZKP_NTT_COSET_IFFT: d.coset_ifft(a, threads); break;
  }
}

#include "marlin_oracle.inc"

extern "C" {

int oracle_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

// VariableBaseMSM::multi_scalar_mul; scalars canonical; out = Jacobian (X,Y,Z) Montgomery
int oracle_msm(int curve, int group, const uint64_t* xy, const uint8_t* inf, const uint64_t* scalars, size_t n,
               int threads, uint64_t* out_xyz) {
  if (curve == ZKP_BN254 && group == 1) msm_entry<Fp<Bn254FqP>>(xy, inf, scalars, n, Bn254FrP::BITS, threads, out_xyz);
  else if (curve == ZKP_BN254 && group == 2) msm_entry<Fp2<Bn254FqP>>(xy, inf, scalars, n, Bn254FrP::BITS, threads, out_xyz);
  else if (curve == ZKP_BLS12_381 && group == 1) msm_entry<Fp<Bls381FqP>>(xy, inf, scalars, n, Bls381FrP::BITS, threads, out_xyz);
  else if (curve == ZKP_BLS12_381 && group == 2) msm_entry<Fp2<Bls381FqP>>(xy, inf, scalars, n, Bls381FrP::BITS, threads, out_xyz);
  else return -1;
  return 0;
}

int oracle_ntt(int curve, uint64_t* data, int log_n, int op, int threads) {
  if (curve == ZKP_BN254) { if (log_n > 28) return ZKP_ERR_DOMAIN_TOO_LARGE; ntt_entry<Bn254FrP>(data, log_n, op, threads); }
  else if (curve == ZKP_BLS12_381) { if (log_n > 32) return ZKP_ERR_DOMAIN_TOO_LARGE; ntt_entry<Bls381FrP>(data, log_n, op, threads); }
  else return -1;
  return 0;
}

int oracle_witness_map(const zkp_groth16_pk_desc* d, const uint64_t* z, int threads, uint64_t* h_out) {
  if (d->curve == ZKP_BN254) {
    std::vector<Fp<Bn254FrP>> zz(d->num_inputs + (size_t)d->num_aux);
    memcpy(zz.data(), z, zz.size() * 32);
    auto h = witness_map<Bn254FrP>(d, zz.data(), threads);
    memcpy(h_out, h.data(), h.size() * 32);
  } else {
    std::vector<Fp<Bls381FrP>> zz(d->num_inputs + (size_t)d->num_aux);
    memcpy(zz.data(), z, zz.size() * 32);
    auto h = witness_map<Bls381FrP>(d, zz.data(), threads);
    memcpy(h_out, h.data(), h.size() * 32);
  }
  return 0;
}

// create_proof; phase_ms[8] (optional): witness_map, MSM A, B1, B2, H, L, assemble, total
int oracle_groth16_prove(const zkp_groth16_pk_desc* d, const uint64_t* z, const uint64_t* r, const uint64_t* s,
                         int threads, uint64_t* proof_out, uint8_t* inf_out, double* phase_ms) {
  if (d->curve == ZKP_BN254) create_proof<Bn254FrP, Bn254FqP>(d, z, r, s, threads, proof_out, inf_out, phase_ms);
  else if (d->curve == ZKP_BLS12_381) create_proof<Bls381FrP, Bls381FqP>(d, z, r, s, threads, proof_out, inf_out, phase_ms);
  else return -1;
  return 0;
}

// sum_i a_i * b_i over Fr (Montgomery in, Montgomery out): the trapdoor-in-the-exponent checks of the full-size tests
// (sum_i z_i a_i(tau) over 2^24 terms) without Python big-int loops
int oracle_fr_dot(int curve, const uint64_t* a, const uint64_t* b, size_t n, int threads, uint64_t* out) {
  auto run = [&](auto tag) {
    using F = Fp<decltype(tag)>;
    int T = std::max(1, std::min(threads, 64));
    std::vector<F> part(T, F::zero());
    size_t per = (n + T - 1) / T;
    auto body = [&](int t) {
      F acc = F::zero();
      for (size_t i = std::min(n, t * per), e = std::min(n, i + per); i < e; i++) acc = acc + F::from_limbs(a + 4 * i) * F::from_limbs(b + 4 * i);
      part[t] = acc;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back(body, t);
    body(0);
    for (auto& th : pool) th.join();
    F acc = F::zero();
    for (auto& p : part) acc = acc + p;
    acc.to_limbs(out);
  };
  if (curve == ZKP_BN254) run(Bn254FrP{});
  else if (curve == ZKP_BLS12_381) run(Bls381FrP{});
  else return -1;
  return 0;
}

// out[i] = first * base^i (Montgomery), i < n
int oracle_fr_powers(int curve, const uint64_t* base, const uint64_t* first, size_t n, int threads, uint64_t* out) {
  auto run = [&](auto tag) {
    using F = Fp<decltype(tag)>;
    const F b = F::from_limbs(base), f0 = F::from_limbs(first);
    int T = std::max(1, std::min(threads, 64));
    size_t per = (n + T - 1) / T;
    auto body = [&](int t) {
      size_t i0 = std::min(n, t * per), i1 = std::min(n, i0 + per);
      if (i0 >= i1) return;
      F p = f0 * b.pow64(i0);
      for (size_t i = i0; i < i1; i++) { p.to_limbs(out + 4 * i); p = p * b; }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back(body, t);
    body(0);
    for (auto& th : pool) th.join();
  };
  if (curve == ZKP_BN254) run(Bn254FrP{});
  else if (curve == ZKP_BLS12_381) run(Bls381FrP{});
  else return -1;
  return 0;
}

// create_proof that also hands back h = witness_map(z) (domain_size Fr, Montgomery): one pass for tests that compare both
int oracle_groth16_prove_h(const zkp_groth16_pk_desc* d, const uint64_t* z, const uint64_t* r, const uint64_t* s, int threads,
                           uint64_t* proof_out, uint8_t* inf_out, double* phase_ms, uint64_t* h_out) {
  if (d->curve == ZKP_BN254) create_proof<Bn254FrP, Bn254FqP>(d, z, r, s, threads, proof_out, inf_out, phase_ms, h_out);
  else if (d->curve == ZKP_BLS12_381) create_proof<Bls381FrP, Bls381FqP>(d, z, r, s, threads, proof_out, inf_out, phase_ms, h_out);
  else return -1;
  return 0;
}

// k_i * P (affine Montgomery out) — for building keys on CPU-only boxes in the `not gpu` tests
int oracle_fixed_base_mul(int curve, int group, const uint64_t* base_xy, const uint64_t* scalars, size_t n,
                          uint64_t* out_xy, uint8_t* out_inf) {
  auto run = [&](auto tag) {
    using F = decltype(tag);
    Aff<F> b = load_points<F>(base_xy, nullptr, 1)[0];
    Jac<F> jb = Jac<F>::zero(); jb.add_mixed(b);
    for (size_t i = 0; i < n; i++) {
      Aff<F> a = scalar_mul(jb, scalars + 4 * i, 4).into_affine();
      store_affine(a, out_xy + 2 * F::N * i, out_inf + i);
    }
  };
  if (curve == ZKP_BN254 && group == 1) run(Fp<Bn254FqP>{});
  else if (curve == ZKP_BN254 && group == 2) run(Fp2<Bn254FqP>{});
  else if (curve == ZKP_BLS12_381 && group == 1) run(Fp<Bls381FqP>{});
  else if (curve == ZKP_BLS12_381 && group == 2) run(Fp2<Bls381FqP>{});
  else return -1;
  return 0;
}

// ------------------------------------------------------------------------------------------------ Marlin (marlin_oracle.inc)
void* oracle_marlin_new(const oracle_marlin_desc* d, int threads) {
  try {
    auto* h = new MarlinHandle{d->curve};
    if (d->curve == ZKP_BN254) h->bn = new MarlinOracle<Bn254FrP, Bn254FqP>(d, threads);
    else if (d->curve == ZKP_BLS12_381) h->bls = new MarlinOracle<Bls381FrP, Bls381FqP>(d, threads);
    else { delete h; return nullptr; }
    return h;
  } catch (const std::exception& e) { fprintf(stderr, "oracle marlin: %s\n", e.what()); return nullptr; }
}
void oracle_marlin_free(void* hv) { delete (MarlinHandle*)hv; }
// out: |X|, |H|, |K|, |B|, max_degree, num_non_zeros, n (square dimension), padding variables
int oracle_marlin_info(void* hv, uint64_t* out) {
  auto* h = (MarlinHandle*)hv;
  MARLIN_DISPATCH(h, (out[0] = o->dx->n, out[1] = o->dh->n, out[2] = o->dk->n, out[3] = o->db->n, out[4] = o->max_degree,
                      out[5] = o->nnz, out[6] = o->n, out[7] = o->pad_aux));
  return 0;
}
int oracle_marlin_set_options(void* hv, int threads, int concurrent_commits) {
  auto* h = (MarlinHandle*)hv;
  MARLIN_DISPATCH(h, (o->threads = threads > 0 ? threads : 1, o->concurrent_commits = concurrent_commits != 0));
  return 0;
}
// committer key: powers_of_g / powers_of_gamma_g, affine Montgomery (an input, as `ipk.committer_key` is to create_random_proof)
int oracle_marlin_set_srs(void* hv, const uint64_t* g_xy, const uint8_t* g_inf, size_t ng, const uint64_t* gg_xy,
                          const uint8_t* gg_inf, size_t ngg) {
  auto* h = (MarlinHandle*)hv;
  MARLIN_DISPATCH(h, o->set_srs(g_xy, g_inf, ng, gg_xy, gg_inf, ngg));
  return 0;
}
int oracle_marlin_index_commit(void* hv, uint64_t* xy, uint8_t* inf) { auto* h = (MarlinHandle*)hv; MARLIN_DISPATCH(h, o->index_commit(xy, inf)); return 0; }
int oracle_marlin_round1(void* hv, const uint64_t* x, const uint64_t* w, size_t nw, const oracle_marlin_rand* R, uint64_t* xy, uint8_t* inf) {
  auto* h = (MarlinHandle*)hv; MARLIN_DISPATCH(h, o->round1(x, w, nw, R, xy, inf)); return 0;
}
int oracle_marlin_round2(void* hv, const uint64_t* ch, uint64_t* xy, uint8_t* inf) { auto* h = (MarlinHandle*)hv; MARLIN_DISPATCH(h, o->round2(ch, xy, inf)); return 0; }
int oracle_marlin_round3(void* hv, const uint64_t* beta, uint64_t* xy, uint8_t* inf) { auto* h = (MarlinHandle*)hv; MARLIN_DISPATCH(h, o->round3(beta, xy, inf)); return 0; }
int oracle_marlin_evaluate(void* hv, const uint64_t* gamma, uint64_t* out) { auto* h = (MarlinHandle*)hv; MARLIN_DISPATCH(h, o->evaluate_all(gamma, out)); return 0; }
// -> number of opening proofs (points in increasing into_repr order), < 0 on error
int oracle_marlin_open(void* hv, const uint64_t* xi, uint64_t* w_xy, uint8_t* w_inf, uint64_t* rand_v, uint8_t* has_rand) {
  auto* h = (MarlinHandle*)hv; int nproofs = 0; MARLIN_DISPATCH(h, nproofs = o->open_all(xi, w_xy, w_inf, rand_v, has_rand)); return nproofs;
}
// seconds: round-1 polynomials, commits, round-2 polynomials, commits, round-3 polynomials, commits, evaluations, openings
int oracle_marlin_phase_seconds(void* hv, double* out) { auto* h = (MarlinHandle*)hv; MARLIN_DISPATCH(h, memcpy(out, o->phase_s, sizeof(o->phase_s))); return 0; }
// polynomial by id (0..8: w z_a z_b mask t g_1 h_1 g_2 h_2; 9..20: a/b/c x row col val row_col): length, and the coefficients when out != NULL
long oracle_marlin_poly(void* hv, int id, uint64_t* out) {
  auto* h = (MarlinHandle*)hv; long len = -1;
  MARLIN_DISPATCH(h, { auto* p = o->poly_by_id(id); if (p) { len = (long)p->size(); if (out) memcpy(out, p->data(), 32 * p->size()); } });
  return len;
}

}  // extern "C"
