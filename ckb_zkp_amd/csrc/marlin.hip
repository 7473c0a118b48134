// Marlin prover behind the C ABI: the host-side orchestration of
//   /root/reference/marlin/src/lib.rs:97-181            create_random_proof (round -> PC::commit -> absorb -> squeeze)
//   /root/reference/marlin/src/ahp/indexer.rs:70-117    AHP::index — the device half: compose_matrix_polynomials
//   /root/reference/marlin/src/ahp/arithmetic.rs:98-172   (row / col / val / row_col over K, interpolation, evaluation over B)
//   /root/reference/marlin/src/ahp/prover.rs:86-427     prover_init, prover_{first,second,third}_round
//   /root/reference/marlin/src/pc/mod.rs:34-160         PC::commit, open, batch_open over KZG10 (pc/kzg10.rs:100-156)
// over the device primitives of this library (NTTs, element-wise Fr kernels, batch inversion, sparse products, gathers,
// vanishing-polynomial folds, Horner evaluation / division, MSMs on the resident SRS powers).  Every vector stays in HBM;
// the host handles scalars: the verifier messages (FiatShamirRng, fs_rng.cpp), a handful of mask coefficients and the
// 32-byte read-backs of evaluations and commitments.  No arithmetic kernels live here.
//
// What stays on the caller's side (Rust): circuit synthesis, make_matrices_square / balance_matrices / the per-row column
// sort of AHP::index (pure index manipulation, ahp/constraint_systems.rs:9-31,100-133), and the zk RNG: the mask
// coefficients and commitment blinders are inputs (`zkp_marlin_rand`).
#include <algorithm>
#include <chrono>
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "host_field.hpp"
#include "internal.hpp"
#include "msm_vtbl.hpp"

// fs_rng.cpp (C ABI, same library)
extern "C" {
int32_t zkp_fs_rng_new(const uint8_t*, size_t, zkp_fs_rng**);
int32_t zkp_fs_rng_free(zkp_fs_rng*);
int32_t zkp_fs_rng_absorb(zkp_fs_rng*, const uint8_t*, size_t);
int32_t zkp_fs_rng_rand_u128(zkp_fs_rng*, uint64_t[2]);
int32_t zkp_fs_rng_rand_fr(zkp_fs_rng*, zkp_curve_t, uint64_t*);
int32_t zkp_fs_rng_sample_outside_domain(zkp_fs_rng*, zkp_curve_t, uint32_t, uint64_t*);
}

namespace zkp {
namespace {

using namespace hostf;

size_t next_pow2(size_t n) {
  size_t s = 1;
  while (s < n) s <<= 1;
  return s;
}
int log2_of(size_t n) {
  int l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}

// ------------------------------------------------------------------------------------------------ device vectors
struct DVec {
  uint64_t* p = nullptr;   // device, Fr Montgomery
  size_t n = 0;
  DVec view(size_t a, size_t b) const { return DVec{p + 4 * a, b - a}; }
  DVec view(size_t a) const { return view(a, n); }
};

// size-keyed free lists: a proof allocates the same ~120 vectors every time, so after the first proof no hipMalloc / hipFree
// (each a device-wide synchronisation) is left in the path; reuse is stream-ordered (one stream)
struct Pool {
  std::map<size_t, std::vector<void*>> free_;
  std::vector<std::pair<void*, size_t>> live;
  void* raw(size_t bytes) {
    bytes = std::max<size_t>(bytes, 32);
    auto& f = free_[bytes];
    void* p = nullptr;
    if (!f.empty()) {
      p = f.back();
      f.pop_back();
    } else if (hipMalloc(&p, bytes) != hipSuccess) {
      throw StatusError{ZKP_ERR_OOM};
    }
    live.push_back({p, bytes});
    return p;
  }
  void release_all() {
    for (auto& l : live) free_[l.second].push_back(l.first);
    live.clear();
  }
  void trim() {
    release_all();
    for (auto& kv : free_)
      for (void* p : kv.second) (void)hipFree(p);
    free_.clear();
  }
};

struct Backend {
  zkp_ctx* ctx;
  int curve;
  HostField F;
  Pool* pool;
  hipStream_t st() const { return ctx->cur->stream; }
  DVec alloc(size_t n) { return DVec{reinterpret_cast<uint64_t*>(pool->raw(std::max<size_t>(n, 1) * 32)), n}; }
  DVec zeros(size_t n) {
    DVec v = alloc(n);
    if (n) ZKP_HIP(hipMemsetAsync(v.p, 0, n * 32, st()));
    return v;
  }
  DVec upload(const uint64_t* host_mont, size_t n) {
    DVec v = alloc(n);
    if (n) ZKP_HIP(hipMemcpyAsync(v.p, host_mont, n * 32, hipMemcpyHostToDevice, st()));
    return v;
  }
  DVec upload(const std::vector<FrE>& els) {
    std::vector<uint64_t> h(els.size() * 4 + 4);
    for (size_t i = 0; i < els.size(); i++) memcpy(&h[4 * i], els[i].data(), 32);
    DVec v = alloc(els.size());
    if (!els.empty()) {
      ZKP_HIP(hipMemcpyAsync(v.p, h.data(), els.size() * 32, hipMemcpyHostToDevice, st()));
      ZKP_HIP(hipStreamSynchronize(st()));             // h dies with this call
    }
    return v;
  }
  void copy_into(DVec dst, DVec src) {
    if (src.n) ZKP_HIP(hipMemcpyAsync(dst.p, src.p, src.n * 32, hipMemcpyDeviceToDevice, st()));
  }
  DVec pad(DVec v, size_t n) {
    DVec out = zeros(n);
    copy_into(out, v.view(0, std::min(v.n, n)));
    return out;
  }
  DVec shift(DVec v, size_t s) {
    DVec out = zeros(v.n + s);
    copy_into(out.view(s), v);
    return out;
  }
  void op(int o, DVec a, const uint64_t* b, DVec out, size_t n, const FrE* k = nullptr) {
    fr_vec_op(ctx, curve, o, a.p, b, k ? reinterpret_cast<const uint64_t*>(k->data()) : nullptr, out.p, n);
  }
  DVec mul(DVec a, DVec b) {
    DVec out = alloc(a.n);
    op(ZKP_VEC_MUL, a, b.p, out, a.n);
    return out;
  }
  DVec scale(DVec a, const FrE& k) {
    DVec out = alloc(a.n);
    op(ZKP_VEC_SCALE, a, nullptr, out, a.n, &k);
    return out;
  }
  DVec addc(DVec a, const FrE& k) {
    DVec out = alloc(a.n);
    op(ZKP_VEC_ADDC, a, nullptr, out, a.n, &k);
    return out;
  }
  DVec axpy(DVec a, DVec b, const FrE& k) {            // a + k*b, zero-extended to max(len)
    DVec out = pad(a, std::max(a.n, b.n));
    op(ZKP_VEC_AXPY, out.view(0, b.n), b.p, out.view(0, b.n), b.n, &k);
    return out;
  }
  void axpy_into(DVec acc, DVec b, const FrE& k, size_t at = 0) {
    DVec t = acc.view(at, at + b.n);
    op(ZKP_VEC_AXPY, t, b.p, t, b.n, &k);
  }
  void add_at(DVec v, size_t i, const FrE& k) {
    DVec e = v.view(i, i + 1);
    op(ZKP_VEC_ADDC, e, nullptr, e, 1, &k);
  }
  DVec sub(DVec a, DVec b) { return axpy(a, b, F.neg(F.one_())); }
  DVec binv(DVec a) {
    DVec out = pad(a, a.n);
    fr_batch_inverse(ctx, curve, out.p, out.n);
    return out;
  }
  DVec ntt(DVec v, size_t size, int o) {
    DVec out = pad(v, size);
    ntt_run(ctx, curve, reinterpret_cast<uint32_t*>(out.p), log2_of(size), o);
    return out;
  }
  uint64_t ntt_count = 0, ntt_elements = 0;            // bookkeeping for zkp_marlin_last_timing
  DVec fft(DVec v, size_t size) {
    ntt_count++;
    ntt_elements += size;
    return ntt(v, size, ZKP_NTT_FFT);
  }
  DVec ifft(DVec v, size_t size) {
    ntt_count++;
    ntt_elements += size;
    return ntt(v, size, ZKP_NTT_IFFT);
  }
  DVec pmul(DVec a, DVec b) {
    size_t size = next_pow2(a.n + b.n - 1);
    return ifft(mul(fft(a, size), fft(b, size)), size).view(0, a.n + b.n - 1);
  }
  std::pair<DVec, DVec> fold(DVec v, size_t n) {       // divide_by_vanishing_poly -> (q, rem)
    DVec q = alloc(v.n > n ? v.n - n : 0), rem = alloc(n);
    poly_vanishing_fold(ctx, curve, v.p, v.n, n, q.n ? q.p : nullptr, rem.p);
    return {q, rem};
  }
  DVec spmv(const uint32_t* rp, const uint32_t* col, const uint64_t* cf, DVec x, size_t nrows) {
    DVec out = alloc(nrows);
    fr_spmv(ctx, curve, rp, col, cf, nrows, x.p, out.p);
    return out;
  }
  DVec gather(DVec v, const int32_t* idx_dev, size_t n) {
    DVec out = alloc(n);
    fr_gather(ctx, v.p, idx_dev, n, out.p);
    return out;
  }
  FrE element(DVec v, size_t i) {
    FrE e{};
    ZKP_HIP(hipMemcpyAsync(e.data(), v.p + 4 * i, 32, hipMemcpyDeviceToHost, st()));
    ZKP_HIP(hipStreamSynchronize(st()));
    return e;
  }
  FrE evaluate(DVec v, const FrE& z) {
    FrE out{};
    if (v.n == 0) return out;
    poly_div_linear(ctx, curve, v.p, v.n, reinterpret_cast<const uint64_t*>(z.data()), nullptr,
                    reinterpret_cast<uint64_t*>(out.data()));
    return out;
  }
};

struct DevCsr3 {
  uint32_t *rp = nullptr, *col = nullptr;
  uint64_t* cf = nullptr;
};

}  // namespace
}  // namespace zkp

using namespace zkp;

// ------------------------------------------------------------------------------------------------ index
struct zkp_marlin_index {
  int curve = 0;
  size_t ni = 0, n = 0, pad_aux = 0, nnz = 0, xs = 0, hs = 0, ks = 0, bs = 0, max_degree = 0;
  DevCsr3 csr[3], csr_t[3];
  DVec h_el;
  DVec on_k[3][3];     // row, col, val
  DVec on_b[3][4];     // row, col, val, row_col
  DVec polys[3][4];    // a_row, a_col, a_val, a_row_col, b_..., c_...  (Index::iter order, indexer.rs:51-67)
  int32_t *w_idx = nullptr, *x_idx = nullptr;
  std::vector<void*> owned;
  Pool pool;           // scratch of the provers that use this index
  uint64_t* early_pinned = nullptr;     // pinned landing zone of the early commitment MSMs (marlin_prove: commit_early)
  ~zkp_marlin_index() {
    pool.trim();
    for (void* p : owned) (void)hipFree(p);
    if (early_pinned) (void)hipHostFree(early_pinned);
  }
};

namespace zkp {
namespace {

template <class T>
T* dev_copy(zkp_marlin_index* ix, hipStream_t st, const T* host, size_t count) {
  T* d = nullptr;
  if (hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  ix->owned.push_back(d);
  if (count) ZKP_HIP(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, st));
  return d;
}
DVec keep(zkp_marlin_index* ix, hipStream_t st, DVec v) {                // pool vector -> index-owned copy
  uint64_t* d = nullptr;
  if (hipMalloc(&d, std::max<size_t>(v.n, 1) * 32) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  ix->owned.push_back(d);
  if (v.n) ZKP_HIP(hipMemcpyAsync(d, v.p, v.n * 32, hipMemcpyDeviceToDevice, st));
  return DVec{d, v.n};
}
size_t reindex_by_subdomain(size_t hs, size_t xs, size_t j) {             // ahp/constraint_systems.rs reindex_by_subdomain
  const size_t period = hs / xs;
  if (j < xs) return j * period;
  const size_t i = j - xs;
  return i + i / (period - 1) + 1;
}

}  // namespace

zkp_marlin_index* marlin_index_upload(zkp_ctx* ctx, const zkp_marlin_index_desc* d) {
  ZKP_REQUIRE(d->curve == ZKP_BN254 || d->curve == ZKP_BLS12_381, ZKP_ERR_UNSUPPORTED_CURVE);
  ZKP_REQUIRE(d->num_inputs >= 1 && d->n >= d->num_inputs, ZKP_ERR_BAD_ARG);
  std::unique_ptr<zkp_marlin_index> ix(new zkp_marlin_index());
  ix->curve = d->curve;
  ix->ni = d->num_inputs;
  ix->n = d->n;
  ix->pad_aux = d->pad_aux;
  const zkp_csr* M[3] = {&d->a, &d->b, &d->c};
  size_t nnz = 0;
  for (auto* m : M) {
    ZKP_REQUIRE(m->row_ptr != nullptr, ZKP_ERR_BAD_ARG);
    nnz = std::max<size_t>(nnz, m->row_ptr[d->n]);
  }
  ZKP_REQUIRE(nnz >= 1, ZKP_ERR_BAD_ARG);
  ix->nnz = nnz;
  const size_t xs = next_pow2(ix->ni), hs = next_pow2(ix->n), ks = next_pow2(nnz), bs = next_pow2(3 * ks - 3);
  const int two_adicity = d->curve == ZKP_BN254 ? 28 : 32;
  ZKP_REQUIRE(log2_of(bs) + 1 <= two_adicity && hs > xs, ZKP_ERR_DOMAIN_TOO_LARGE);
  ix->xs = xs;
  ix->hs = hs;
  ix->ks = ks;
  ix->bs = bs;
  ix->max_degree = std::max(3 * hs + 2 * 1 - 1, 3 * ks - 3);              // ahp/mod.rs:66-84, zk_bound = 1
  Backend be{ctx, d->curve, fr_field(d->curve), &ix->pool};
  hipStream_t st = be.st();
  // H as a device vector: the evaluations of X over the domain; diagonal_evals^-1 = u / |H|
  DVec xpoly = be.zeros(hs);
  be.add_at(xpoly, 1, be.F.one_());
  DVec h_el = be.fft(xpoly, hs);
  ix->h_el = keep(ix.get(), st, h_el);
  DVec diag_inv = be.scale(h_el, be.F.inverse(be.F.from_u64(hs)));
  const char* names[4] = {"row", "col", "val", "row_col"};
  (void)names;
  for (int m = 0; m < 3; m++) {
    const uint32_t* rp = M[m]->row_ptr;
    const size_t k = rp[d->n];
    ix->csr[m].rp = dev_copy(ix.get(), st, rp, d->n + 1);
    ix->csr[m].col = dev_copy(ix.get(), st, M[m]->col, k);
    ix->csr[m].cf = dev_copy(ix.get(), st, M[m]->coeff, k * 4);
    // transposed, re-indexed matrix: t_on_h[kk] = sum_{(i, j): reindex(j) = kk} coeff * r_alpha[i]   (prover.rs:259-269)
    std::vector<uint32_t> jj(k), rows(k), tptr(hs + 1, 0), tcol(k);
    std::vector<uint64_t> tcf(k * 4 + 4);
    for (size_t i = 0; i < d->n; i++)
      for (uint32_t e = rp[i]; e < rp[i + 1]; e++) {
        rows[e] = (uint32_t)i;
        ZKP_REQUIRE(M[m]->col[e] < d->n, ZKP_ERR_BAD_ARG);
        jj[e] = (uint32_t)reindex_by_subdomain(hs, xs, M[m]->col[e]);
        tptr[jj[e] + 1]++;
      }
    for (size_t i = 0; i < hs; i++) tptr[i + 1] += tptr[i];
    {
      std::vector<uint32_t> cur(tptr.begin(), tptr.end() - 1);
      for (size_t e = 0; e < k; e++) {                                      // stable: entry order within a target row kept
        const uint32_t pos = cur[jj[e]]++;
        tcol[pos] = rows[e];
        memcpy(&tcf[4 * (size_t)pos], M[m]->coeff + 4 * e, 32);
      }
    }
    ix->csr_t[m].rp = dev_copy(ix.get(), st, tptr.data(), hs + 1);
    ix->csr_t[m].col = dev_copy(ix.get(), st, tcol.data(), k);
    ix->csr_t[m].cf = dev_copy(ix.get(), st, tcf.data(), k * 4);
    // row / col / val over K (arithmetic.rs:98-172), padded to |K| with (h_0, h_0, 0)
    std::vector<int32_t> irow(ks, 0), icol(ks, 0), idiag(ks, -1);
    std::vector<uint64_t> vcf(ks * 4, 0);
    for (size_t e = 0; e < k; e++) {
      irow[e] = (int32_t)jj[e];
      icol[e] = (int32_t)rows[e];
      idiag[e] = (int32_t)jj[e];
    }
    memcpy(vcf.data(), M[m]->coeff, k * 32);
    int32_t* d_irow = dev_copy(ix.get(), st, irow.data(), ks);
    int32_t* d_icol = dev_copy(ix.get(), st, icol.data(), ks);
    int32_t* d_idiag = dev_copy(ix.get(), st, idiag.data(), ks);
    DVec row = be.gather(h_el, d_irow, ks), colv = be.gather(h_el, d_icol, ks);
    DVec val = be.mul(be.upload(vcf.data(), ks), be.gather(diag_inv, d_idiag, ks));
    DVec rc = be.mul(row, colv);
    ZKP_HIP(hipStreamSynchronize(st));                 // the host staging vectors of this matrix die here
    DVec ev[4] = {row, colv, val, rc};
    for (int q = 0; q < 3; q++) ix->on_k[m][q] = keep(ix.get(), st, ev[q]);
    for (int q = 0; q < 4; q++) {
      DVec p = be.ifft(ev[q], ks);
      ix->polys[m][q] = keep(ix.get(), st, p);
      ix->on_b[m][q] = keep(ix.get(), st, be.fft(p, bs));
    }
    ix->pool.release_all();
    h_el = ix->h_el;                                   // pool vectors are gone; rebuild what the next matrix needs
    diag_inv = be.scale(h_el, be.F.inverse(be.F.from_u64(hs)));
  }
  // w_evals_on_h[i] = 0 if i % ratio == 0 else w_ext[i - i/ratio - 1] - x_evals_on_h[i]   (prover.rs:176-186)
  {
    const size_t ratio = hs / xs;
    std::vector<int32_t> wi(hs), xi(hs);
    for (size_t i = 0; i < hs; i++) {
      const bool in_x = i % ratio == 0;
      wi[i] = in_x ? -1 : (int32_t)(i - i / ratio - 1);
      xi[i] = in_x ? -1 : (int32_t)i;
    }
    ix->w_idx = dev_copy(ix.get(), st, wi.data(), hs);
    ix->x_idx = dev_copy(ix.get(), st, xi.data(), hs);
    ZKP_HIP(hipStreamSynchronize(st));
  }
  ix->pool.trim();
  return ix.release();
}

void marlin_index_info(const zkp_marlin_index* ix, uint64_t info[6]) {
  const uint64_t v[6] = {ix->xs, ix->hs, ix->ks, ix->bs, ix->max_degree, ix->nnz};
  memcpy(info, v, sizeof v);
}

void marlin_index_free(zkp_ctx* ctx, zkp_marlin_index* ix) {
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  delete ix;
}

// -> index_commitments[12] (affine Montgomery + identity flags), Index::iter order (lib.rs:77-83)
void marlin_index_commit(zkp_ctx* ctx, zkp_marlin_index* ix, uint64_t powers_g, uint64_t* comms_xy, uint8_t* inf) {
  const MsmVtbl* v1 = msm_vtbl(ix->curve, 1);
  const size_t jw64 = 3 * (size_t)v1->fN / 2;
  for (int m = 0; m < 3; m++)
    for (int q = 0; q < 4; q++) {
      std::vector<uint64_t> jac(jw64);
      msm_run(ctx, powers_g, 0, ix->polys[m][q].p, ix->polys[m][q].n, true, jac.data());
      point_into_affine(ctx, ix->curve, 1, jac.data(), comms_xy + (size_t)(4 * m + q) * 12, inf + 4 * m + q);
    }
}

// ------------------------------------------------------------------------------------------------ prover
namespace {

struct Challenger {                                    // lib.rs:105-158
  zkp_fs_rng* rng = nullptr;
  const uint64_t* fixed = nullptr;                     // 7 x 4 limbs Montgomery: alpha, eta_a, eta_b, eta_c, beta, gamma, xi
  int curve;
  HostField Fr, Fq;
  int log_hs;
  ~Challenger() {
    if (rng) zkp_fs_rng_free(rng);
  }
  FrE fixed_at(int i) const {
    FrE e{};
    memcpy(e.data(), fixed + 4 * i, 32);
    return e;
  }
  void put_fq(std::vector<uint8_t>& out, const uint32_t* mont) const {
    FrE a{};
    memcpy(a.data(), mont, 4 * Fq.N);
    FrE c = Fq.to_canonical(a);
    const uint8_t* b = reinterpret_cast<const uint8_t*>(c.data());
    out.insert(out.end(), b, b + 4 * Fq.N);
  }
  // GroupAffine::write: x, y, infinity byte; ark's zero() is (0, 1, true)
  void put_g1(std::vector<uint8_t>& out, const uint64_t* xy, bool inf) const {
    if (inf) {
      out.insert(out.end(), 4 * Fq.N, 0);
      std::vector<uint8_t> one(4 * Fq.N, 0);
      one[0] = 1;
      out.insert(out.end(), one.begin(), one.end());
      out.push_back(1);
      return;
    }
    put_fq(out, reinterpret_cast<const uint32_t*>(xy));
    put_fq(out, reinterpret_cast<const uint32_t*>(xy) + Fq.N);
    out.push_back(0);
  }
  void absorb(const std::vector<uint8_t>& b) {
    if (rng) zkp_fs_rng_absorb(rng, b.data(), b.size());
  }
  FrE rand_fr() {
    FrE e{};
    zkp_fs_rng_rand_fr(rng, (zkp_curve_t)curve, reinterpret_cast<uint64_t*>(e.data()));
    return e;
  }
  FrE outside() {
    FrE e{};
    zkp_fs_rng_sample_outside_domain(rng, (zkp_curve_t)curve, (uint32_t)log_hs, reinterpret_cast<uint64_t*>(e.data()));
    return e;
  }
};

struct Commitment {
  std::vector<uint64_t> xy, sxy;                       // affine Montgomery; sxy empty = no degree bound
  uint8_t inf = 0, sinf = 0;
};

}  // namespace

void marlin_prove(zkp_ctx* ctx, zkp_marlin_index* ix, uint64_t powers_g, uint64_t powers_gamma_g, const uint8_t* ivk_bytes,
                  size_t ivk_len, const uint64_t* x_mont, const uint64_t* w_mont, size_t n_w, const zkp_marlin_rand* rnd,
                  const uint64_t* fixed_challenges, zkp_marlin_proof* out) {
  const int curve = ix->curve;
  Backend be{ctx, curve, fr_field(curve), &ix->pool};
  const HostField& F = be.F;
  const MsmVtbl* v1 = msm_vtbl(curve, 1);
  const size_t aw64 = (size_t)v1->fN, jw64 = 3 * (size_t)v1->fN / 2;      // u64 words of an affine / Jacobian G1 point
  ZKP_REQUIRE(aw64 <= 12, ZKP_ERR_BAD_ARG);
  const size_t xs = ix->xs, hs = ix->hs, ks = ix->ks, bs = ix->bs, D = ix->max_degree, ni = ix->ni;
  ZKP_REQUIRE(ni + n_w + ix->pad_aux == ix->n, ZKP_ERR_BAD_ARG);
  ZKP_REQUIRE(bases_len(ctx, powers_g) >= D + 1 && bases_len(ctx, powers_gamma_g) >= 2, ZKP_ERR_BAD_ARG);
  hipStream_t st = be.st();
  // Pool memory and the index's pinned slots are recycled by the next proof: on EVERY exit (an exception between an early
  // commitment and the round that collects it included) the prover's stream and the lane's MSM workspace streams are drained
  // before the pool lets go of anything an in-flight MSM or its read-back may still touch.
  struct Release {
    Pool* p;
    zkp_lane* lane;
    hipStream_t st;
    ~Release() {
      for (int w = 1; w < zkp_lane::N_WS; w++)
        if (lane->ws[w].stream) (void)hipStreamSynchronize(lane->ws[w].stream);
      (void)hipStreamSynchronize(st);
      p->release_all();
    }
  } release{&ix->pool, ctx->cur, st};

  zkp_marlin_timing tm{};
  auto clk = [] { return std::chrono::steady_clock::now(); };
  auto t_begin = clk(), t_mark = t_begin;
  auto lap = [&](double* dst) {                          // phase boundary: the stream is drained, the host clock read
    ZKP_HIP(hipStreamSynchronize(st));
    auto now = clk();
    *dst += std::chrono::duration<double, std::milli>(now - t_mark).count();
    t_mark = now;
  };
  Challenger chal;
  chal.curve = curve;
  chal.Fr = F;
  chal.Fq = fq_field(curve);
  chal.log_hs = log2_of(hs);
  chal.fixed = fixed_challenges;
  if (!fixed_challenges) {
    // FiatShamirRng::from_seed(&to_bytes![index_verifier_key, public_input])  (lib.rs:105-106); public input = x[1..]
    ZKP_REQUIRE(ivk_bytes != nullptr, ZKP_ERR_BAD_ARG);
    std::vector<uint8_t> seed(ivk_bytes, ivk_bytes + ivk_len);
    for (size_t i = 1; i < ni; i++) {
      FrE e{};
      memcpy(e.data(), x_mont + 4 * i, 32);
      FrE c = F.to_canonical(e);
      const uint8_t* b = reinterpret_cast<const uint8_t*>(c.data());
      seed.insert(seed.end(), b, b + 32);
    }
    ZKP_REQUIRE(zkp_fs_rng_new(seed.data(), seed.size(), &chal.rng) == ZKP_OK, ZKP_ERR_OOM);
  }
  auto fr_of = [&](const uint64_t* p) {
    FrE e{};
    memcpy(e.data(), p, 32);
    return e;
  };

  // the mask polynomial is an input: prepared first so that its commitment (the largest MSM of the round) runs under the
  // interpolations below
  const size_t mask_len = 3 * hs + 2 * 1 - 2;
  DVec mask = be.alloc(mask_len);
  ZKP_HIP(hipMemcpyAsync(mask.p, rnd->mask, mask_len * 32, rnd->mask_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  {
    DVec mrem = be.fold(mask, hs).second;
    be.add_at(mask, 0, F.neg(be.element(mrem, 0)));   // mask - sigma_H(mask)/|H| ... exactly prover.rs:202-205
  }
  // label order used everywhere below: first round w, z_a, z_b, mask; second t, g_1, h_1; third g_2, h_2
  enum { W_, ZA_, ZB_, MASK_, T_, G1_, H1_, G2_, H2_, NLAB };
  DVec poly[NLAB];
  poly[MASK_] = mask;
  const bool bounded[NLAB] = {false, false, false, false, false, true, false, true, false};
  // Early commitments (round 3): a polynomial that is final long before its round ends (the mask polynomial — an input — and t(X)
  // of the second round) starts its commitment MSM at once on an MSM workspace stream of this lane, under the NTTs / pointwise
  // kernels that compute the rest of the round (which otherwise run alone: 12.6 + 4.9 ms per proof in a kernel trace);
  // commit_round collects the result instead of launching the MSM.  ZKP_MARLIN_EARLY=0: every MSM inside commit_round.
  static const bool early_on = !(getenv("ZKP_MARLIN_EARLY") && atoi(getenv("ZKP_MARLIN_EARLY")) == 0);
  constexpr int EARLY_MAX = 16;
  if (!ix->early_pinned) ZKP_HIP(hipHostMalloc(reinterpret_cast<void**>(&ix->early_pinned), EARLY_MAX * 24 * 8));
  struct Early {
    int label, ws;
    bool shifted;
    int kind;               // 0: commitment MSM over powers_of_g; 1: blinding MSM over powers_of_gamma_g (label / shifted: whose);
  };                        // 2: blinding term of opening proof `label`
  std::vector<Early> early;
  // -> index into `early` (-1: not started, the caller runs the MSM itself later).  offset: first SRS power (shifted commitments)
  auto start_early = [&](int label, const DVec& v, size_t offset = 0, bool shifted = false, int kind = 0) -> int {
    if (!early_on || (int)early.size() >= EARLY_MAX || v.n == 0) return -1;
    zkp_lane* L = ctx->cur;
    const int w = 1 + (int)(early.size() % (zkp_lane::N_WS_MSM - 1));
    hipStream_t ws_st = L->ws[w].stream;
    const uint64_t handle = kind == 0 ? powers_g : powers_gamma_g;
    ZKP_HIP(hipEventRecord(L->ev_fork, st));                         // v is complete on the prover's stream
    ZKP_HIP(hipStreamWaitEvent(ws_st, L->ev_fork, 0));
    ZKP_REQUIRE(offset <= bases_len(ctx, handle), ZKP_ERR_BAD_ARG);
    const size_t n = std::min(v.n, bases_len(ctx, handle) - offset);     // ark min(len) truncation, as msm_run_multi
    msm_run(ctx, handle, offset, v.p, n, true, nullptr, nullptr, nullptr, nullptr, w);
    ZKP_HIP(hipMemcpyAsync(ix->early_pinned + early.size() * 24, L->ws[w].out.p, jw64 * 8, hipMemcpyDeviceToHost, ws_st));
    ZKP_HIP(hipEventRecord(L->ws[w].done, ws_st));
    early.push_back({label, w, shifted, kind});
    return (int)early.size() - 1;
  };
  auto find_early = [&](int l, bool shifted, int kind) {
    for (size_t e = 0; e < early.size(); e++)
      if (early[e].label == l && early[e].shifted == shifted && early[e].kind == kind) return (int)e;
    return -1;
  };
  // result of early MSM e (Jacobian, jw64 words) once its workspace stream has reached it
  auto collect_early = [&](int e, uint64_t* dst) {
    ZKP_HIP(hipStreamWaitEvent(st, ctx->cur->ws[early[e].ws].done, 0));
    ZKP_HIP(hipStreamSynchronize(st));
    memcpy(dst, ix->early_pinned + (size_t)e * 24, jw64 * 8);
  };
  const size_t bound_of[NLAB] = {0, 0, 0, 0, 0, hs - 2, 0, ks - 2, 0};
  // a polynomial that is final: its commitment MSM (and the one against the shifted powers of a degree-bounded polynomial) starts now
  auto commit_early = [&](int l) {
    (void)start_early(l, poly[l]);
    if (bounded[l]) (void)start_early(l, poly[l], D - bound_of[l], true);
  };

  commit_early(MASK_);
  // the hiding blinders are inputs too: their two-term MSMs over powers_of_gamma_g (one full bucket pipeline each: ~1 ms of latency-
  // bound launches, little work) start now and are collected by the round that needs them — they used to run as a second, blocking
  // batch behind the commitment MSMs of every round
  const bool hide[NLAB] = {true, true, true, false, false, true, false, false, false};
  const uint64_t* blind_host[NLAB] = {rnd->blind_w, rnd->blind_z_a, rnd->blind_z_b, nullptr, nullptr, rnd->blind_g_1,
                                      nullptr,      nullptr,        nullptr};
  DVec blind_dev[NLAB], blind_s_dev;
  for (int l = 0; l < NLAB; l++)
    if (hide[l]) blind_dev[l] = be.upload(blind_host[l], 2);
  blind_s_dev = be.upload(rnd->blind_shifted_g_1, 2);
  for (int l = 0; l < NLAB; l++)
    if (hide[l]) {
      (void)start_early(l, blind_dev[l], 0, false, 1);
      if (bounded[l]) (void)start_early(l, blind_s_dev, 0, true, 1);
    }
  // (round 4: the witness crosses PCIe from pageable host memory AFTER the mask MSM is under way — the copy blocks the host for
  //  about a millisecond, during which the device used to idle)
  // ---- prover_init (prover.rs:86-147): z = x ++ w ++ padding ones; z_a = A z, z_b = B z
  DVec z = be.alloc(ix->n);
  ZKP_HIP(hipMemcpyAsync(z.p, x_mont, ni * 32, hipMemcpyHostToDevice, st));
  if (n_w) ZKP_HIP(hipMemcpyAsync(z.p + 4 * ni, w_mont, n_w * 32, hipMemcpyHostToDevice, st));
  if (ix->pad_aux) {
    DVec ones = be.zeros(ix->pad_aux);
    DVec tgt = z.view(ni + n_w, ix->n);
    const FrE o = F.one_();
    be.op(ZKP_VEC_ADDC, ones, nullptr, tgt, ix->pad_aux, &o);
  }
  DVec z_a_ev = be.spmv(ix->csr[0].rp, ix->csr[0].col, ix->csr[0].cf, z, ix->n);
  DVec z_b_ev = be.spmv(ix->csr[1].rp, ix->csr[1].col, ix->csr[1].cf, z, ix->n);
  // ---- first round (prover.rs:150-222)
  DVec x_poly = be.ifft(z.view(0, ni), xs);
  DVec x_on_h = be.fft(x_poly, hs);
  DVec w_ext = be.pad(z.view(ni), hs - xs);
  DVec w_on_h = be.sub(be.gather(w_ext, ix->w_idx, hs), be.gather(x_on_h, ix->x_idx, hs));
  auto masked = [&](DVec ev, const FrE& r) {          // interpolate(ev) + r * v_H
    DVec p = be.pad(be.ifft(ev, hs), hs + 1);
    be.add_at(p, 0, F.neg(r));
    be.add_at(p, hs, r);
    return p;
  };
  DVec w_poly = be.fold(masked(w_on_h, fr_of(rnd->w)), xs).first;
  poly[W_] = w_poly;
  commit_early(W_);                                    // under the interpolations of z_a and z_b
  DVec z_a = masked(z_a_ev, fr_of(rnd->z_a));
  poly[ZA_] = z_a;
  commit_early(ZA_);
  DVec z_b = masked(z_b_ev, fr_of(rnd->z_b));
  poly[ZB_] = z_b;
  // Round 5: three of the five product-domain transforms of the SECOND round (z_a, z_b and z = w v_X + x over the 4|H| domain) do not
  // depend on the verifier's first message, and the first round leaves the device under-used while its early commitment MSMs drain
  // (kernel trace: accumulate 0.4-0.9 of the time, nothing else): they run here instead of alone after alpha is known (3.5 ms of
  // transforms with no MSM to overlap).  MEASURED NEUTRAL (profiles/r05_marlin_ab.txt: rounds 1 + 2 incl.
  // commits 16.1 ms without, 16.5 with: the first round's early MSMs simply take the vector ALUs the transforms left, the device is
  // saturated either way), so it is OFF by default; ZKP_MARLIN_EARLY_FFT=1 enables it; same proof bytes.
  static const bool early_fft = getenv("ZKP_MARLIN_EARLY_FFT") && atoi(getenv("ZKP_MARLIN_EARLY_FFT")) != 0;
  DVec z_poly = be.axpy(be.sub(be.shift(w_poly, xs), w_poly), x_poly, F.one_());      // w * v_X + x  (prover.rs:277-281)
  const size_t r2_size = next_pow2(std::max({mask.n, hs + (z_a.n + z_b.n - 1), hs + z_poly.n}));   // r_alpha and t have |H| coefficients
  DVec ZA_early, ZB_early, ZZ_early;
  if (early_fft) {
    ZA_early = be.fft(z_a, r2_size);
    ZB_early = be.fft(z_b, r2_size);
    ZZ_early = be.fft(z_poly, r2_size);
  }
  const size_t bound[NLAB] = {0, 0, 0, 0, 0, hs - 2, 0, ks - 2, 0};
  Commitment comm[NLAB];

  // PC::commit (pc/mod.rs:34-71) of one round: one batched MSM call per base vector
  auto commit_round = [&](std::initializer_list<int> labels) {
    std::vector<size_t> offs, ns;
    std::vector<const uint64_t*> ptrs;
    std::vector<std::pair<int, bool>> slot;
    std::vector<std::pair<int, size_t>> early_slot;                 // (label, index into `early`) of the polynomials already under way
    std::vector<bool> early_shifted;
    for (int l : labels) {
      for (int sh = 0; sh <= (bounded[l] ? 1 : 0); sh++) {
        const int e = find_early(l, sh != 0, 0);
        if (e >= 0) {
          tm.commit_points += poly[l].n;
          early_slot.push_back({l, (size_t)e});
          early_shifted.push_back(sh != 0);
          continue;
        }
        offs.push_back(sh ? D - bound[l] : 0);                       // shifted_powers(bound) = powers[D - bound ..]
        ns.push_back(poly[l].n);
        ptrs.push_back(poly[l].p);
        slot.push_back({l, sh != 0});
      }
    }
    for (size_t nn : ns) tm.commit_points += nn;
    std::vector<uint64_t> jac((slot.size() + early_slot.size()) * jw64);
    msm_run_batch(ctx, powers_g, slot.size(), offs.data(), ptrs.data(), ns.data(), true, jac.data());
    if (!early_slot.empty()) {
      for (auto& es : early_slot) ZKP_HIP(hipStreamWaitEvent(st, ctx->cur->ws[early[es.second].ws].done, 0));
      ZKP_HIP(hipStreamSynchronize(st));
      for (size_t i = 0; i < early_slot.size(); i++) {
        memcpy(jac.data() + slot.size() * jw64, ix->early_pinned + early_slot[i].second * 24, jw64 * 8);
        slot.push_back({early_slot[i].first, (bool)early_shifted[i]});
      }
    }
    std::vector<size_t> boffs, bns;
    std::vector<const uint64_t*> bptrs;
    std::vector<size_t> bslot(slot.size(), (size_t)-1);
    std::vector<int> bearly(slot.size(), -1);
    for (size_t k = 0; k < slot.size(); k++)
      if (hide[slot[k].first]) {
        bearly[k] = find_early(slot[k].first, slot[k].second, 1);
        if (bearly[k] >= 0) continue;
        bslot[k] = bptrs.size();
        boffs.push_back(0);
        bns.push_back(2);
        bptrs.push_back(slot[k].second ? blind_s_dev.p : blind_dev[slot[k].first].p);
      }
    std::vector<uint64_t> bjac(std::max<size_t>(bptrs.size(), 1) * jw64);
    if (!bptrs.empty())
      msm_run_batch(ctx, powers_gamma_g, bptrs.size(), boffs.data(), bptrs.data(), bns.data(), true, bjac.data());
    {
      // commitment k = affine(MSM_k + blinding MSM_k): all of the round in ONE launch (a single-lane inversion each, side by side)
      std::vector<uint64_t> bj(slot.size() * jw64, 0), axy(slot.size() * aw64);
      std::vector<uint8_t> has(slot.size(), 0), ainf(slot.size(), 0);
      for (size_t k = 0; k < slot.size(); k++)
        if (bslot[k] != (size_t)-1) {
          has[k] = 1;
          memcpy(bj.data() + k * jw64, bjac.data() + bslot[k] * jw64, jw64 * 8);
        } else if (bearly[k] >= 0) {
          has[k] = 1;
          collect_early(bearly[k], bj.data() + k * jw64);
        }
      static const bool host_tail = !(getenv("ZKP_MARLIN_HOST_AFFINE") && atoi(getenv("ZKP_MARLIN_HOST_AFFINE")) == 0);
      if (host_tail) {
        std::vector<HostJac> pts(slot.size());
        for (size_t k = 0; k < slot.size(); k++) {
          pts[k] = host_jac_load(chal.Fq, jac.data() + k * jw64);
          if (has[k]) pts[k] = host_jac_add(chal.Fq, pts[k], host_jac_load(chal.Fq, bj.data() + k * jw64));
        }
        host_into_affine(chal.Fq, pts, axy.data(), aw64, ainf.data());
      } else {
        points_fold_into_affine(ctx, curve, 1, jac.data(), bj.data(), has.data(), slot.size(), axy.data(), ainf.data());
      }
      for (size_t k = 0; k < slot.size(); k++) {
        Commitment& c = comm[slot[k].first];
        std::vector<uint64_t>& dst = slot[k].second ? c.sxy : c.xy;
        dst.assign(axy.begin() + k * aw64, axy.begin() + (k + 1) * aw64);
        (slot[k].second ? c.sinf : c.inf) = ainf[k];
      }
    }
    // to_bytes![round commitments]: comm, shifted_exists byte, shifted or the empty commitment (pc/data_structures.rs:143-154)
    std::vector<uint8_t> bytes;
    for (int l : labels) {
      chal.put_g1(bytes, comm[l].xy.data(), comm[l].inf != 0);
      bytes.push_back(bounded[l] ? 1 : 0);
      if (bounded[l]) chal.put_g1(bytes, comm[l].sxy.data(), comm[l].sinf != 0);
      else chal.put_g1(bytes, nullptr, true);
    }
    chal.absorb(bytes);
  };

  lap(&tm.ms_round[0]);
  commit_round({W_, ZA_, ZB_, MASK_});                                     // lib.rs:109-112
  lap(&tm.ms_commit[0]);
  const FrE alpha = chal.rng ? chal.outside() : chal.fixed_at(0);          // ahp/verifier.rs:53-56
  const FrE ea = chal.rng ? chal.rand_fr() : chal.fixed_at(1);
  const FrE eb = chal.rng ? chal.rand_fr() : chal.fixed_at(2);
  const FrE ec = chal.rng ? chal.rand_fr() : chal.fixed_at(3);
  // ---- second round (prover.rs:230-321)
  const FrE one = F.one_();
  const FrE v_alpha = F.sub(F.pow2k(alpha, log2_of(hs)), one);
  DVec r_alpha_on_h = be.scale(be.binv(be.addc(be.scale(ix->h_el, F.neg(one)), alpha)), v_alpha);
  DVec r_alpha = be.ifft(r_alpha_on_h, hs);
  DVec t_on_h = be.zeros(hs);
  const FrE etas[3] = {ea, eb, ec};
  for (int m = 0; m < 3; m++)
    be.axpy_into(t_on_h, be.spmv(ix->csr_t[m].rp, ix->csr_t[m].col, ix->csr_t[m].cf, r_alpha_on_h, hs), etas[m]);
  DVec t_poly = be.ifft(t_on_h, hs);
  poly[T_] = t_poly;
  commit_early(T_);                                                        // under the product FFTs below
  {
    // m(X) = eta_c z_a z_b + eta_a z_a + eta_b z_b (z_a.n + z_b.n - 1 coefficients) is formed pointwise over the product domain
    const size_t m_n = z_a.n + z_b.n - 1;
    const size_t size = next_pow2(std::max({mask.n, r_alpha.n + m_n, t_poly.n + z_poly.n}));
    DVec prod = be.alloc(size);
    {
      const bool pre = early_fft && size == r2_size;
      DVec RA = be.fft(r_alpha, size), TT = be.fft(t_poly, size);
      DVec ZA = pre ? ZA_early : be.fft(z_a, size), ZB = pre ? ZB_early : be.fft(z_b, size), ZZ = pre ? ZZ_early : be.fft(z_poly, size);
      uint64_t kh[3 * 4];
      memcpy(kh, ea.data(), 32);
      memcpy(kh + 4, eb.data(), 32);
      memcpy(kh + 8, ec.data(), 32);
      marlin_round2_prod(ctx, curve, RA.p, ZA.p, ZB.p, TT.p, ZZ.p, kh, prod.p, size);
    }
    DVec q1 = be.axpy(be.ifft(prod, size), mask, one);
    auto hx = be.fold(q1, hs);
    poly[T_] = t_poly;
    poly[G1_] = hx.second.view(1, hs);
    poly[H1_] = hx.first.view(0, 2 * hs);
  }
  lap(&tm.ms_round[1]);
  commit_round({T_, G1_, H1_});                                            // lib.rs:117-120
  lap(&tm.ms_commit[1]);
  const FrE beta = chal.rng ? chal.outside() : chal.fixed_at(4);           // ahp/verifier.rs:76
  // ---- third round (prover.rs:331-427)
  const FrE v_beta = F.sub(F.pow2k(beta, log2_of(hs)), one);
  const FrE vab = F.mul(v_alpha, v_beta);
  DVec acc = be.alloc(ks);                                // = v_H(alpha) v_H(beta) sum_m eta_m val_m / ((beta - row_m)(alpha - col_m)) over K
  {
    const uint64_t* onk[9];
    for (int m = 0; m < 3; m++)
      for (int q = 0; q < 3; q++) onk[3 * m + q] = ix->on_k[m][q].p;
    uint64_t kh[5 * 4];
    const FrE kk[5] = {alpha, beta, F.mul(etas[0], vab), F.mul(etas[1], vab), F.mul(etas[2], vab)};
    for (int q = 0; q < 5; q++) memcpy(kh + 4 * q, kk[q].data(), 32);
    marlin_t3_evals(ctx, curve, onk, kh, acc.p, ks);
  }
  DVec t3 = be.ifft(acc, ks);
  poly[G2_] = t3.view(1, ks);
  commit_early(G2_);                                   // g_2 and its shifted commitment run under the |B|-sized transforms below
  const FrE ab = F.mul(alpha, beta);
  if (4 * ks - 3 <= bs && ks >= 4) {
    // (a - b t) over B in one fused pass + ONE inverse transform: b's evaluations over B are already the transform pmul would
    // recompute (|B| >= deg(b t) + 1: no wrap-around), so the interpolations of a and b and the forward transform of b_poly drop out
    // (five transforms of size |B| -> two); see marlin_h2_numerator (poly.hip)
    DVec t_on_b = be.fft(t3, bs);
    const uint64_t* onb[12];
    for (int m = 0; m < 3; m++)
      for (int q = 0; q < 4; q++) onb[4 * m + q] = ix->on_b[m][q].p;
    uint64_t kh[6 * 4];
    const FrE kk[6] = {alpha, beta, ab, F.mul(etas[0], vab), F.mul(etas[1], vab), F.mul(etas[2], vab)};
    for (int q = 0; q < 6; q++) memcpy(kh + 4 * q, kk[q].data(), 32);
    DVec num = be.alloc(bs);
    marlin_h2_numerator(ctx, curve, onb, t_on_b.p, kh, num.p, bs);
    DVec h2 = be.fold(be.ifft(num, bs).view(0, 4 * ks - 3), ks).first;
    poly[H2_] = h2.view(0, 3 * ks - 3);
  } else {
    DVec den[3];
    for (int m = 0; m < 3; m++)
      den[m] = be.addc(be.axpy(be.axpy(ix->on_b[m][3], ix->on_b[m][0], F.neg(alpha)), ix->on_b[m][1], F.neg(beta)), ab);
    DVec a_on_b = be.zeros(bs);
    for (int m = 0; m < 3; m++)
      be.axpy_into(a_on_b, be.mul(be.mul(ix->on_b[m][2], den[(m + 1) % 3]), den[(m + 2) % 3]), etas[m]);
    DVec a_poly = be.ifft(be.scale(a_on_b, vab), bs);
    DVec b_poly = be.ifft(be.mul(be.mul(den[0], den[1]), den[2]), bs);
    DVec h2 = be.fold(be.sub(a_poly.view(0, 3 * ks - 2), be.pmul(b_poly.view(0, 3 * ks - 2), t3)), ks).first;
    poly[H2_] = h2.view(0, 3 * ks - 3);
  }
  // Round 5: h_2's commitment starts before the round's host synchronisation, and the seven evaluations at beta (first- and
  // second-round polynomials: all known since the second round) run under its bucket sort instead of after gamma is known
  // — MEASURED SLOWER (profiles/r05_marlin_ab.txt: evaluations 1.0 -> 0.8 ms, batch_open +0.7 ms): OFF by default, ZKP_MARLIN_EARLY_EVAL=1 enables
  static const bool early_eval = getenv("ZKP_MARLIN_EARLY_EVAL") && atoi(getenv("ZKP_MARLIN_EARLY_EVAL")) != 0;
  uint64_t beta_evals[4 * G2_];
  if (early_eval) {
    commit_early(H2_);
    const uint64_t* qp[G2_];
    size_t qn[G2_];
    uint64_t qz[4 * G2_];
    for (int l = 0; l < G2_; l++) {
      qp[l] = poly[l].p;
      qn[l] = poly[l].n;
      memcpy(qz + 4 * l, beta.data(), 32);
    }
    poly_evaluate_batch(ctx, curve, G2_, qp, qn, qz, beta_evals);
  }
  lap(&tm.ms_round[2]);
  commit_round({G2_, H2_});                                                // lib.rs:124-127
  lap(&tm.ms_commit[2]);
  const FrE gamma = chal.rng ? chal.rand_fr() : chal.fixed_at(5);          // ahp/verifier.rs:86
  // ---- evaluations in query-set order: BTreeSet<(label, point)>, i.e. by label (lib.rs:147-156)
  struct Q {
    std::string label;
    DVec p;
    bool at_beta;
    int l;                                             // prover label index or -1 (index polynomial)
  };
  std::vector<Q> query;
  static const char* LAB[NLAB] = {"w", "z_a", "z_b", "mask", "t", "g_1", "h_1", "g_2", "h_2"};
  for (int l = 0; l < NLAB; l++) query.push_back({LAB[l], poly[l], l < G2_, l});
  static const char* MAT[3] = {"a", "b", "c"};
  static const char* KIND[4] = {"row", "col", "val", "row_col"};
  for (int m = 0; m < 3; m++)
    for (int q = 0; q < 4; q++) query.push_back({std::string(MAT[m]) + "_" + KIND[q], ix->polys[m][q], false, -1});
  std::sort(query.begin(), query.end(), [](const Q& a, const Q& b) { return a.label < b.label; });
  ZKP_REQUIRE(query.size() == ZKP_MARLIN_NUM_EVALS, ZKP_ERR_BAD_ARG);
  std::vector<uint8_t> ev_bytes;
  {
    // all 21 Horner chains are enqueued, then ONE read-back (a host round trip per polynomial cost 2.5 ms of a proof)
    std::vector<const uint64_t*> qp;
    std::vector<size_t> qn, slot;
    std::vector<uint64_t> qz;
    for (size_t i = 0; i < query.size(); i++) {
      if (early_eval && query[i].at_beta && query[i].l >= 0) {            // evaluated under the h_2 commitment above
        memcpy(out->evaluations + 4 * i, beta_evals + 4 * query[i].l, 32);
        continue;
      }
      qp.push_back(query[i].p.p);
      qn.push_back(query[i].p.n);
      slot.push_back(i);
      const FrE& pt = query[i].at_beta ? beta : gamma;
      qz.insert(qz.end(), reinterpret_cast<const uint64_t*>(pt.data()), reinterpret_cast<const uint64_t*>(pt.data()) + 4);
    }
    std::vector<uint64_t> ev(4 * std::max<size_t>(qp.size(), 1));
    poly_evaluate_batch(ctx, curve, qp.size(), qp.data(), qn.data(), qz.data(), ev.data());
    for (size_t k = 0; k < slot.size(); k++) memcpy(out->evaluations + 4 * slot[k], ev.data() + 4 * k, 32);
  }
  for (size_t i = 0; i < query.size(); i++) {
    FrE e{};
    memcpy(e.data(), out->evaluations + 4 * i, 32);
    const FrE c = F.to_canonical(e);
    const uint8_t* b = reinterpret_cast<const uint8_t*>(c.data());
    ev_bytes.insert(ev_bytes.end(), b, b + 32);
  }
  chal.absorb(ev_bytes);                                                   // lib.rs:157
  lap(&tm.ms_evaluations);
  FrE xi;
  uint64_t xi128[2] = {0, 0};
  if (chal.rng) {
    zkp_fs_rng_rand_u128(chal.rng, xi128);                                 // lib.rs:158: u128::rand(..).into()
    uint32_t limbs[8] = {(uint32_t)xi128[0], (uint32_t)(xi128[0] >> 32), (uint32_t)xi128[1], (uint32_t)(xi128[1] >> 32), 0, 0, 0, 0};
    xi = F.from_canonical(limbs);
  } else {
    xi = chal.fixed_at(6);
  }
  // ---- PC::batch_open (pc/mod.rs:73-160): per query point (ascending), labels ascending
  const int order = F.cmp_canonical(beta, gamma);
  const FrE pts[2] = {order <= 0 ? beta : gamma, order <= 0 ? gamma : beta};
  const bool pt_is_beta[2] = {order <= 0, order > 0};
  const int npts = order == 0 ? 1 : 2;
  const FrE xi2 = F.mul(xi, xi);
  std::vector<DVec> wq(npts);
  std::vector<int> wq_early(npts, -1);
  std::vector<std::array<FrE, 2>> rbs(npts);
  // the blinding polynomial rb0 + rb1 X of every opening is a combination of INPUTS (the hiding blinders) with powers of xi: known
  // now.  Its witness at any point is the constant rb1, so the one-term MSMs rb1 * gamma_g[0] start here, under the combinations,
  // divisions and witness MSMs below (they used to run as a blocking batch at the very end of the proof).
  std::vector<int> rb_early(npts, -1);
  std::vector<DVec> rb1_dev(npts);
  for (int k = 0; k < npts; k++) {
    std::array<FrE, 2> rb = {F.zero(), F.zero()};
    FrE c = one;
    for (const Q& q : query) {
      if (order != 0 && q.at_beta != pt_is_beta[k]) continue;
      const int l = q.l;
      if (l >= 0 && hide[l])
        for (int i = 0; i < 2; i++) rb[i] = F.add(rb[i], F.mul(c, fr_of(blind_host[l] + 4 * i)));
      if (l >= 0 && bounded[l] && hide[l]) {
        const FrE sc = F.mul(c, xi);
        for (int i = 0; i < 2; i++) rb[i] = F.add(rb[i], F.mul(sc, fr_of(rnd->blind_shifted_g_1 + 4 * i)));
      }
      c = F.mul(c, xi2);
    }
    rbs[k] = rb;
    if (!F.is_zero(rb[0]) || !F.is_zero(rb[1])) {
      rb1_dev[k] = be.upload(std::vector<FrE>{rb[1]});
      rb_early[k] = start_early(k, rb1_dev[k], 0, false, 2);
    }
  }
  for (int k = 0; k < npts; k++) {
    DVec p = be.zeros(D + 1);
    FrE c = one;
    for (const Q& q : query) {
      if (order != 0 && q.at_beta != pt_is_beta[k]) continue;
      be.axpy_into(p, q.p, c);
      const int l = q.l;
      if (l >= 0 && bounded[l]) be.axpy_into(p, q.p, F.mul(c, xi), D - bound[l]);
      c = F.mul(c, xi2);
    }
    DVec qv = be.alloc(D);
    poly_div_linear(ctx, curve, p.p, D + 1, reinterpret_cast<const uint64_t*>(pts[k].data()), qv.p, nullptr);
    wq[k] = qv;
    if (k + 1 < npts) wq_early[k] = start_early(-1, qv);            // under the next point's combination + division
  }
  {
    std::vector<size_t> offs, ns;
    std::vector<const uint64_t*> ptrs;
    std::vector<int> at(npts, -1);
    for (int k = 0; k < npts; k++)
      if (wq_early[k] < 0) {
        at[k] = (int)ptrs.size();
        offs.push_back(0);
        ns.push_back(D);
        ptrs.push_back(wq[k].p);
      }
    std::vector<uint64_t> wjac(npts * jw64), part(std::max<size_t>(ptrs.size(), 1) * jw64);
    tm.open_points += (uint64_t)npts * D;
    msm_run_batch(ctx, powers_g, ptrs.size(), offs.data(), ptrs.data(), ns.data(), true, part.data());
    for (int k = 0; k < npts; k++) {
      if (wq_early[k] >= 0) {
        ZKP_HIP(hipStreamWaitEvent(st, ctx->cur->ws[early[wq_early[k]].ws].done, 0));
        ZKP_HIP(hipStreamSynchronize(st));
        memcpy(wjac.data() + k * jw64, ix->early_pinned + (size_t)wq_early[k] * 24, jw64 * 8);
      } else {
        memcpy(wjac.data() + k * jw64, part.data() + (size_t)at[k] * jw64, jw64 * 8);
      }
    }
    out->num_opening_proofs = (uint32_t)npts;
    // witness of the blinding polynomial rb0 + rb1 X at the point: quotient rb1 (degree 0), rand_v = rb(point) = rb0 + rb1 z — host
    // arithmetic; the one-term MSMs rb1 * gamma_g[0] of both points run as ONE batched call, the sums and the affine conversions of
    // both opening proofs on the host with one inversion (was: per point an upload, a division launch chain, an MSM, a fold launch and
    // an into_affine launch, each with its own synchronisation)
    std::vector<int> has_rand(npts, 0);
    std::vector<FrE> rb1s;
    for (int k = 0; k < npts; k++) {
      out->opening_has_rand[k] = 0;
      memset(out->opening_rand_v + 4 * k, 0, 32);
      if (!F.is_zero(rbs[k][0]) || !F.is_zero(rbs[k][1])) {
        has_rand[k] = 1;
        const FrE ev = F.add(rbs[k][0], F.mul(rbs[k][1], pts[k]));
        out->opening_has_rand[k] = 1;
        memcpy(out->opening_rand_v + 4 * k, ev.data(), 32);
        rb1s.push_back(rbs[k][1]);
      }
    }
    std::vector<uint64_t> bj(std::max<size_t>(rb1s.size(), 1) * jw64);
    {
      std::vector<size_t> boffs, bns;
      std::vector<const uint64_t*> bptrs;
      std::vector<size_t> dst;
      for (int k = 0, b = 0; k < npts; k++) {
        if (!has_rand[k]) continue;
        if (rb_early[k] >= 0) {
          collect_early(rb_early[k], bj.data() + (size_t)b * jw64);
        } else {                                                       // early MSMs off / slots exhausted: blocking batch
          if (!rb1_dev[k].p) rb1_dev[k] = be.upload(std::vector<FrE>{rbs[k][1]});
          boffs.push_back(0);
          bns.push_back(1);
          bptrs.push_back(rb1_dev[k].p);
          dst.push_back((size_t)b);
        }
        b++;
      }
      if (!bptrs.empty()) {
        std::vector<uint64_t> tmp(bptrs.size() * jw64);
        msm_run_batch(ctx, powers_gamma_g, bptrs.size(), boffs.data(), bptrs.data(), bns.data(), true, tmp.data());
        for (size_t i2 = 0; i2 < dst.size(); i2++) memcpy(bj.data() + dst[i2] * jw64, tmp.data() + i2 * jw64, jw64 * 8);
      }
    }
    std::vector<HostJac> hp(npts);
    for (int k = 0, b = 0; k < npts; k++) {
      hp[k] = host_jac_load(chal.Fq, wjac.data() + (size_t)k * jw64);
      if (has_rand[k]) hp[k] = host_jac_add(chal.Fq, hp[k], host_jac_load(chal.Fq, bj.data() + (size_t)(b++) * jw64));
    }
    uint8_t oinf[2] = {0, 0};
    memset(out->opening_w, 0, sizeof out->opening_w);
    host_into_affine(chal.Fq, hp, out->opening_w, 12, oinf);
    for (int k = 0; k < npts; k++) out->opening_w_inf[k] = oinf[k];
  }
  // ---- results
  for (int l = 0; l < NLAB; l++) {
    memset(out->comm + l * 12, 0, 96);
    memcpy(out->comm + l * 12, comm[l].xy.data(), aw64 * 8);
    out->comm_inf[l] = comm[l].inf;
  }
  memset(out->shifted, 0, sizeof out->shifted);
  memcpy(out->shifted, comm[G1_].sxy.data(), aw64 * 8);
  memcpy(out->shifted + 12, comm[G2_].sxy.data(), aw64 * 8);
  out->shifted_inf[0] = comm[G1_].sinf;
  out->shifted_inf[1] = comm[G2_].sinf;
  const FrE* chs[6] = {&alpha, &ea, &eb, &ec, &beta, &gamma};
  for (int i = 0; i < 6; i++) memcpy(out->challenges + 4 * i, chs[i]->data(), 32);
  memcpy(out->challenges + 24, xi.data(), 32);
  lap(&tm.ms_open);
  tm.ms_total = std::chrono::duration<double, std::milli>(clk() - t_begin).count();
  tm.ntt_count = be.ntt_count;
  tm.ntt_elements = be.ntt_elements;
  ctx->last_marlin_timing = tm;
}

}  // namespace zkp
