// Groth16 prover orchestration on the device: the host-side mirror of
// /root/reference/groth16/src/prover.rs:124-211 (`create_proof`) and
// /root/reference/groth16/src/r1cs_to_qap.rs:113-172 (`R1CStoQAP::witness_map`).
//
// What stays on the caller's side: circuit synthesis (Rust closures filling `ProvingAssignment`,
// prover.rs:16-95).  What crosses the boundary: the three sparse matrices at/bt/ct (fixed per circuit,
// uploaded with the key) and the full assignment z = input_assignment ++ aux_assignment (Montgomery Fr).
//
// Device pipeline for one proof (everything resident in HBM, one stream, no host round trips until the
// 3 proof points come back):
//   1. a,b,c = A z, B z, C z (CSR, one lane per constraint row)            r1cs_to_qap.rs:131-142,154-159
//   2. ifft, coset_fft on a,b,c; ab = (a*b - c) / Z(g); coset_ifft -> h     r1cs_to_qap.rs:144-169
//   3. five MSMs; `into_repr()` (prover.rs:150-161) is fused into the MSM digit scan
//   4. assemble A, B, C                                                      prover.rs:164-210
//
// Restructuring that keeps the result identical (a proof is three group elements, so any evaluation order
// is bit-exact): the small fixed-point terms of prover.rs:165-177,183,213-228 are folded INTO the MSMs by
// extending each query with a few key points and the scalar vector with (1, r, s, -rs):
//     S      = z ++ [1, r, s, -r*s]
//     A_ext  = a_query    ++ [alpha_g1, delta_g1, inf,      inf     ]   -> g_a  = <A_ext , S>
//     B1_ext = b_g1_query ++ [beta_g1 , inf,      delta_g1, inf     ]   -> g1_b = <B1_ext, S>
//     B2_ext = b_g2_query ++ [beta_g2 , inf,      delta_g2, inf     ]   -> g2_b = <B2_ext, S>
//     L_ext  = l_query    ++ [inf,      inf,      inf,      delta_g1]   -> l'   = <L_ext , S[num_inputs..]>
// (z[0] = 1 so query[0] needs no special case; r == 0 makes r*g1_b the identity exactly as the reference's
// `if r != 0` branch does).  What remains is  C = s*g_a + r*g1_b + l' + h_acc  and three `into_affine()`.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "field_dev.hpp"
#include "host_field.hpp"
#include "internal.hpp"
#include "msm_vtbl.hpp"

using namespace zkp;

struct DevCsr {
  uint32_t* row_ptr = nullptr;
  uint32_t* col = nullptr;
  uint32_t* coeff = nullptr;
  size_t nnz = 0;
};

struct zkp_groth16_pk {
  int curve = 0;
  uint32_t num_inputs = 0, num_aux = 0, num_constraints = 0;
  int log_n = 0;
  size_t N = 0, nz = 0;
  DevCsr m[3];
  uint64_t hA = 0, hB1 = 0, hB2 = 0, hH = 0, hL = 0;
  bool share_b_sort = false;     // b_g1_query / b_g2_query: same length, window configuration and identity pattern
  bool chain_lh = false;         // H accumulates into L's buckets: one bucket reduction for l' + h_acc (bucket chaining, ctx.hpp)
  bool share_al_sort = false;    // L (stored index-aligned with z) reuses A's bucket sort: same scalars, same identity pattern
  bool share_l1 = false;         // A, L and (b_in_l1) B2 (+B1) share ONE level-1 sort pass over z (each filters its identities at level 2)
  bool b_in_l1 = false;          // the B queries have A's window configuration and take part in the shared pass
  bool c_folded = false;         // hL holds L - C^T G (fold_c_into_l): the prover skips C z and the c chain of the witness map
  bool h_lagrange = false;       // hH holds the H query in EVALUATION form over the coset (lagrange_h below): the prover skips the last transform
  // Base-sharded key (BASELINE configs[4]): this rank holds elements [q_lo, q_lo + q_n) of every
  // (extended) query; world == 0 means the whole key.  Index order: A, B1, B2, H, L.
  int shard_rank = 0, shard_world = 0;
  size_t q_lo[5] = {0, 0, 0, 0, 0}, q_n[5] = {0, 0, 0, 0, 0};
  struct PerLane {   // per in-flight proof (zkp_ctx lanes)
    DevBuf abc;      // 3 * N Fr
    DevBuf S;        // nz + 4 Fr
    DevBuf results;  // 6 XYZZ (G2-sized slots)
    DevBuf proof;    // [r, s] + device proof + flags
    // hipGraph of one proof on this lane (ZKP_GRAPH=1): state 0 = never run, 1 = ran eagerly once (scratch allocated,
    // signature recorded), 2 = captured.  `sig` = every device pointer the captured launches embed.
    bool warmed = false;   // a proof has run on this lane with this key: every scratch buffer has its size
    int graph_state = 0;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    std::vector<const void*> sig;
  } lane[zkp_ctx::N_LANES];
  DevBuf consts;     // zinv etc.
};

namespace zkp {

// a[row] = sum coeff * z[col] ; rows >= num_constraints: a gets z[row - nc] for the next num_inputs rows, 0 after
template <class P>
__global__ __launch_bounds__(256) void csr_eval_kernel(const uint32_t* __restrict__ row_ptr,
                                                       const uint32_t* __restrict__ col,
                                                       const uint32_t* __restrict__ coeff,
                                                       const uint32_t* __restrict__ z, uint32_t nc, uint32_t N,
                                                       uint32_t num_inputs, int is_a, uint32_t* __restrict__ out) {
  using F = Fp<P>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  F acc = F::zero();
  if (i < nc) {
    const F one = F::one();
    for (uint32_t k = row_ptr[i]; k < row_ptr[i + 1]; k++) {
      F v = F::load(z + (size_t)col[k] * 8);
      F cf = F::load(coeff + (size_t)k * 8);
      if (cf == one) acc = acc + v;          // evaluate_constraint's is_one fast path (r1cs_to_qap.rs:39-43)
      else acc = acc + v * cf;
    }
  } else if (is_a && i - nc < num_inputs) {
    acc = F::load(z + (size_t)(i - nc) * 8); // r1cs_to_qap.rs:140-142
  }
  acc.store(out + (size_t)i * 8);
}

// ab[i] = (a[i]*b[i] - c[i]) * zinv      (r1cs_to_qap.rs:150,164-168)
template <class P>
__global__ __launch_bounds__(256) void qap_pointwise_kernel(uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                            const uint32_t* __restrict__ c,
                                                            const uint32_t* __restrict__ zinv, uint32_t N) {
  using F = Fp<P>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  F x = F::load(a + (size_t)i * 8) * F::load(b + (size_t)i * 8) - F::load(c + (size_t)i * 8);
  (x * F::load(zinv)).store(a + (size_t)i * 8);
}

// a[i] = a[i]*b[i] * zinv: the pointwise step of a key with C folded into its L query (fold_c_into_l)
template <class P>
__global__ __launch_bounds__(256) void qap_pointwise_ab_kernel(uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                               const uint32_t* __restrict__ zinv, uint32_t N) {
  using F = Fp<P>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  (F::load(a + (size_t)i * 8) * F::load(b + (size_t)i * 8) * F::load(zinv)).store(a + (size_t)i * 8);
}

// out[i] = canonical(mult * base^i), 8 words each: the scalar tables of the key transforms (lagrange_h)
template <class P>
__global__ __launch_bounds__(256) void pow_canon_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ base,
                                                        const uint32_t* __restrict__ mult, uint32_t count) {
  using F = Fp<P>;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  (F::load(base).pow_u64(i) * F::load(mult)).from_mont().store(out + (size_t)i * 8);
}

// consts[0] = (g^N - 1)^-1
template <class P>
__global__ void qap_consts_kernel(uint32_t* consts, int log_n) {
  if (threadIdx.x || blockIdx.x) return;
  using F = Fp<P>;
  F g;
#pragma unroll
  for (int i = 0; i < 8; i++) g.v[i] = P::GEN[i];
  for (int i = 0; i < log_n; i++) g = g.sqr();
  (g - F::one()).inv().store(consts);
}

// S tail: [1, r, s, -r*s]
template <class P>
__global__ void scalar_tail_kernel(uint32_t* tail, const uint32_t* rs) {
  if (threadIdx.x || blockIdx.x) return;
  using F = Fp<P>;
  F r = F::load(rs), s = F::load(rs + 8);
  F::one().store(tail);
  r.store(tail + 8);
  s.store(tail + 16);
  (r * s).neg().store(tail + 24);
}

static DevCsr upload_csr(zkp_ctx* ctx, const zkp_csr& m, uint32_t rows) {
  DevCsr d;
  ZKP_REQUIRE(m.row_ptr != nullptr, ZKP_ERR_BAD_ARG);
  d.nnz = m.row_ptr[rows];
  ZKP_REQUIRE(d.nnz == 0 || (m.col && m.coeff), ZKP_ERR_BAD_ARG);
  if (hipMalloc(&d.row_ptr, ((size_t)rows + 1) * 4) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  if (hipMalloc(&d.col, std::max<size_t>(d.nnz, 1) * 4) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  if (hipMalloc(&d.coeff, std::max<size_t>(d.nnz, 1) * 32) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  ZKP_HIP(hipMemcpyAsync(d.row_ptr, m.row_ptr, ((size_t)rows + 1) * 4, hipMemcpyHostToDevice, ctx->cur->stream));
  if (d.nnz) {
    ZKP_HIP(hipMemcpyAsync(d.col, m.col, d.nnz * 4, hipMemcpyHostToDevice, ctx->cur->stream));
    ZKP_HIP(hipMemcpyAsync(d.coeff, m.coeff, d.nnz * 32, hipMemcpyHostToDevice, ctx->cur->stream));
  }
  return d;
}

// query ++ tail points (tail entry nullptr = identity)
// contiguous, balanced index ranges: the first (n % world) ranks get one extra element
static void shard_bounds(size_t n, int rank, int world, size_t* lo, size_t* cnt) {
  if (world <= 0) {
    *lo = 0;
    *cnt = n;
    return;
  }
  const size_t base = n / (size_t)world, rem = n % (size_t)world, r = (size_t)rank;
  *lo = r * base + std::min(r, rem);
  *cnt = base + (r < rem ? 1 : 0);
}
// only elements [lo, lo + cnt) of the extended query are uploaded (the whole of it for an unsharded key)
// `lead` identity points are put in front of the query (the L query is stored index-aligned with z: lead = num_inputs);
// flags_out (optional) receives the identity flags of the uploaded slice
static uint64_t upload_ext(zkp_ctx* ctx, int curve, int group, const uint64_t* q, const uint8_t* inf, size_t n,
                           size_t limbs_per_point, const uint64_t* const tail[4], size_t lo, size_t cnt, size_t lead = 0,
                           std::vector<uint8_t>* flags_out = nullptr, int c_hint = 0, int cap_hint = 0, int lgk = -1) {
  std::vector<uint64_t> xy(std::max<size_t>(cnt, 1) * limbs_per_point, 0);
  std::vector<uint8_t> fl(std::max<size_t>(cnt, 1), 0);
  for (size_t j = 0; j < cnt; j++) {
    if (lo + j < lead) {
      fl[j] = 1;
      continue;
    }
    const size_t i = lo + j - lead;
    if (i < n) {
      memcpy(xy.data() + j * limbs_per_point, q + i * limbs_per_point, limbs_per_point * 8);
      fl[j] = inf ? inf[i] : 0;
    } else if (tail[i - n]) {
      memcpy(xy.data() + j * limbs_per_point, tail[i - n], limbs_per_point * 8);
    } else {
      fl[j] = 1;
    }
  }
  if (flags_out) *flags_out = fl;
  return bases_upload(ctx, curve, group, xy.data(), fl.data(), cnt, c_hint, cap_hint, lgk);
}

// The H query in evaluation form (round 4).  h = coset_ifft(v) (r1cs_to_qap.rs:169) is linear, h_i = g^-i / N sum_j w^-ij v_j, so
//     sum_i h_i H_i = sum_j v_j H'_j     with     H'_j = sum_i w^-ij (g^-i / N) H_i
// — a group-element transform of the key, done once at upload (N log N / 2 scalar multiplications on the device, 0.5 s at 2^20),
// after which the prover feeds the pointwise values v = (a b - c) / Z(g) on the coset straight to the H MSM: 18 instead of 21
// transform passes per proof and one transform less on the witness map -> H chain.  The proof is the same group element; terms
// beyond min(h_len, N) are the identity (prover.rs:186-187 truncates there).  ZKP_H_LAGRANGE=0: the coefficient-form key.
// (the devices of an in-process multi-GPU key are uploaded one after the other from the same descriptor: one transform serves all)
static thread_local bool lagrange_keep_cache = false;
static thread_local struct LagrangeCache {
  const void* q = nullptr;
  size_t used = 0;
  int log_n = -1, curve = -1;
  std::vector<uint64_t> xy;
  std::vector<uint8_t> inf;
  void clear() {
    q = nullptr;
    std::vector<uint64_t>().swap(xy);
    std::vector<uint8_t>().swap(inf);
  }
} lagrange_cache, lagrange_cache_g;                       // the transformed H query | the bases of the C part (mode 1)
// mode 1: G_k = (zinv / N) sum_i w^-ik H_i instead (the inverse transform of H itself, scaled by zinv = 1 / Z(g)): the bases of
// the C part of h, see fold_c_into_l below.
static void lagrange_h(zkp_ctx* ctx, int curve, const uint64_t* h_query, const uint8_t* h_inf, size_t h_used, int log_n,
                       std::vector<uint64_t>* xy_out, std::vector<uint8_t>* inf_out, int mode = 0) {
  using namespace hostf;
  const HostField F = fr_field(curve);
  const size_t N = (size_t)1 << log_n, fq = curve == ZKP_BN254 ? 4 : 6;
  const MsmVtbl* v1 = msm_vtbl(curve, 1);
  auto ld = [&](const uint32_t* p) {
    HostField::E e{};
    memcpy(e.data(), p, 32);
    return e;
  };
  const int adicity = curve == ZKP_BN254 ? consts::Bn254Fr::TWO_ADICITY : consts::Bls381Fr::TWO_ADICITY;
  HostField::E wi = ld(curve == ZKP_BN254 ? consts::Bn254Fr::ROOT_INV : consts::Bls381Fr::ROOT_INV);
  const HostField::E gi = ld(curve == ZKP_BN254 ? consts::Bn254Fr::GEN_INV : consts::Bls381Fr::GEN_INV);
  wi = F.pow2k(wi, adicity - log_n);                                   // w_N^-1
  // scal_i = g^-i / N (mode 1: zinv / N for every i), tw_e = w^-e: generated on the device (canonical words) from four Montgomery constants
  HostField::E mult = F.inverse(F.from_u64((uint64_t)N));
  HostField::E sbase = gi;
  if (mode == 1) {                                                     // zinv = 1 / (g^N - 1)
    const HostField::E g = ld(curve == ZKP_BN254 ? consts::Bn254Fr::GEN : consts::Bls381Fr::GEN);
    mult = F.mul(mult, F.inverse(F.sub(F.pow2k(g, log_n), F.one_())));
    sbase = F.one_();
  }
  const HostField::E one = F.one_();
  uint32_t hconst[4 * 8];
  memcpy(hconst, sbase.data(), 32);
  memcpy(hconst + 8, mult.data(), 32);
  memcpy(hconst + 16, wi.data(), 32);
  memcpy(hconst + 24, one.data(), 32);
  hipStream_t st = ctx->cur->stream;
  DevBuf d_xy, d_inf, d_scal, d_tw, d_X, d_oxy, d_oinf;
  const size_t ab = v1->aff_bytes;
  char* xy = d_xy.as<char>(std::max<size_t>(h_used, 1) * ab);
  uint8_t* inf = d_inf.as<uint8_t>(std::max<size_t>(h_used, 1));
  ZKP_REQUIRE(ab == 2 * fq * 8, ZKP_ERR_BAD_ARG);
  if (h_used) ZKP_HIP(hipMemcpyAsync(xy, h_query, h_used * ab, hipMemcpyHostToDevice, st));
  if (h_used && h_inf) ZKP_HIP(hipMemcpyAsync(inf, h_inf, h_used, hipMemcpyHostToDevice, st));
  else ZKP_HIP(hipMemsetAsync(inf, 0, std::max<size_t>(h_used, 1), st));
  const size_t ntw = std::max<size_t>(N / 2, 1);
  uint32_t* dscal = d_scal.as<uint32_t>(N * 8 + 32);
  uint32_t* dtw = d_tw.as<uint32_t>(ntw * 8);
  uint32_t* dconst = dscal + N * 8;
  ZKP_HIP(hipMemcpyAsync(dconst, hconst, sizeof hconst, hipMemcpyHostToDevice, st));
  if (curve == ZKP_BN254) {
    hipLaunchKernelGGL(pow_canon_kernel<Bn254Fr>, dim3((N + 255) / 256), dim3(256), 0, st, dscal, dconst, dconst + 8, (uint32_t)N);
    hipLaunchKernelGGL(pow_canon_kernel<Bn254Fr>, dim3((ntw + 255) / 256), dim3(256), 0, st, dtw, dconst + 16, dconst + 24, (uint32_t)ntw);
  } else {
    hipLaunchKernelGGL(pow_canon_kernel<Bls381Fr>, dim3((N + 255) / 256), dim3(256), 0, st, dscal, dconst, dconst + 8, (uint32_t)N);
    hipLaunchKernelGGL(pow_canon_kernel<Bls381Fr>, dim3((ntw + 255) / 256), dim3(256), 0, st, dtw, dconst + 16, dconst + 24, (uint32_t)ntw);
  }
  char* X = d_X.as<char>(N * v1->bucket_bytes);           // stage points in the bucket (unsaturated) layout
  char* oxy = d_oxy.as<char>(N * ab);
  uint8_t* oinf = d_oinf.as<uint8_t>(N);
  v1->gfft(st, xy, inf, h_used, dscal, dtw, (uint32_t)log_n, X, oxy, oinf);
  ZKP_HIP(hipGetLastError());
  xy_out->resize(N * 2 * fq);
  inf_out->resize(N);
  ZKP_HIP(hipMemcpyAsync(xy_out->data(), oxy, N * ab, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipMemcpyAsync(inf_out->data(), oinf, N, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
}

// The C matrix folded into the L query (round 4).  The quotient's coefficients are h = coset_ifft(a_c b_c zinv) - zinv C(X) (the
// coset evaluations c_c of C(X) = interpolant of C z go through coset_ifft unchanged), so
//     sum_i h_i H_i = sum_j (a_c b_c zinv)_j H'_j  -  sum_k (C z)_k G_k,      G_k = (zinv / N) sum_i w^-ik H_i
// and the second sum is LINEAR in the assignment: sum_m z_m D_m with D_m = sum_k C_km G_k.  With L'_m = L_m - D_m (inputs: -D_m,
// in the slots that are the identity in the index-aligned L table) the prover needs neither C z nor the ifft -> coset_fft chain
// of c (r1cs_to_qap.rs:155-162): 12 instead of 18 transform passes per proof, for every assignment, satisfying or not.
static void fold_c_into_l(zkp_ctx* ctx, const zkp_groth16_pk_desc* d, size_t nz, const std::vector<uint64_t>& g_xy,
                          const std::vector<uint8_t>& g_inf, std::vector<uint64_t>* l_xy, std::vector<uint8_t>* l_inf) {
  using namespace hostf;
  const HostField F = fr_field(d->curve);
  const MsmVtbl* v1 = msm_vtbl(d->curve, 1);
  const size_t fq = d->curve == ZKP_BN254 ? 4 : 6, ab = v1->aff_bytes, ni = d->num_inputs;
  const size_t nnz = d->ct.row_ptr[d->num_constraints];
  // CSC of C: column = variable (inputs ++ aux), rows ascending
  std::vector<uint32_t> col_ptr(nz + 1, 0), rows(std::max<size_t>(nnz, 1)), coeff(std::max<size_t>(nnz, 1) * 8, 0);
  std::vector<uint8_t> kind(std::max<size_t>(nnz, 1), 0);
  for (size_t e = 0; e < nnz; e++) {
    ZKP_REQUIRE(d->ct.col[e] < nz, ZKP_ERR_BAD_ARG);
    col_ptr[d->ct.col[e] + 1]++;
  }
  // HEAVY columns: the kernel gives one lane per variable, and that lane walks its column serially — a 255-step
  // double-and-add per general coefficient.  A dense column (the constant-one variable of packing / x * x_inv = 1 constraints, any
  // variable with 1e5+ entries of C) would be a single-lane chain of 1e6-1e8 point operations.  Columns above a cost threshold leave
  // the kernel's CSC and take ONE variable-base MSM each (msm_var_run over the gathered G rows), subtracted on the host below.
  // Threshold, MEASURED in round 6 (tests/test_gpu_fuzz.py::test_groth16_key_fold_cut_is_chosen_per_key): one cost unit = one
  // point operation of a single lane ≈ 8 µs alone and ≈ 16 µs when the lanes of a wave diverge over different coefficients (40 columns
  // x 57 000 units: 0.45 s; 300 x 53 200: 0.85 s), while a heavy column = host gather + H2D + msm_var_run + sync ≈ 1.7-3 ms ≈ 150 units,
  // one after the other.  The fixed 50 000 of round 5 was therefore far too HIGH for the common case (a handful of dense columns: each
  // up to 0.8 s of kernel chain instead of a 2 ms MSM) and only right when there are hundreds of them.  So the cut is chosen per key:
  // with the column costs sorted descending, k heavy columns cost about cost[k] + 150 k — the k that minimises it, never cutting below
  // 1000 (≈ 10 ms of chain).  12 columns of 38 000: twelve MSMs (≈ 30 ms) instead of a 0.3-0.6 s chain; 3000 columns of 50 000 stay in
  // the kernel (≈ 0.8 s, side by side) instead of 5 s of serial MSMs.
  // cfg.lfold_heavy_cost > 0 (zkp_ctx_config.c_fold_heavy_cost / ZKP_LFOLD_HEAVY_COST) fixes the threshold instead.
  HostField::E one = F.one_(), minus_one = F.neg(one);
  std::vector<uint8_t> heavy(nz, 0);
  std::vector<uint32_t> heavy_cols;
  {
    std::vector<uint64_t> cost(nz, 0);
    for (size_t e = 0; e < nnz; e++) {
      HostField::E c{};
      memcpy(c.data(), d->ct.coeff + e * 4, 32);
      cost[d->ct.col[e]] += (c == one || c == minus_one) ? 1 : 380;       // 255 doublings + ~125 additions
    }
    uint64_t heavy_cost = ctx->cfg.lfold_heavy_cost > 0 ? (uint64_t)ctx->cfg.lfold_heavy_cost : 0;
    if (!heavy_cost) {
      const uint64_t floor_cost = 1000, per_msm = 150;
      std::vector<uint64_t> big;
      for (uint64_t c : cost)
        if (c > floor_cost) big.push_back(c);
      std::sort(big.begin(), big.end(), std::greater<uint64_t>());
      heavy_cost = floor_cost;                                   // k = all of `big` heavy
      uint64_t best = floor_cost + per_msm * big.size();
      for (size_t k = 0; k < big.size(); k++)                    // k heavy columns, the kernel's longest chain is big[k]
        if (big[k] + per_msm * k < best) {
          best = big[k] + per_msm * k;
          heavy_cost = big[k];
        }
    }
    for (size_t m = 0; m < nz; m++)
      if (cost[m] > heavy_cost) {
        heavy[m] = 1;
        heavy_cols.push_back((uint32_t)m);
      }
  }
  // (row, entry) lists of the heavy columns: 8 bytes per entry; the points and coefficients of ONE column at a time are gathered
  // right before its MSM below (gathering every heavy column up front held 100 bytes per entry on the host)
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> hv(heavy_cols.size());
  if (!heavy_cols.empty()) {
    std::vector<uint32_t> slot(nz, 0);
    for (size_t i = 0; i < heavy_cols.size(); i++) slot[heavy_cols[i]] = (uint32_t)i;
    for (uint32_t k = 0; k < d->num_constraints; k++)
      for (uint32_t e = d->ct.row_ptr[k]; e < d->ct.row_ptr[k + 1]; e++) {
        const uint32_t m = d->ct.col[e];
        if (!heavy[m]) continue;
        hv[slot[m]].emplace_back(k, e);
        col_ptr[m + 1]--;                                        // (counts, before the prefix sum below)
      }
  }
  for (size_t m = 0; m < nz; m++) col_ptr[m + 1] += col_ptr[m];
  std::vector<uint32_t> cur(col_ptr.begin(), col_ptr.end() - 1);
  for (uint32_t k = 0; k < d->num_constraints; k++)
    for (uint32_t e = d->ct.row_ptr[k]; e < d->ct.row_ptr[k + 1]; e++) {
      if (heavy[d->ct.col[e]]) continue;
      const uint32_t pos = cur[d->ct.col[e]]++;
      rows[pos] = k;
      HostField::E c{};
      memcpy(c.data(), d->ct.coeff + (size_t)e * 4, 32);
      if (c == one) kind[pos] = 1;
      else if (c == minus_one) kind[pos] = 2;
      else memcpy(&coeff[(size_t)pos * 8], c.data(), 32);          // Montgomery words; the kernel converts
    }
  // the L query index-aligned with the assignment (leading identities for the inputs)
  std::vector<uint64_t> lq(nz * 2 * fq, 0);
  std::vector<uint8_t> li(nz, 1);
  for (size_t m = ni; m < nz; m++) {
    li[m] = d->l_inf ? d->l_inf[m - ni] : 0;
    if (!li[m]) memcpy(&lq[m * 2 * fq], d->l_query + (m - ni) * 2 * fq, ab);
  }
  hipStream_t st = ctx->cur->stream;
  DevBuf b_l, b_li, b_cp, b_rows, b_kind, b_coeff, b_g, b_gi, b_o, b_oi;
  auto up = [&](DevBuf& b, const void* p, size_t bytes) {
    void* dp = b.get(std::max<size_t>(bytes, 16));
    ZKP_HIP(hipMemcpyAsync(dp, p, bytes, hipMemcpyHostToDevice, st));
    return dp;
  };
  const char* dl = (const char*)up(b_l, lq.data(), lq.size() * 8);
  const uint8_t* dli = (const uint8_t*)up(b_li, li.data(), li.size());
  const uint32_t* dcp = (const uint32_t*)up(b_cp, col_ptr.data(), col_ptr.size() * 4);
  const uint32_t* drows = (const uint32_t*)up(b_rows, rows.data(), rows.size() * 4);
  const uint8_t* dkind = (const uint8_t*)up(b_kind, kind.data(), kind.size());
  const uint32_t* dcoeff = (const uint32_t*)up(b_coeff, coeff.data(), coeff.size() * 4);
  const char* dg = (const char*)up(b_g, g_xy.data(), g_xy.size() * 8);
  const uint8_t* dgi = (const uint8_t*)up(b_gi, g_inf.data(), g_inf.size());
  char* o = (char*)b_o.get(nz * ab);
  uint8_t* oi = (uint8_t*)b_oi.get(nz);
  v1->lfold(st, dl, dli, nz, dcp, drows, dkind, dcoeff, dg, dgi, o, oi);
  ZKP_HIP(hipGetLastError());
  l_xy->resize(nz * 2 * fq);
  l_inf->resize(nz);
  ZKP_HIP(hipMemcpyAsync(l_xy->data(), o, nz * ab, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipMemcpyAsync(l_inf->data(), oi, nz, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
  if (!heavy_cols.empty()) {                                   // L'_m = L_m - D_m, D_m = one MSM over column m
    const HostField Q = fq_field(d->curve);
    const size_t fw = 2 * fq;                                  // u64 words of an affine point
    std::vector<HostJac> res(heavy_cols.size());
    std::vector<uint64_t> xyz(3 * fq);
    std::vector<uint64_t> cxy, csc;
    std::vector<uint8_t> cinf;
    for (size_t i = 0; i < heavy_cols.size(); i++) {
      const uint32_t m = heavy_cols[i];
      const size_t cn = hv[i].size();
      cxy.resize(cn * fw);
      csc.resize(cn * 4);
      cinf.resize(cn);
      for (size_t j = 0; j < cn; j++) {
        const uint32_t k = hv[i][j].first, e = hv[i][j].second;
        memcpy(&cxy[j * fw], &g_xy[(size_t)k * fw], 8 * fw);
        cinf[j] = g_inf[k];
        memcpy(&csc[j * 4], d->ct.coeff + (size_t)e * 4, 32);
      }
      std::vector<std::pair<uint32_t, uint32_t>>().swap(hv[i]);
      msm_var_run(ctx, d->curve, 1, cxy.data(), cinf.data(), csc.data(), cn, /*montgomery=*/true, xyz.data());
      HostJac D = host_jac_load(Q, xyz.data());
      D.y = Q.neg(D.y);
      HostJac Lm{};
      if ((*l_inf)[m]) {
        Lm.x = Q.one_();
        Lm.y = Q.one_();
      } else {
        memcpy(Lm.x.data(), l_xy->data() + (size_t)m * fw, 8 * fq);
        memcpy(Lm.y.data(), l_xy->data() + (size_t)m * fw + fq, 8 * fq);
        Lm.z = Q.one_();
      }
      res[i] = host_jac_add(Q, Lm, D);
    }
    std::vector<uint64_t> axy(heavy_cols.size() * fw);
    std::vector<uint8_t> ainf(heavy_cols.size());
    host_into_affine(Q, res, axy.data(), fw, ainf.data());
    for (size_t i = 0; i < heavy_cols.size(); i++) {
      memcpy(l_xy->data() + (size_t)heavy_cols[i] * fw, axy.data() + i * fw, 8 * fw);
      (*l_inf)[heavy_cols[i]] = ainf[i];
    }
  }
}

zkp_groth16_pk* groth16_pk_upload(zkp_ctx* ctx, const zkp_groth16_pk_desc* d, int rank, int world, int flags) {
  ZKP_REQUIRE(d->curve == ZKP_BN254 || d->curve == ZKP_BLS12_381, ZKP_ERR_UNSUPPORTED_CURVE);
  ZKP_REQUIRE(d->num_inputs >= 1, ZKP_ERR_BAD_ARG);
  ZKP_REQUIRE(world >= 0 && (world == 0 ? rank == 0 : (rank >= 0 && rank < world)), ZKP_ERR_BAD_ARG);
  std::unique_ptr<zkp_groth16_pk> pk(new zkp_groth16_pk());
  pk->shard_rank = rank;
  pk->shard_world = world;
  pk->curve = d->curve;
  pk->num_inputs = d->num_inputs;
  pk->num_aux = d->num_aux;
  pk->num_constraints = d->num_constraints;
  pk->nz = (size_t)d->num_inputs + d->num_aux;
  size_t dom = (size_t)d->num_constraints + d->num_inputs;       // r1cs_to_qap.rs:123-126
  int lg = 0;
  while (((size_t)1 << lg) < dom) lg++;
  ZKP_REQUIRE(lg <= (d->curve == ZKP_BN254 ? 28 : 32) && lg <= 30, ZKP_ERR_DOMAIN_TOO_LARGE);
  pk->log_n = lg;
  pk->N = (size_t)1 << lg;
  const bool matrices_only = d->a_query == nullptr;
  if (!matrices_only) {
    ZKP_REQUIRE(d->a_len == pk->nz && d->b_g1_len == pk->nz && d->b_g2_len == pk->nz, ZKP_ERR_BAD_ARG);
    ZKP_REQUIRE(d->l_len == d->num_aux, ZKP_ERR_BAD_ARG);
  }
  const size_t fq = d->curve == ZKP_BN254 ? 4 : 6;
  const uint64_t* tA[4] = {d->alpha_g1, d->delta_g1, nullptr, nullptr};
  const uint64_t* tB1[4] = {d->beta_g1, nullptr, d->delta_g1, nullptr};
  const uint64_t* tB2[4] = {d->beta_g2, nullptr, d->delta_g2, nullptr};
  const uint64_t* tL[4] = {nullptr, nullptr, nullptr, d->delta_g1};
  pk->m[0] = upload_csr(ctx, d->at, d->num_constraints);
  pk->m[1] = upload_csr(ctx, d->bt, d->num_constraints);
  pk->m[2] = upload_csr(ctx, d->ct, d->num_constraints);
  ZKP_REQUIRE(!(matrices_only && world > 0), ZKP_ERR_BAD_ARG);
  if (!matrices_only) {
    for (auto p : {d->alpha_g1, d->beta_g1, d->delta_g1, d->beta_g2, d->delta_g2}) ZKP_REQUIRE(p, ZKP_ERR_BAD_ARG);
    // slice of every query this rank keeps resident (everything for an unsharded key).  The H slice is cut from the
    // min(h_len, N) terms the MSM actually uses (prover.rs:186-187)
    const size_t h_used = std::min<size_t>(d->h_len, pk->N);
    shard_bounds(pk->nz + 4, rank, world, &pk->q_lo[0], &pk->q_n[0]);
    pk->q_lo[1] = pk->q_lo[2] = pk->q_lo[0];
    pk->q_n[1] = pk->q_n[2] = pk->q_n[0];
    shard_bounds(h_used, rank, world, &pk->q_lo[3], &pk->q_n[3]);
    // L is stored index-aligned with z (num_inputs leading identity points): its MSM then runs over the same scalar
    // slice as A / B1 / B2 and can reuse their bucket sort
    pk->q_lo[4] = pk->q_lo[0];
    pk->q_n[4] = pk->q_n[0];
    // Window-group size for the WHOLE key (msm.hip BasesEntry::lgk): the five queries are sized together so that they keep one
    // window configuration (sort sharing) and their tables fit ZKP_TABLE_BUDGET_GB / the free device memory.
    // ONE predicate for "the H query goes to evaluation form" (the planning block and the upload below used to evaluate two
    // different ones): flags bit 0 (zkp_groth16_pk_upload_ex, ZKP_PK_KEEP_FORM) keeps the key as given — no group transforms at upload
    // (one-shot and low-volume callers: the transforms cost ~0.65 s per 2^20 and pay back after ~2000 proofs), 7 transforms per proof;
    // zkp_ctx_config.h_evaluation_form / ZKP_H_LAGRANGE switches it per context.
    const bool lagrange_on = ctx->cfg.h_lagrange && !(flags & 1) && d->h_query && h_used > 0 && pk->log_n >= 1;
    int lgk = 0;
    {
      const int groups[5] = {1, 1, 2, 1, 1};
      // (the evaluation-form H query has N points, one more than the h_len = N - 1 the reference keeps: ADVICE r4; a sharded key
      //  re-shards it over N, so its slice is planned from N as well)
      size_t hq_lo = 0, hq_n = pk->q_n[3];
      if (lagrange_on) shard_bounds(pk->N, rank, world, &hq_lo, &hq_n);
      const size_t ns[5] = {pk->q_n[0], pk->q_n[1], pk->q_n[2], lagrange_on ? hq_n : (world > 0 ? pk->q_n[3] : (size_t)d->h_len), pk->q_n[4]};
      lgk = bases_plan_lgk(ctx, d->curve, groups, ns, 5);
      if (lgk > 0 && getenv("ZKP_DEBUG_MSM")) fprintf(stderr, "[groth16] window tables do not fit: window groups of %d\n", 1 << lgk);
    }
    std::vector<uint8_t> fA, fB1, fB2, fL;
    pk->hA = upload_ext(ctx, d->curve, 1, d->a_query, d->a_inf, d->a_len, 2 * fq, tA, pk->q_lo[0], pk->q_n[0], 0, &fA, 0, 0, lgk);
    // Window bits of the B queries (round 3).  A G2 bucket costs two Fq2 point additions in the reduction against 6 mixed
    // additions' worth of accumulate per entry, and half of a typical B query are identity points: with the c = round(log2 n) of
    // the G1 queries the B2 reduction (2^19 buckets) cost 0.70 ms per 2^20 proof beside 1.30 ms of accumulate.  Sized by the LIVE
    // bases, c = round(log2 live) - 2 (17 for the MiMC chain: 15 windows, 2^16 buckets, tasks of <= 32 entries so that the
    // accumulate kernel still fills the machine).  B1 takes the same configuration so that it keeps reusing B2's bucket sort.
    int cB = 0, capB = 0;
    {
      // Measured (2^20 MiMC chain, same box): B2 MSM 3.72 -> 3.27-3.45 ms standalone, but the B queries then leave the shared
      // level-1 pass (one more digit scan per proof) and their accumulate kernels run 18-30 % longer instead of the 15 % more
      // entries: 139.2 (c = 17), 139.8 (c = 18), 137.2 (c = 16) vs 139.9 proofs/s with the shared configuration.  Off by
      // default; ZKP_B_WINDOW=1 enables it (ZKP_B_WINDOW_BITS / ZKP_B_TASK_CAP override the choice).
      static const bool on = getenv("ZKP_B_WINDOW") && atoi(getenv("ZKP_B_WINDOW")) != 0;
      size_t live = 0;
      for (size_t i = 0; i < d->b_g2_len; i++) live += !(d->b_g2_inf && d->b_g2_inf[i]);
      int lg2 = 0;
      while (((size_t)2 << lg2) <= std::max<size_t>(live, 1)) lg2++;
      if (lg2 < 62 && (double)live >= 1.41421356 * (double)((size_t)1 << lg2)) lg2++;
      if (on && world == 0 && live >= ((size_t)1 << 14) && !ctx->cfg.msm_c && !ctx->cfg.msm_c_g2) {
        cB = std::max(12, std::min(20, lg2 - 2));
        if (const char* e = getenv("ZKP_B_WINDOW_BITS")) cB = atoi(e);
        capB = 32;
        if (const char* e = getenv("ZKP_B_TASK_CAP")) capB = atoi(e);
      }
    }
    pk->hB1 = upload_ext(ctx, d->curve, 1, d->b_g1_query, d->b_g1_inf, d->b_g1_len, 2 * fq, tB1, pk->q_lo[1], pk->q_n[1], 0, &fB1, cB, capB, lgk);
    pk->hB2 = upload_ext(ctx, d->curve, 2, d->b_g2_query, d->b_g2_inf, d->b_g2_len, 4 * fq, tB2, pk->q_lo[2], pk->q_n[2], 0, &fB2, cB, capB, lgk);
    {
      // B1 reuses B2's bucket sort + task schedule (same scalars, window configuration and identity pattern): -0.55 ms of
      // memory-bound sort kernels per proof.  With 4 hardware queues this LOST 2 % (84.7 -> 83.0 proofs/s: the wait on
      // B2's stream idled a queue); with 16 queues it gains 1-2 % (97.9 -> 99.3).  ZKP_SHARE_B_SORT=0 disables it.
      static const bool on = !(getenv("ZKP_SHARE_B_SORT") && atoi(getenv("ZKP_SHARE_B_SORT")) == 0);
      const bool same_inf = (!d->b_g1_inf && !d->b_g2_inf) ||
                            (d->b_g1_inf && d->b_g2_inf && memcmp(d->b_g1_inf, d->b_g2_inf, d->b_g1_len) == 0);
      pk->share_b_sort = on && same_inf && bases_same_shape(ctx, pk->hB1, pk->hB2);
    }
    if (lagrange_on) {
      LagrangeCache& cache = lagrange_cache;
      if (!(lagrange_keep_cache && cache.q == d->h_query && cache.used == h_used && cache.log_n == pk->log_n && cache.curve == d->curve)) {
        lagrange_h(ctx, d->curve, d->h_query, d->h_inf, h_used, pk->log_n, &cache.xy, &cache.inf);
        cache.q = d->h_query;
        cache.used = h_used;
        cache.log_n = pk->log_n;
        cache.curve = d->curve;
      }
      shard_bounds(pk->N, rank, world, &pk->q_lo[3], &pk->q_n[3]);
      pk->hH = bases_upload(ctx, d->curve, 1, cache.xy.data() + pk->q_lo[3] * 2 * fq, cache.inf.data() + pk->q_lo[3],
                            pk->q_n[3], 0, 0, lgk);
      pk->h_lagrange = true;
      if (!lagrange_keep_cache) cache.clear();
    } else
    pk->hH = bases_upload(ctx, d->curve, 1, d->h_query ? d->h_query + pk->q_lo[3] * 2 * fq : nullptr,
                          d->h_inf ? d->h_inf + pk->q_lo[3] : nullptr, world > 0 ? pk->q_n[3] : (size_t)d->h_len, 0, 0, lgk);
    if (pk->h_lagrange && ctx->cfg.c_fold && d->l_query) {                // zkp_ctx_config.c_fold / ZKP_C_FOLD
      std::vector<uint64_t> lxy;
      std::vector<uint8_t> linf;
      LagrangeCache& gc = lagrange_cache_g;
      if (!(lagrange_keep_cache && gc.q == d->h_query && gc.used == h_used && gc.log_n == pk->log_n && gc.curve == d->curve)) {
        lagrange_h(ctx, d->curve, d->h_query, d->h_inf, h_used, pk->log_n, &gc.xy, &gc.inf, 1);
        gc.q = d->h_query;
        gc.used = h_used;
        gc.log_n = pk->log_n;
        gc.curve = d->curve;
      }
      fold_c_into_l(ctx, d, pk->nz, gc.xy, gc.inf, &lxy, &linf);          // the whole L'; upload_ext keeps this rank's slice
      if (!lagrange_keep_cache) gc.clear();
      pk->hL = upload_ext(ctx, d->curve, 1, lxy.data(), linf.data(), pk->nz, 2 * fq, tL, pk->q_lo[4], pk->q_n[4], 0, &fL, 0, 0, lgk);
      pk->c_folded = true;
    } else
    pk->hL = upload_ext(ctx, d->curve, 1, d->l_query, d->l_inf, d->l_len, 2 * fq, tL, pk->q_lo[4], pk->q_n[4],
                        d->num_inputs, &fL, 0, 0, lgk);
    {
      // L reuses A's bucket sort + task schedule (L is stored index-aligned with z, so both MSMs run over the same scalar
      // slice): one digit scan + level-2 sort + schedule less per proof.  The shared sort drops a base only if it is the
      // identity in BOTH queries (A's scan then uses the intersection of the two flag vectors, bases_set_sort_flags); a base
      // that is the identity in one query only stays in the list and the accumulate kernel skips it when it gathers it.
      // That is only worth it when such bases are rare: a lane whose point is the identity idles while its wave-mates add
      // (measured: B2 sharing a full-pattern sort ran 4.47 instead of 1.91 ms — half of the MiMC chain's B-query is
      // identities).  In the MiMC chain A and L each have ~210 K identity bases of 1.26 M that the other does not: sharing
      // made both accumulate kernels 20 % longer (1.02 -> 1.25 ms) for one sort less, 122.2 vs 121.0 proofs/s and + 0.2 ms
      // single-proof latency — not taken: the limit is 1/16 of the bases.
      // ZKP_SHARE_AL_SORT=0 disables it.
      static const bool on = !(getenv("ZKP_SHARE_AL_SORT") && atoi(getenv("ZKP_SHARE_AL_SORT")) == 0);
      bool same = fA.size() == fL.size() && pk->q_lo[0] == pk->q_lo[4];
      size_t differ = 0;
      std::vector<uint8_t> both(fA.size());
      for (size_t k = 0; same && k < fA.size(); k++) {
        both[k] = fA[k] & fL[k];
        differ += fA[k] != fL[k];
      }
      pk->share_al_sort = on && same && differ <= fA.size() / 16 && bases_same_shape(ctx, pk->hL, pk->hA) && pk->q_n[0] > 0;
      if (pk->share_al_sort && differ) bases_set_sort_flags(ctx, pk->hA, both.data(), both.size());
      // One level-1 pass over z for A, B2 (whose sort B1 reuses) and L: the digit scan, the (bin, tile) counts and the scatter
      // into bins are done once, by A, over the bases that are NOT the identity in all three queries; every query then
      // runs its own level-2 sort and drops its own identities there (three flag bits in the top of every entry, bases_set_group), so
      // each accumulate kernel still sees exactly its own entries.  Two of the four digit scans of a proof disappear
      // (ZKP_SHARE_L1=0 disables it).
      static const bool on_l1 = !(getenv("ZKP_SHARE_L1") && atoi(getenv("ZKP_SHARE_L1")) == 0);
      const bool aligned = fA.size() == fL.size() && fA.size() == fB2.size() && pk->q_lo[0] == pk->q_lo[4] &&
                           pk->q_lo[0] == pk->q_lo[2] && pk->q_n[0] > 0;
      if (on_l1 && aligned && !pk->share_al_sort && bases_same_shape(ctx, pk->hL, pk->hA)) {
        // (hB2 is a G2 table: same n, window configuration checked through hB1 / share_b_sort).  B queries with their own
        // window configuration (above) run their own level-1 pass: the group is then A and L only.
        const bool with_b = pk->share_b_sort && bases_same_shape(ctx, pk->hB1, pk->hA);
        std::vector<uint8_t> all(fA.size()), member(fA.size());
        for (size_t k = 0; k < fA.size(); k++) {
          all[k] = fA[k] & fL[k] & (with_b ? fB2[k] : (uint8_t)1);
          member[k] = all[k] ? 0 : (uint8_t)((fA[k] ? 1 : 0) | (with_b && fB2[k] ? 2 : 0) | (fL[k] ? 4 : 0));
        }
        bases_set_sort_flags(ctx, pk->hA, all.data(), all.size());
        bases_set_group(ctx, pk->hA, member.data(), member.size());
        bases_set_filter_bit(ctx, pk->hA, 0);
        if (with_b) bases_set_filter_bit(ctx, pk->hB2, 1);
        bases_set_filter_bit(ctx, pk->hL, 2);
        pk->share_l1 = true;
        pk->b_in_l1 = with_b;
      }
      {
        // C only ever needs l' + h_acc (prover.rs:189-196): with equal bucket ranges and no window groups H's accumulate kernel
        // continues from L's finished buckets and ONE reduction yields the sum; L's own result is the identity (ZKP_CHAIN_LH=0: off)
        static const bool on_chain = !(getenv("ZKP_CHAIN_LH") && atoi(getenv("ZKP_CHAIN_LH")) == 0);
        uint64_t iL[5], iH[5];
        bases_info(ctx, pk->hL, iL);
        bases_info(ctx, pk->hH, iH);
        pk->chain_lh = on_chain && iL[0] == iH[0] && iL[2] == 1 && iH[2] == 1 && pk->q_n[3] > 0 && pk->q_n[4] > 0;
      }
      if (getenv("ZKP_DEBUG_MSM"))
        fprintf(stderr, "[groth16] A/L sort sharing: %d (flags differ at %zu of %zu bases), B1/B2: %d, shared level 1: %d\n",
                (int)pk->share_al_sort, differ, fA.size(), (int)pk->share_b_sort, (int)pk->share_l1);
    }
  }
  uint32_t* consts = pk->consts.as<uint32_t>(64);
  if (d->curve == ZKP_BN254) hipLaunchKernelGGL(qap_consts_kernel<Bn254Fr>, dim3(1), dim3(64), 0, ctx->cur->stream, consts, lg);
  else hipLaunchKernelGGL(qap_consts_kernel<Bls381Fr>, dim3(1), dim3(64), 0, ctx->cur->stream, consts, lg);
  ZKP_HIP(hipGetLastError());
  for (int l = 0; l < 2; l++) {                     // further lanes allocate on first use
    auto& L = pk->lane[l];
    L.abc.get(3 * pk->N * 32);
    L.S.get((pk->nz + 4) * 32);
    L.results.get(6 * 16 * 24 * 4 + 64);
    L.proof.get(4096);
  }
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  return pk.release();
}

void groth16_pk_free(zkp_ctx* ctx, zkp_groth16_pk* pk) {
  (void)hipDeviceSynchronize();
  for (auto& PL : pk->lane) {
    if (PL.graph_exec) (void)hipGraphExecDestroy(PL.graph_exec);
    if (PL.graph) (void)hipGraphDestroy(PL.graph);
  }
  for (uint64_t h : {pk->hA, pk->hB1, pk->hB2, pk->hH, pk->hL})
    if (h) bases_drop(ctx, h);
  for (auto& m : pk->m) {
    if (m.row_ptr) (void)hipFree(m.row_ptr);
    if (m.col) (void)hipFree(m.col);
    if (m.coeff) (void)hipFree(m.coeff);
  }
  delete pk;
}

uint64_t groth16_domain_size(zkp_groth16_pk* pk) { return pk->N; }
void groth16_pk_info(zkp_ctx* ctx, zkp_groth16_pk* pk, uint64_t info[8]) {
  for (int i = 0; i < 8; i++) info[i] = 0;
  if (!pk->hA) return;
  uint64_t a[5], b[5];
  bases_info(ctx, pk->hA, a);
  bases_info(ctx, pk->hB2, b);
  info[0] = a[2];
  for (uint64_t h : {pk->hA, pk->hB1, pk->hB2, pk->hH, pk->hL}) {
    uint64_t t[5];
    bases_info(ctx, h, t);
    info[1] += t[4];
  }
  info[2] = a[0];
  info[3] = a[1];
  info[4] = a[3];
  info[5] = b[0];
  info[6] = (pk->share_b_sort ? 1 : 0) | (pk->share_al_sort ? 2 : 0) | (pk->share_l1 ? 4 : 0);
  info[7] = (pk->h_lagrange ? 1 : 0) | (pk->c_folded ? 2 : 0) | (pk->chain_lh ? 4 : 0);       // key form (slots L / H)
}

// z_dev: nz Fr (device).  Leaves h (N Fr, Montgomery) in pk->abc[0..N)
// coeffs: h in coefficient form wanted (zkp_groth16_witness_map; the prover of a coefficient-form key); false with an evaluation-form
// key: the pointwise values v on the coset, the scalars of its H MSM
template <class P>
static uint32_t* witness_map_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint32_t* z_dev, bool coeffs = true) {
  const uint32_t N = (uint32_t)pk->N;
  uint32_t* a = pk->lane[ctx->cur_idx].abc.as<uint32_t>(3 * pk->N * 8);
  uint32_t* b = a + pk->N * 8;
  uint32_t* c = b + pk->N * 8;
  uint32_t* bufs[3] = {a, b, c};
  hipStream_t st = ctx->cur->stream;
  const int nchain = (!coeffs && pk->c_folded) ? 2 : 3;            // C folded into the L query: no C z, no c chain
  for (int k = 0; k < nchain; k++)
    hipLaunchKernelGGL(csr_eval_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, st, pk->m[k].row_ptr, pk->m[k].col,
                       pk->m[k].coeff, z_dev, pk->num_constraints, N, pk->num_inputs, k == 0 ? 1 : 0, bufs[k]);
  // One launch for a, b, c (grid.y = 3) shortens the witness map in isolation (1.49 -> 1.35 ms at 2^20) but its 3x larger
  // launches delay the MSM streams of the other lane: 79 vs 85 proofs/s pipelined.  Off unless ZKP_NTT_BATCH=1.
  static const bool batch = getenv("ZKP_NTT_BATCH") && atoi(getenv("ZKP_NTT_BATCH")) != 0;
  if (batch) {
    ntt_run_batch(ctx, pk->curve, bufs, nchain, pk->log_n, ZKP_NTT_IFFT);
    ntt_run_batch(ctx, pk->curve, bufs, nchain, pk->log_n, ZKP_NTT_COSET_FFT);
  } else {
    for (int k = 0; k < nchain; k++) {
      // ifft -> coset_fft as one chain of passes (1/N folded into the coset table, no odd-pass copies); falls back to the two
      // separate transforms above the full-table domain limit
      if (!ntt_ifft_coset_fft(ctx, pk->curve, bufs[k], pk->log_n)) {
        ntt_run(ctx, pk->curve, bufs[k], pk->log_n, ZKP_NTT_IFFT);
        ntt_run(ctx, pk->curve, bufs[k], pk->log_n, ZKP_NTT_COSET_FFT);
      }
    }
  }
  if (!coeffs) {
    if (pk->c_folded) hipLaunchKernelGGL(qap_pointwise_ab_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, st, a, b, pk->consts.as<uint32_t>(64), N);
    else hipLaunchKernelGGL(qap_pointwise_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, st, a, b, c, pk->consts.as<uint32_t>(64), N);
    ZKP_HIP(hipGetLastError());
    return a;
  }
  // (a*b - c) / Z(g) fused into the first pass of coset_ifft; h lands in a or b
  if (uint32_t* hq = ntt_qap_coset_ifft(ctx, pk->curve, a, b, c, pk->consts.as<uint32_t>(64), pk->log_n)) {
    ZKP_HIP(hipGetLastError());
    return hq;
  }
  hipLaunchKernelGGL(qap_pointwise_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, st, a, b, c,
                     pk->consts.as<uint32_t>(64), N);
  ntt_run(ctx, pk->curve, a, pk->log_n, ZKP_NTT_COSET_IFFT);
  ZKP_HIP(hipGetLastError());
  return a;
}

void groth16_witness_map(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, uint64_t* h, bool on_device) {
  const uint32_t* zd = reinterpret_cast<const uint32_t*>(z);
  if (!on_device) {
    uint32_t* s = pk->lane[ctx->cur_idx].S.as<uint32_t>((pk->nz + 4) * 8);
    ZKP_HIP(hipMemcpyAsync(s, z, pk->nz * 32, hipMemcpyHostToDevice, ctx->cur->stream));
    zd = s;
  }
  uint32_t* hd = pk->curve == ZKP_BN254 ? witness_map_dev<Bn254Fr>(ctx, pk, zd) : witness_map_dev<Bls381Fr>(ctx, pk, zd);
  ZKP_HIP(hipMemcpyAsync(h, hd, pk->N * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->cur->stream));
  if (!on_device) ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
}

// Enqueue one proof on the current lane (ctx->cur): everything is asynchronous up to and including the read-back of
// the 3 proof points into the lane's pinned host buffer; prove_finish() synchronises the lane and hands them out.
// part: 1 = copy the inputs (z, r, s) into the lane's buffers, 2 = everything else (the part a hipGraph captures), 3 = both
template <class FrP>
static void prove_enqueue_part(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, bool z_on_device, const uint64_t* r,
                               const uint64_t* s, int part, char* partial_out = nullptr, bool latency_plan = false) {
  // partial_out != nullptr (base-sharded key): no assembly; the 5 XYZZ sums of this rank's slices (A | B1 | B2 | H | L,
  // G2-sized slots) are left at partial_out (device) for the all-gather
  ZKP_REQUIRE((pk->shard_world > 0) == (partial_out != nullptr), ZKP_ERR_BAD_ARG);
  ZKP_REQUIRE(pk->hA != 0, ZKP_ERR_BAD_ARG);             // matrices-only key: use the sharded path
  zkp_groth16_pk::PerLane& PL = pk->lane[ctx->cur_idx];
  hipStream_t st = ctx->cur->stream;
  const bool prof = ctx->profiling;
  // zkp_ctx_config.host_affine = ZKP_OFF / ZKP_HOST_AFFINE=0: into_affine of the proof points on the device (rounds 1-3) instead of the host
  const bool host_tail = ctx->cfg.host_affine && !partial_out;
  zkp_groth16_timing tm{};
  struct ProfEvents {                                   // destroyed on every exit path (msm_run may throw)
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    ~ProfEvents() {
      for (hipEvent_t x : e)
        if (x) (void)hipEventDestroy(x);
    }
  } pev;
  if (prof)
    for (hipEvent_t& x : pev.e) ZKP_HIP(hipEventCreate(&x));
  hipEvent_t &e0 = pev.e[0], &e1 = pev.e[1], &eT0 = pev.e[2], &eT1 = pev.e[3];
  auto tic = [&] { if (prof) ZKP_HIP(hipEventRecord(e0, st)); };
  auto toc = [&](float* dst) {
    if (!prof) return;
    ZKP_HIP(hipEventRecord(e1, st));
    ZKP_HIP(hipEventSynchronize(e1));
    ZKP_HIP(hipEventElapsedTime(dst, e0, e1));
  };
  if (prof) ZKP_HIP(hipEventRecord(eT0, st));
  uint32_t* S = PL.S.as<uint32_t>((pk->nz + 4) * 8);
  uint32_t* rs = PL.proof.as<uint32_t>(1024);         // [r, s] then proof words then flags
  uint32_t* proof_dev = rs + 16;
  uint32_t* flags_dev = proof_dev + 256;
  if (part & 1) {
    ZKP_HIP(hipMemcpyAsync(S, z, pk->nz * 32, z_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    ZKP_HIP(hipMemcpyAsync(rs, r, 32, hipMemcpyHostToDevice, st));
    ZKP_HIP(hipMemcpyAsync(rs + 8, s, 32, hipMemcpyHostToDevice, st));
  }
  if (!(part & 2)) return;
  hipLaunchKernelGGL(scalar_tail_kernel<FrP>, dim3(1), dim3(64), 0, st, S + pk->nz * 8, rs);

  const MsmVtbl* v1 = msm_vtbl(pk->curve, 1);
  const MsmVtbl* v2 = msm_vtbl(pk->curve, 2);
  const size_t slot = v2->xyzz_bytes;                  // uniform slot size
  char* res = reinterpret_cast<char*>(PL.results.get(6 * slot));
  const uint64_t* Sd = reinterpret_cast<const uint64_t*>(S);
  bool l_done_in_fan = false, g2_done_in_fan = false, chained = false;
  float acc_ms = 0.f, scan_ms = 0.f;
  uint64_t ent = 0, scan_bytes = 0, scan_runs = 0;
  // Schedule.  profiling: everything on the main stream, one MSM at a time, with per-phase events.
  // otherwise: three streams —  main: witness_map -> H ;  ws1: A -> L ;  ws2: B1 -> B2  — joined before assembly.
  // ZKP_SINGLE_STREAM=1: one stream per proof (no fan-out inside a proof); concurrency then comes from the lanes only
  static const bool single_stream = getenv("ZKP_SINGLE_STREAM") && atoi(getenv("ZKP_SINGLE_STREAM")) != 0;
  const bool fan = !prof && !single_stream;
  uint32_t* h = nullptr;
  auto run = [&](int idx, uint64_t handle, const uint64_t* sc, size_t n, int w, int sort_src = -1, int l1_src = -1) {
    float ms = 0.f, ms_sc = 0.f;
    uint64_t e = 0;
    static const char* const names[5] = {"A", "B1", "B2", "H", "L"};
    ctx->tl_tag = names[idx];
    tic();
#ifdef ZKP_ABLATION   // timing ablation builds only (ZKP_BUILD_DEFS=-DZKP_ABLATION -> variants/<tag>/): WRONG proofs, never in the shipped library
    static const int skip_k8 = [] { const char* e = getenv("ZKP_DEBUG_SKIP_K8_MASK"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
    ctx->dbg_skip_k8 = ((skip_k8 >> idx) & 1) && ctx->batch_mode;
#endif
    msm_run(ctx, handle, 0, sc, n, true, nullptr, res + idx * slot, prof ? &ms : nullptr, &e, fan ? w : 0,
            fan ? sort_src : -1, prof ? &ms_sc : nullptr, fan ? l1_src : -1);
    ctx->dbg_skip_k8 = false;
    toc(&tm.ms_msm[idx]);
    acc_ms += ms;
    ent += e;
    tm.ms_msm_acc[idx] = ms;
    tm.msm_entries[idx] = e;
    if (ms_sc > 0.f) {                       // algorithmic bytes of the scalar scan: scalars read by both passes + one 8-B word per entry
      scan_ms += ms_sc;
      scan_bytes += 2 * 32 * (uint64_t)n + 8 * e;
      scan_runs += 1;
    }
    tm.msm_accumulate_launches += 1;
  };
  ctx->tl_tag.clear();
  ctx->mark(st, "start");
  if (fan) {
    ZKP_HIP(hipEventRecord(ctx->cur->ev_fork, st));                       // S is complete
    for (int w = 1; w < zkp_ctx::N_WS; w++) ZKP_HIP(hipStreamWaitEvent(ctx->cur->ws[w].stream, ctx->cur->ev_fork, 0));
    // The witness map heads the longest chain of a proof (witness map -> H MSM -> assembly): its launches go out FIRST, before
    // the ~150 launches of the four other MSMs (0.5 ms of host time): single-proof latency 10.1 -> 9.65 ms (ZKP_WM_FIRST=0: after
    // them, as in round 1).  Making the other MSMs' accumulate kernels wait for it — a kernel timeline shows machine-filling
    // accumulates from three streams leaving its kernels few wave slots for milliseconds — was measured too: 10.3 ms, not kept.
    static const bool wm_first = !(getenv("ZKP_WM_FIRST") && atoi(getenv("ZKP_WM_FIRST")) == 0);
    if (wm_first) {
      h = witness_map_dev<FrP>(ctx, pk, S, !pk->h_lagrange);
      ctx->mark(st, "wm");
    }
    // (Round 4, measured with ZKP_TIMELINE=1 and removed again: holding the accumulate kernels of A / B1 / B2 / L until the witness map
    //  is done — the map ends at 5.4 ms of an 8.2 ms proof when they start at 0.9 ms — and giving L a workspace of its own so that its
    //  accumulate need not wait for A's reduction chain: 8.6-8.8 / 8.6-8.9 / 9.3-9.6 ms (gate / workspace / both) against 8.0-8.4.  The
    //  machine-filling accumulates only change places; what ends last is still one accumulate + its reduction.  A third plan — main: map ->
    //  H (sorted, accumulated, reduction deferred) -> L on top of H's buckets; B2 at once; A / B1 accumulates held until H is sorted —
    //  proved correct and ran 9.2-9.7 ms.  profiles/r04_latency_experiments.txt.)
    static const bool l_own = !(getenv("ZKP_L_OWN_STREAM") && atoi(getenv("ZKP_L_OWN_STREAM")) == 0);
    static const int lat_env = [] { const char* e = getenv("ZKP_LATENCY_PLAN"); return e ? atoi(e) : -1; }();
    (void)latency_plan;
    const bool lat = lat_env >= 0 ? lat_env != 0 : true;
    if (lat) {
      // Stream plan (round 2; default for single proofs AND the pipelined batch — measured 113.6 vs 111.2 proofs/s and 11.0 vs
      // 11.9 ms single-proof latency against the round-1 plan below, which ZKP_LATENCY_PLAN=0 restores):
      //   ws2: B2 | ws1: A -> L | ws3: B1 (B2's sort), then s*g_a + r*g1_b as soon as A exists | main: witness_map -> H
      // (A is enqueued first: with a shared level-1 pass B2's stream waits on an event that A's stream must have recorded)
      const int l1 = pk->share_l1 ? 1 : -1;
      const bool b_l1 = pk->share_l1 && pk->b_in_l1;
      if (!b_l1) run(2, pk->hB2, Sd + 4 * pk->q_lo[2], pk->q_n[2], 2);
      run(0, pk->hA, Sd + 4 * pk->q_lo[0], pk->q_n[0], 1);
      ZKP_HIP(hipEventRecord(ctx->cur->ev_a, ctx->cur->ws[1].stream));
      if (b_l1) run(2, pk->hB2, Sd + 4 * pk->q_lo[2], pk->q_n[2], 2, -1, l1);
      // proof.b needs B2 only: its into_affine runs on B2's stream as soon as the MSM is done instead of in the tail of the proof
      static const bool g2_early = !(getenv("ZKP_G2_EARLY") && atoi(getenv("ZKP_G2_EARLY")) == 0);
      if (!partial_out && g2_early && !host_tail) {
        v2->assemble_g2(ctx->cur->ws[2].stream, res, slot, proof_dev, flags_dev, 2 * v1->fN);
        g2_done_in_fan = true;
      }
      run(1, pk->hB1, Sd + 4 * pk->q_lo[1], pk->q_n[1], 3, pk->share_b_sort ? 2 : -1);
      ctx->msm_defer_reduce = chained = pk->chain_lh;
      run(4, pk->hL, Sd + 4 * pk->q_lo[4], pk->q_n[4], 1, pk->share_al_sort ? 1 : -1, l1);   // A's sort / level-1 pass, still in this workspace
      if (!partial_out) {
        ZKP_HIP(hipStreamWaitEvent(ctx->cur->ws[3].stream, ctx->cur->ev_a, 0));
        v1->assemble_g1_part1(ctx->cur->ws[3].stream, res, slot, rs, host_tail ? nullptr : proof_dev, flags_dev);
        ctx->tl_tag.clear();
        ctx->mark(ctx->cur->ws[3].stream, "part1");
      }
    } else {
    // stream plan (longest chain first): ws2: B2 | ws1: A -> B1 | main: witness_map -> H | ws3: L, then part 1 after A, B1
    // (4 lanes x 4 streams = 16 streams = one hardware queue each under GPU_MAX_HW_QUEUES=16)
    run(2, pk->hB2, Sd + 4 * pk->q_lo[2], pk->q_n[2], 2);                          // prover.rs:182-184
    run(0, pk->hA, Sd + 4 * pk->q_lo[0], pk->q_n[0], 1);                           // prover.rs:164-167
    run(1, pk->hB1, Sd + 4 * pk->q_lo[1], pk->q_n[1], 1, pk->share_b_sort ? 2 : -1);   // prover.rs:170-177 (B2's bucket sort reused)
    // (A's sort is reused only when B1 does not re-sort on the same workspace in the meantime, i.e. when B1 takes B2's sort)
    if (l_own) run(4, pk->hL, Sd + 4 * pk->q_lo[4], pk->q_n[4], 3, pk->share_al_sort && pk->share_b_sort ? 1 : -1);   // prover.rs:189-190
    ZKP_HIP(hipEventRecord(ctx->cur->ev_b1, ctx->cur->ws[1].stream));
    // the two dynamic scalar multiplications (s*g_a, r*g1_b) are a ~2 ms single-lane chain each: start them as soon as
    // A and B1 exist so they hide under the remaining MSMs
    if (!partial_out) {
      ZKP_HIP(hipStreamWaitEvent(ctx->cur->ws[3].stream, ctx->cur->ev_b1, 0));
      v1->assemble_g1_part1(ctx->cur->ws[3].stream, res, slot, rs, host_tail ? nullptr : proof_dev, flags_dev);
    }
    }
    l_done_in_fan = lat || l_own;
  }
  tic();
  if (!h) h = witness_map_dev<FrP>(ctx, pk, S, !pk->h_lagrange);
  toc(&tm.ms_witness_map);
  if (!fan) {
    run(0, pk->hA, Sd + 4 * pk->q_lo[0], pk->q_n[0], 0);
    run(1, pk->hB1, Sd + 4 * pk->q_lo[1], pk->q_n[1], 0);
    run(2, pk->hB2, Sd + 4 * pk->q_lo[2], pk->q_n[2], 0);
  }
  if (chained) ctx->msm_acc_into = 1;                 // on top of L's buckets (workspace 1): result slot 3 = h_acc + l', slot 4 = identity
  run(3, pk->hH, reinterpret_cast<const uint64_t*>(h) + 4 * pk->q_lo[3], pk->q_n[3], 0);  // :186-187 (min(len) truncation in q_n)
  if (!(fan && l_done_in_fan)) run(4, pk->hL, Sd + 4 * pk->q_lo[4], pk->q_n[4], 0);                                 // :189-190
  if (fan) {
    for (int w = 1; w < zkp_ctx::N_WS; w++) {
      ZKP_HIP(hipEventRecord(ctx->cur->ws[w].done, ctx->cur->ws[w].stream));
      ZKP_HIP(hipStreamWaitEvent(st, ctx->cur->ws[w].done, 0));
    }
  }
  tm.ms_msm_accumulate = acc_ms;
  tm.msm_points = ent;
  tm.ms_msm_scan = scan_ms;
  tm.msm_scan_bytes = scan_bytes;
  tm.msm_scan_launches = scan_runs;
  if (partial_out) {
    ZKP_HIP(hipMemcpyAsync(partial_out, res, 5 * slot, hipMemcpyDeviceToDevice, st));
    ZKP_HIP(hipGetLastError());
    if (prof) {
      ZKP_HIP(hipEventRecord(eT1, st));
      ZKP_HIP(hipStreamSynchronize(st));
      ZKP_HIP(hipEventElapsedTime(&tm.ms_total, eT0, eT1));
      ctx->last_timing = tm;
    }
    return;
  }

  tic();
  // proof layout (32-bit words): A = 2*fN1 | B = 2*fN2 | C = 2*fN1
  if (!fan) v1->assemble_g1_part1(st, res, slot, rs, host_tail ? nullptr : proof_dev, flags_dev);
  v1->assemble_g1_part2(st, res, slot, host_tail ? nullptr : proof_dev, flags_dev, 2 * v1->fN + 2 * v2->fN);
  if (!g2_done_in_fan && !host_tail) v2->assemble_g2(st, res, slot, proof_dev, flags_dev, 2 * v1->fN);
  ZKP_HIP(hipGetLastError());
  toc(&tm.ms_assemble);
  const size_t proof_words = 4 * (size_t)v1->fN + 2 * (size_t)v2->fN;
  if (host_tail) {
    // the three proof points go back as XYZZ (A = slot 0, B = slot 2, C = slot 5); prove_finish makes them affine on the host
    // with one inversion (host_field.hpp) — three single-lane Fermat chains (0.25-0.32 ms each) leave the tail of the proof
    uint32_t* hp = ctx->cur->host_proof;
    ZKP_HIP(hipMemcpyAsync(hp, res, 16 * (size_t)v1->fN, hipMemcpyDeviceToHost, st));
    ZKP_HIP(hipMemcpyAsync(hp + 4 * v1->fN, res + 2 * slot, 16 * (size_t)v2->fN, hipMemcpyDeviceToHost, st));
    ZKP_HIP(hipMemcpyAsync(hp + 4 * v1->fN + 4 * v2->fN, res + 5 * slot, 16 * (size_t)v1->fN, hipMemcpyDeviceToHost, st));
  } else {
    ZKP_HIP(hipMemcpyAsync(ctx->cur->host_proof, proof_dev, proof_words * 4, hipMemcpyDeviceToHost, st));
    ZKP_HIP(hipMemcpyAsync(ctx->cur->host_proof + 256, flags_dev, 12, hipMemcpyDeviceToHost, st));
  }
  ctx->tl_tag.clear();
  ctx->mark(st, "copied");
  ctx->cur->host_tail = host_tail;
  ctx->cur->busy = true;
  if (prof) {
    ZKP_HIP(hipEventRecord(eT1, st));
    ZKP_HIP(hipStreamSynchronize(st));
    ZKP_HIP(hipEventElapsedTime(&tm.ms_total, eT0, eT1));
    ctx->last_timing = tm;
  }
}

// every device pointer a captured proof embeds: if any scratch buffer was reallocated since the capture (another, larger
// key used the lane), the graph is stale
static std::vector<const void*> lane_signature(zkp_ctx* ctx, zkp_groth16_pk* pk) {
  std::vector<const void*> sig;
  zkp_lane* L = ctx->cur;
  zkp_groth16_pk::PerLane& PL = pk->lane[ctx->cur_idx];
  for (DevBuf* b : {&PL.abc, &PL.S, &PL.results, &PL.proof, &L->ntt_scratch}) sig.push_back(b->p);
  for (int w = 0; w < zkp_lane::N_WS; w++) {
    MsmWorkspace& ws = L->ws[w];
    for (DevBuf* b : {&ws.keys, &ws.vals, &ws.keys2, &ws.vals2, &ws.sort_tmp, &ws.offsets, &ws.buckets, &ws.tmp, &ws.out,
                      &ws.sched, &ws.scan_tmp, &ws.scan_tmp2, &ws.partial, &ws.redo})
      sig.push_back(b->p);
  }
  sig.push_back(pk);
  return sig;
}

// One proof on the current lane.  With ZKP_GRAPH=1 the ~250 launches of a proof (4 streams, fork/join by events) are
// captured once per (key, lane) into a hipGraph and replayed; the inputs are copied in front of the graph launch.
template <class FrP>
static void prove_enqueue(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, bool z_on_device, const uint64_t* r,
                          const uint64_t* s, bool latency_plan = false) {
  static const bool graph_on = getenv("ZKP_GRAPH") && atoi(getenv("ZKP_GRAPH")) != 0;
  static const bool single_stream = getenv("ZKP_SINGLE_STREAM") && atoi(getenv("ZKP_SINGLE_STREAM")) != 0;
  if (!graph_on || ctx->profiling || single_stream) {
    prove_enqueue_part<FrP>(ctx, pk, z, z_on_device, r, s, 3, nullptr, latency_plan);
    return;
  }
  // (graph replay always uses the throughput plan: a captured graph embeds its stream plan)
  zkp_groth16_pk::PerLane& PL = pk->lane[ctx->cur_idx];
  hipStream_t st = ctx->cur->stream;
  if (PL.graph_state == 2 && PL.sig == lane_signature(ctx, pk)) {
    prove_enqueue_part<FrP>(ctx, pk, z, z_on_device, r, s, 1);
    ZKP_HIP(hipGraphLaunch(PL.graph_exec, st));
    ctx->cur->busy = true;
    return;
  }
  if (PL.graph_state == 2) {                              // stale
    (void)hipGraphExecDestroy(PL.graph_exec);
    (void)hipGraphDestroy(PL.graph);
    PL.graph_exec = nullptr;
    PL.graph = nullptr;
    PL.graph_state = 0;
  }
  if (PL.graph_state == 1 && PL.sig == lane_signature(ctx, pk)) {
    prove_enqueue_part<FrP>(ctx, pk, z, z_on_device, r, s, 1);
    ZKP_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    try {
      prove_enqueue_part<FrP>(ctx, pk, z, z_on_device, r, s, 2);
    } catch (...) {
      hipGraph_t g = nullptr;
      (void)hipStreamEndCapture(st, &g);
      if (g) (void)hipGraphDestroy(g);
      throw;
    }
    ZKP_HIP(hipStreamEndCapture(st, &PL.graph));
    ZKP_HIP(hipGraphInstantiate(&PL.graph_exec, PL.graph, nullptr, nullptr, 0));
    PL.graph_state = 2;
    ZKP_HIP(hipGraphLaunch(PL.graph_exec, st));
    ctx->cur->busy = true;
    return;
  }
  prove_enqueue_part<FrP>(ctx, pk, z, z_on_device, r, s, 3);    // eager run: allocates every scratch buffer
  PL.sig = lane_signature(ctx, pk);
  PL.graph_state = 1;
}

static void prove_finish(zkp_ctx* ctx, zkp_groth16_pk* pk, uint64_t* proof_out, uint8_t* inf_out) {
  const MsmVtbl* v1 = msm_vtbl(pk->curve, 1);
  const MsmVtbl* v2 = msm_vtbl(pk->curve, 2);
  const size_t proof_words = 4 * (size_t)v1->fN + 2 * (size_t)v2->fN;
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  if (!ctx->tl.empty()) {
    std::string line = "[zkp timeline ms]";
    for (auto& m : ctx->tl) {
      float ms = 0.f;
      (void)hipEventSynchronize(m.second);
      (void)hipEventElapsedTime(&ms, ctx->tl[0].second, m.second);
      char buf[64];
      snprintf(buf, sizeof buf, " %s=%.2f", m.first.c_str(), ms);
      line += buf;
      if (&m != &ctx->tl[0]) (void)hipEventDestroy(m.second);
    }
    (void)hipEventDestroy(ctx->tl[0].second);
    ctx->tl.clear();
    fprintf(stderr, "%s\n", line.c_str());
  }
  if (ctx->cur->host_tail) {
    const hostf::HostField Fq = hostf::fq_field(pk->curve);
    const uint32_t* hp = ctx->cur->host_proof;
    hostf::groth16_points_into_affine(Fq, hp, hp + 4 * v1->fN, hp + 4 * v1->fN + 4 * v2->fN,
                                      reinterpret_cast<uint32_t*>(proof_out), inf_out);
  } else {
    memcpy(proof_out, ctx->cur->host_proof, proof_words * 4);
    for (int i = 0; i < 3; i++) inf_out[i] = (uint8_t)ctx->cur->host_proof[256 + i];
  }
  ctx->cur->busy = false;
}

void groth16_assemble(zkp_ctx* ctx, int curve, const uint64_t* sums, const uint64_t* r, const uint64_t* s,
                      uint64_t* proof_out, uint8_t* inf_out) {
  const MsmVtbl* v1 = msm_vtbl(curve, 1);
  const MsmVtbl* v2 = msm_vtbl(curve, 2);
  hipStream_t st = ctx->cur->stream;
  const size_t slot = v2->xyzz_bytes;
  const size_t j1 = 3 * (size_t)v1->fN, j2 = 3 * (size_t)v2->fN;          // words
  const size_t in_words = 4 * j1 + j2;
  uint32_t* buf = ctx->msm_misc.as<uint32_t>(in_words + 16 + 6 * slot / 4 + 256 + 16);
  uint32_t* jac = buf;
  uint32_t* rs = jac + in_words;
  char* res = reinterpret_cast<char*>(rs + 16);
  uint32_t* proof_dev = reinterpret_cast<uint32_t*>(res + 6 * slot);
  uint32_t* flags_dev = proof_dev + 240;
  ZKP_HIP(hipMemcpyAsync(jac, sums, in_words * 4, hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemcpyAsync(rs, r, 32, hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemcpyAsync(rs + 8, s, 32, hipMemcpyHostToDevice, st));
  const size_t off[5] = {0, j1, 2 * j1, 2 * j1 + j2, 3 * j1 + j2};
  for (int i = 0; i < 5; i++) (i == 2 ? v2 : v1)->from_jacobian(st, jac + off[i], res + i * slot);
  v1->assemble_g1_part1(st, res, slot, rs, proof_dev, flags_dev);
  v1->assemble_g1_part2(st, res, slot, proof_dev, flags_dev, 2 * v1->fN + 2 * v2->fN);
  v2->assemble_g2(st, res, slot, proof_dev, flags_dev, 2 * v1->fN);
  ZKP_HIP(hipGetLastError());
  const size_t proof_words = 4 * (size_t)v1->fN + 2 * (size_t)v2->fN;
  uint32_t flags_host[4] = {0, 0, 0, 0};
  ZKP_HIP(hipMemcpyAsync(proof_out, proof_dev, proof_words * 4, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipMemcpyAsync(flags_host, flags_dev, 12, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < 3; i++) inf_out[i] = (uint8_t)flags_host[i];
}

size_t groth16_partials_bytes(int curve) { return 5 * msm_vtbl(curve, 2)->xyzz_bytes; }

// Base-sharded step, part 1 (this rank): witness map (replicated) + the five partial MSMs over this rank's slices.
// Everything stays in HBM: out_dev receives groth16_partials_bytes() bytes, complete when the call returns.
void groth16_prove_partials(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_dev, const uint64_t* r, const uint64_t* s,
                            void* out_dev) {
  ZKP_REQUIRE(pk->shard_world > 0, ZKP_ERR_BAD_ARG);
  ctx->cur = &ctx->lanes[0];
  ctx->cur_idx = 0;
  if (pk->curve == ZKP_BN254) prove_enqueue_part<Bn254Fr>(ctx, pk, z_dev, true, r, s, 3, (char*)out_dev, true);
  else prove_enqueue_part<Bls381Fr>(ctx, pk, z_dev, true, r, s, 3, (char*)out_dev, true);
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
}

// Base-sharded step, part 2 (every rank, after the all-gather): fold the `world` rank buffers slot by slot and assemble
// the proof (prover.rs:192-210).  gathered_dev: world x groth16_partials_bytes() bytes of device memory.
void groth16_fold_assemble(zkp_ctx* ctx, int curve, const void* gathered_dev, int world, const uint64_t* r,
                           const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out) {
  ZKP_REQUIRE(world >= 1, ZKP_ERR_BAD_ARG);
  const MsmVtbl* v1 = msm_vtbl(curve, 1);
  const MsmVtbl* v2 = msm_vtbl(curve, 2);
  hipStream_t st = ctx->cur->stream;
  const size_t slot = v2->xyzz_bytes;
  uint32_t* buf = ctx->msm_misc.as<uint32_t>(16 + 6 * slot / 4 + 256 + 16);
  uint32_t* rs = buf;
  char* res = reinterpret_cast<char*>(rs + 16);
  uint32_t* proof_dev = reinterpret_cast<uint32_t*>(res + 6 * slot);
  uint32_t* flags_dev = proof_dev + 240;
  ZKP_HIP(hipMemcpyAsync(rs, r, 32, hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemcpyAsync(rs + 8, s, 32, hipMemcpyHostToDevice, st));
  v1->fold_slots(st, (const char*)gathered_dev, 5 * slot, world, slot, 0x1b, res);     // A, B1, H, L
  v2->fold_slots(st, (const char*)gathered_dev, 5 * slot, world, slot, 0x04, res);     // B2
  v1->assemble_g1_part1(st, res, slot, rs, proof_dev, flags_dev);
  v1->assemble_g1_part2(st, res, slot, proof_dev, flags_dev, 2 * v1->fN + 2 * v2->fN);
  v2->assemble_g2(st, res, slot, proof_dev, flags_dev, 2 * v1->fN);
  ZKP_HIP(hipGetLastError());
  const size_t proof_words = 4 * (size_t)v1->fN + 2 * (size_t)v2->fN;
  uint32_t flags_host[4] = {0, 0, 0, 0};
  ZKP_HIP(hipMemcpyAsync(proof_out, proof_dev, proof_words * 4, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipMemcpyAsync(flags_host, flags_dev, 12, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < 3; i++) inf_out[i] = (uint8_t)flags_host[i];
}

void groth16_prove(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, bool z_on_device, const uint64_t* r,
                   const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out) {
  ZKP_REQUIRE(pk->shard_world == 0, ZKP_ERR_BAD_ARG);     // a sharded key yields partial sums only
  ctx->cur = &ctx->lanes[0];
  ctx->cur_idx = 0;
  static const bool timeline = getenv("ZKP_TIMELINE") && atoi(getenv("ZKP_TIMELINE")) != 0;
  ctx->tl_on = timeline && !ctx->profiling;
  if (pk->curve == ZKP_BN254) prove_enqueue<Bn254Fr>(ctx, pk, z, z_on_device, r, s, true);
  else prove_enqueue<Bls381Fr>(ctx, pk, z, z_on_device, r, s, true);
  ctx->tl_on = false;
  prove_finish(ctx, pk, proof_out, inf_out);
}

// n proofs, software-pipelined over the two lanes: proof i+1 is enqueued before proof i is awaited, so the
// latency-bound tails (bucket reduction, assembly) of one proof overlap the throughput-bound kernels of the next.
void groth16_prove_batch(zkp_ctx* ctx, zkp_groth16_pk* pk, size_t n, const uint64_t* const* z_dev, const uint64_t* r,
                         const uint64_t* s, uint64_t* proofs_out, uint8_t* inf_out, bool z_on_device) {
  const MsmVtbl* v1 = msm_vtbl(pk->curve, 1);
  const MsmVtbl* v2 = msm_vtbl(pk->curve, 2);
  const size_t pw64 = (4 * (size_t)v1->fN + 2 * (size_t)v2->fN) / 2;
  const bool prof = ctx->profiling;
  ZKP_REQUIRE(pk->shard_world == 0, ZKP_ERR_BAD_ARG);
  size_t pending[zkp_ctx::N_LANES] = {};
  // proofs in flight: 8 lanes x 4 streams = 2 streams per hardware queue (measured optimum at 2^20: 4 lanes 102, 6 lanes 103,
  // 8 lanes 108-110, 12 lanes 104 proofs/s); 4 above 2^22 where a lane's scratch is tens of GB.  ZKP_LANES overrides.
  const int lanes_default = pk->log_n <= 22 ? 8 : 4;          // zkp_ctx_config.lanes / ZKP_LANES override it per context
  const int nl = std::max(1, std::min(ctx->cfg.lanes > 0 ? ctx->cfg.lanes : lanes_default, (int)zkp_ctx::N_LANES));
  auto select = [&](int l) {
    ctx->cur = &ctx->lanes[l];
    ctx->cur_idx = l;
  };
  struct BatchMode {                             // several proofs in flight: throughput tuning (msm.hip: segmented-sum chunk)
    zkp_ctx* c;
    ~BatchMode() { c->batch_mode = false; }
  } batch_mode{ctx};
  ctx->batch_mode = n > 1 && !prof;
  try {
    // First pipelined batch of this key: a lane's scratch (bucket arrays, sort buffers, NTT scratch: GiBs at 2^22) is allocated
    // when a proof first runs on it, and hipMalloc synchronises the device.  A batch shorter than the lane count would leave
    // that to a later call's steady state (measured: a 12-proof batch after a 4-proof one ran at 11 instead of 19 proofs/s at
    // 2^22 BLS12-381), so the lanes this key has never used run the first proof once, results discarded.
    if (ctx->batch_mode) {
      std::vector<uint64_t> scratch_proof(pw64 + 8);
      uint8_t scratch_inf[4];
      bool any = false;
      for (int l = 0; l < nl; l++) {
        if (pk->lane[l].warmed) continue;
        select(l);
        if (pk->curve == ZKP_BN254) prove_enqueue<Bn254Fr>(ctx, pk, z_dev[0], z_on_device, r, s);
        else prove_enqueue<Bls381Fr>(ctx, pk, z_dev[0], z_on_device, r, s);
        pk->lane[l].warmed = true;
        any = true;
      }
      if (any)
        for (int l = 0; l < nl; l++) {
          select(l);
          if (ctx->cur->busy) prove_finish(ctx, pk, scratch_proof.data(), scratch_inf);
        }
    }
    for (size_t i = 0; i < n; i++) {
      const int l = prof ? 0 : (int)(i % nl);
      select(l);
      if (ctx->cur->busy) prove_finish(ctx, pk, proofs_out + pending[l] * pw64, inf_out + pending[l] * 3);
      if (pk->curve == ZKP_BN254) prove_enqueue<Bn254Fr>(ctx, pk, z_dev[i], z_on_device, r + 4 * i, s + 4 * i);
      else prove_enqueue<Bls381Fr>(ctx, pk, z_dev[i], z_on_device, r + 4 * i, s + 4 * i);
      pending[l] = i;
    }
    for (int l = 0; l < zkp_ctx::N_LANES; l++) {
      select(l);
      if (ctx->cur->busy) prove_finish(ctx, pk, proofs_out + pending[l] * pw64, inf_out + pending[l] * 3);
    }
  } catch (...) {
    (void)hipDeviceSynchronize();
    for (auto& L : ctx->lanes) L.busy = false;
    select(0);
    throw;
  }
  select(0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-process multi-GPU (BASELINE configs[4]): the caller of `create_proof` (prover.rs:124) is ONE
// process, so the library owns one context per device (zkp_ctx_create_multi) and the exchange step itself.
//   shard      every (extended) query is split by index over the devices (rank k keeps 1/n of the window tables); one proof
//              uses all devices: partial MSMs per device -> partial sums gathered on device 0 over xGMI (peer copies of
//              n x 1.25 KiB; optionally an RCCL all-gather) -> fold + assembly on device 0.  With >= 3 devices the witness map
//              is task-split: the three independent chains of r1cs_to_qap.rs:144-162 (A z / B z / C z -> ifft -> coset_fft) run
//              on devices 0 / 1 / 2, the b and c coset evaluations are gathered on device 0 (pointwise step + coset_ifft), and
//              every device receives its slice of h for its share of the H MSM.
//   replicate  full key on every device, independent proofs dealt round-robin, one host thread per device driving that
//              device's pipelined lanes (zkp_groth16_prove_batch semantics).
// Everything is enqueued from the calling thread with streams and cross-device events; device ids may repeat (several ranks
// on ONE GPU: the tests on a one-GPU box), peer copies then degenerate to device-to-device copies.
}  // namespace zkp
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>

#include <thread>
struct zkp_groth16_pk_multi {
  int mode = 0, curve = 0;
  std::vector<zkp_groth16_pk*> pk;          // one per device of the root context, rank order
  std::vector<hipEvent_t> ev_chain, ev_done, ev_h_read;
  hipEvent_t ev_h = nullptr;
  void* gathered = nullptr;                 // device 0: n x partials bytes
  std::vector<void*> gathered_all;          // RCCL exchange: a receive buffer on every device
  std::vector<void*> partial;               // per device: 5 XYZZ slots
  // run-time choices of the sharded prover (zkp_groth16_multi_info): which exchange carried the last proof, and whether the
  // three chains of the witness map are split over devices 0..2 — measured on this key's own first proofs, not assumed
  int exchange = 0;                         // 0 = peer copies, 1 = RCCL all-gather, 2 = peer copies after the RCCL watchdog gave up
  bool exchange_fallback = false;
  int rccl_ranks = 0;
  int split_choice = -1;                    // -1 undecided, 0 replicated witness map, 1 three-way split
  int calls = 0;
  double ms_variant[2] = {0.0, 0.0};        // wall time of a proof with the replicated / split witness map (calls 3 and 4)
};
namespace zkp {

namespace {
// RCCL all-gather of the partial sums, resolved at run time (no link-time dependency: the host may already have loaded its
// own librccl).  Needs distinct devices.  Falls back to peer copies when unavailable.
//
// Watchdog (round 6).  The first multi-rank ncclCommInitAll and the first collective are the two places where a mis-configured
// node HANGS instead of failing (no IPC handles, a dead link, a peer that never joins), and a hung create_proof is worse than a
// slow one.  So the whole bring-up — dlopen, ncclCommInitAll, and a PROBE all-gather of 64 bytes per rank on throw-away streams
// and buffers — runs on a helper thread; the calling thread waits for it with a deadline (zkp_ctx_config.multi_exchange_timeout_ms
// / ZKP_MULTI_EXCHANGE_TIMEOUT_MS, default 30 s per stage).  The helper polls the probe's streams itself and, when they do not
// drain in time, aborts the communicators (ncclCommAbort makes RCCL's kernels exit).  Either way — probe failed, or the helper
// never came back — RCCL is marked unusable for the life of the process, the reason is printed once on stderr, the key's
// zkp_groth16_multi_info says 2 ("peer copies after the RCCL watchdog"), and the partial sums travel by hipMemcpyPeerAsync.
// The proof's streams only ever see an all-gather on communicators whose probe completed.
// ZKP_DEBUG_RCCL_HANG=init | probe simulates the two hangs on any box (tests/test_gpu_multi.py).
struct RcclApi {
  struct Fns {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  };
  // what the helper thread and the caller share; the caller may walk away from it (shared_ptr keeps it alive for a helper
  // that returns late)
  struct Attempt {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, ok = false, abandoned = false;
    Fns f;
    std::vector<void*> comms;
    std::string why;
  };
  Fns f;
  std::mutex mu;                    // the communicators are process-wide: one bring-up, and one group of collective calls, at a time
  // one set of communicators per device set, kept for the life of the process (a second root over other devices must not tear
  // down the communicators a first root is about to use)
  std::map<std::vector<int>, std::vector<void*>> comms_of;
  bool unusable = false, watchdog_fired = false;
  std::string why;

  static void bring_up(std::shared_ptr<Attempt> at, std::vector<int> ids, int timeout_ms) {
    const char* hang = getenv("ZKP_DEBUG_RCCL_HANG");
    Fns f;
    std::vector<void*> comms(ids.size(), nullptr);
    std::string why;
    bool ok = false;
    do {
      for (const char* name : {"librccl.so", "librccl.so.1"})
        if ((f.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
      if (!f.lib) { why = "librccl.so not loadable"; break; }
      f.CommInitAll = reinterpret_cast<decltype(f.CommInitAll)>(dlsym(f.lib, "ncclCommInitAll"));
      f.CommDestroy = reinterpret_cast<decltype(f.CommDestroy)>(dlsym(f.lib, "ncclCommDestroy"));
      f.CommAbort = reinterpret_cast<decltype(f.CommAbort)>(dlsym(f.lib, "ncclCommAbort"));
      f.GroupStart = reinterpret_cast<decltype(f.GroupStart)>(dlsym(f.lib, "ncclGroupStart"));
      f.GroupEnd = reinterpret_cast<decltype(f.GroupEnd)>(dlsym(f.lib, "ncclGroupEnd"));
      f.AllGather = reinterpret_cast<decltype(f.AllGather)>(dlsym(f.lib, "ncclAllGather"));
      if (!f.CommInitAll || !f.CommDestroy || !f.GroupStart || !f.GroupEnd || !f.AllGather) { why = "librccl.so lacks a symbol"; break; }
      if (hang && !strcmp(hang, "init")) std::this_thread::sleep_for(std::chrono::hours(24));      // simulated: ncclCommInitAll never returns
      if (f.CommInitAll(comms.data(), (int)ids.size(), ids.data()) != 0) { why = "ncclCommInitAll failed"; comms.clear(); break; }
      // probe: one all-gather of 64 bytes per rank on streams and buffers nothing else uses
      const int n = (int)ids.size();
      std::vector<hipStream_t> st(n, nullptr);
      std::vector<void*> snd(n, nullptr), rcv(n, nullptr);
      bool issued = true;
      for (int k = 0; k < n && issued; k++)
        issued = hipSetDevice(ids[k]) == hipSuccess && hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) == hipSuccess &&
                 hipMalloc(&snd[k], 64) == hipSuccess && hipMalloc(&rcv[k], 64 * (size_t)n) == hipSuccess &&
                 hipMemsetAsync(snd[k], k + 1, 64, st[k]) == hipSuccess;
      if (issued) {
        issued = f.GroupStart() == 0;
        for (int k = 0; k < n && issued; k++)
          issued = hipSetDevice(ids[k]) == hipSuccess && f.AllGather(snd[k], rcv[k], 64, /*ncclUint8*/ 1, comms[k], st[k]) == 0;
        issued = f.GroupEnd() == 0 && issued;
      }
      bool drained = false, right = true;
      if (issued) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
        while (!drained && std::chrono::steady_clock::now() < deadline) {
          drained = !(hang && !strcmp(hang, "probe"));                                     // simulated: the collective never completes
          for (int k = 0; k < n && drained; k++) drained = hipStreamQuery(st[k]) == hipSuccess;
          if (!drained) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        if (drained) {                                                                   // every rank holds byte k + 1 in slot k
          std::vector<uint8_t> got(64 * (size_t)n);
          for (int k = 0; k < n && right; k++) {
            right = hipSetDevice(ids[k]) == hipSuccess && hipMemcpy(got.data(), rcv[k], got.size(), hipMemcpyDeviceToHost) == hipSuccess;
            for (int j = 0; j < n && right; j++) right = got[64 * (size_t)j] == (uint8_t)(j + 1) && got[64 * (size_t)j + 63] == (uint8_t)(j + 1);
          }
        }
      }
      if (!issued || !drained || !right) {
        why = !issued ? "the probe all-gather could not be enqueued" : !drained ? "the probe all-gather did not complete within " + std::to_string(timeout_ms) + " ms"
                      : "the probe all-gather returned wrong bytes";
        for (void* c : comms)
          if (c) (void)(f.CommAbort ? f.CommAbort(c) : f.CommDestroy(c));
        comms.clear();
        if (!drained) break;                                                             // streams may still be wedged: leak the probe's resources
      }
      for (int k = 0; k < n; k++) {
        (void)hipSetDevice(ids[k]);
        if (st[k]) (void)hipStreamDestroy(st[k]);
        if (snd[k]) (void)hipFree(snd[k]);
        if (rcv[k]) (void)hipFree(rcv[k]);
      }
      ok = issued && drained && right;
    } while (false);
    std::lock_guard<std::mutex> lk(at->m);
    if (at->abandoned) {                                  // the caller gave up on this attempt: nobody will use these communicators
      for (void* c : comms)
        if (c && f.CommAbort) (void)f.CommAbort(c);
      return;
    }
    at->f = f;
    at->comms = ok ? comms : std::vector<void*>();
    at->ok = ok;
    at->why = why;
    at->done = true;
    at->cv.notify_all();
  }

  // -> the communicators of this device set (rank order), or nullptr
  const std::vector<void*>* ready(const std::vector<zkp_ctx*>& devs, int timeout_ms) {
    if (unusable) return nullptr;
    std::vector<int> ids;
    for (zkp_ctx* d : devs) ids.push_back(d->device);
    if (auto it = comms_of.find(ids); it != comms_of.end()) return &it->second;
    for (size_t i = 0; i < ids.size(); i++)
      for (size_t j = i + 1; j < ids.size(); j++)
        if (ids[i] == ids[j]) { why = "duplicate device ids"; return nullptr; }           // RCCL refuses duplicate devices (not sticky)
    timeout_ms = std::max(1, timeout_ms);
    auto at = std::make_shared<Attempt>();
    std::thread(bring_up, at, ids, timeout_ms).detach();
    std::unique_lock<std::mutex> lk(at->m);
    // two stages on the helper (communicator setup, probe), one deadline each, and a margin for the teardown
    const bool back = at->cv.wait_for(lk, std::chrono::milliseconds(2 * (long long)timeout_ms + 2000), [&] { return at->done; });
    if (!back) {
      at->abandoned = true;
      unusable = watchdog_fired = true;
      why = "RCCL bring-up (ncclCommInitAll + probe all-gather) did not return within " + std::to_string(2 * (long long)timeout_ms + 2000) + " ms";
      return nullptr;
    }
    f = at->f;
    if (!at->ok) {
      why = at->why;
      // a missing library is an ordinary "unavailable"; a failed init or probe is what the watchdog exists for
      unusable = true;
      watchdog_fired = f.lib != nullptr && at->why != "librccl.so lacks a symbol";
      return nullptr;
    }
    return &(comms_of[ids] = at->comms);
  }
};
RcclApi& rccl() {
  static RcclApi* api = new RcclApi();      // never destructed: communicators must not be torn down after the HIP runtime
  return *api;
}

void copy_between(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st) {
  if (!bytes) return;
  if (dst_dev == src_dev) ZKP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
  else ZKP_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st));
}

// one of the three chains of the witness map (k = 0 / 1 / 2: a / b / c), on ctx's current stream; result in region k of abc
template <class P>
void witness_chain(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint32_t* z_dev, int k) {
  const uint32_t N = (uint32_t)pk->N;
  uint32_t* abc = pk->lane[ctx->cur_idx].abc.as<uint32_t>(3 * pk->N * 8);
  uint32_t* buf = abc + (size_t)k * pk->N * 8;
  hipLaunchKernelGGL(csr_eval_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, ctx->cur->stream, pk->m[k].row_ptr, pk->m[k].col,
                     pk->m[k].coeff, z_dev, pk->num_constraints, N, pk->num_inputs, k == 0 ? 1 : 0, buf);
  if (!ntt_ifft_coset_fft(ctx, pk->curve, buf, pk->log_n)) {
    ntt_run(ctx, pk->curve, buf, pk->log_n, ZKP_NTT_IFFT);
    ntt_run(ctx, pk->curve, buf, pk->log_n, ZKP_NTT_COSET_FFT);
  }
  ZKP_HIP(hipGetLastError());
}
// second half of the witness map (pointwise step + coset_ifft) over the three regions of abc; returns h
template <class P>
uint32_t* witness_tail(zkp_ctx* ctx, zkp_groth16_pk* pk) {
  const uint32_t N = (uint32_t)pk->N;
  uint32_t* a = pk->lane[ctx->cur_idx].abc.as<uint32_t>(3 * pk->N * 8);
  uint32_t* b = a + pk->N * 8;
  uint32_t* c = b + pk->N * 8;
  if (pk->h_lagrange) {                                      // evaluation-form key: the H MSM takes the pointwise values
    if (pk->c_folded)
      hipLaunchKernelGGL(qap_pointwise_ab_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, ctx->cur->stream, a, b,
                         pk->consts.as<uint32_t>(64), N);
    else
    hipLaunchKernelGGL(qap_pointwise_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, ctx->cur->stream, a, b, c,
                       pk->consts.as<uint32_t>(64), N);
    ZKP_HIP(hipGetLastError());
    return a;
  }
  if (uint32_t* hq = ntt_qap_coset_ifft(ctx, pk->curve, a, b, c, pk->consts.as<uint32_t>(64), pk->log_n)) return hq;
  hipLaunchKernelGGL(qap_pointwise_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, ctx->cur->stream, a, b, c,
                     pk->consts.as<uint32_t>(64), N);
  ntt_run(ctx, pk->curve, a, pk->log_n, ZKP_NTT_COSET_IFFT);
  ZKP_HIP(hipGetLastError());
  return a;
}

template <class FrP>
void prove_multi_t(zkp_ctx* root, zkp_groth16_pk_multi* M, const uint64_t* const* z, bool z_on_device, const uint64_t* r,
                   const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out) {
  const int n = (int)root->devs.size();
  const MsmVtbl* v2 = msm_vtbl(M->curve, 2);
  const size_t slot = v2->xyzz_bytes, pb = 5 * slot;
  // Witness map: replicated on every device, or its a / b / c chains on devices 0 / 1 / 2 with two N x 32-byte vectors shipped to
  // device 0 and the slices of h shipped back (n >= 3).  Which one is faster depends on the link (one xGMI hop vs. a PCIe switch)
  // and on the domain, so the key MEASURES it: proofs 1-2 warm both variants up, proofs 3-4 time them (wall clock of the whole
  // call, same witness), from proof 5 on the faster one runs.  The proof bytes do not depend on the choice.
  // ZKP_MULTI_WM_SPLIT=0 / 1 forces a variant.
  const int split_env = root->cfg.multi_wm_split;               // zkp_ctx_config.multi_witness_split / ZKP_MULTI_WM_SPLIT
  bool split = false;
  const int call = M->calls++;
  if (n >= 3) {
    if (split_env >= 0) M->split_choice = split_env ? 1 : 0;
    if (M->split_choice >= 0) split = M->split_choice == 1;
    else split = (call & 1) == 1;                                   // calls 0, 2: replicated; 1, 3: split
  } else {
    M->split_choice = 0;
  }
  const auto t_call = std::chrono::steady_clock::now();
  // Exchange: RCCL all-gather (the exchange BASELINE.json names) whenever the devices are distinct and librccl loads; peer copies
  // otherwise, said once on stderr.  ZKP_MULTI_EXCHANGE=peer / rccl forces one.
  // ZKP_MULTI_EXCHANGE=rccl also takes the RCCL branch with ONE rank (a one-rank ncclCommInitAll + ncclAllGather is legal): the
  // hand-declared prototypes below get executed on a one-GPU box (tests/test_gpu_multi.py) before any 8-GPU node sees them.
  const int want = root->cfg.multi_exchange;                     // zkp_ctx_config.multi_exchange / ZKP_MULTI_EXCHANGE, per context
  bool use_rccl = false;
  const std::vector<void*>* rccl_comms = nullptr;
  if ((n > 1 && want != ZKP_EXCHANGE_PEER) || (n == 1 && want == ZKP_EXCHANGE_RCCL)) {
    std::lock_guard<std::mutex> lk(rccl().mu);
    rccl_comms = rccl().ready(root->devs, root->cfg.multi_exchange_timeout_ms);
    use_rccl = rccl_comms != nullptr;
    static bool told = false;
    if (!use_rccl && !told) {
      told = true;
      fprintf(stderr, "[zkp_accel] sharded prover: RCCL all-gather unavailable (%s)%s — the partial sums travel by peer copies\n",
              rccl().why.c_str(), rccl().watchdog_fired ? " [watchdog: RCCL stays off for this process]" : "");
    }
    if (!use_rccl && rccl().watchdog_fired) M->exchange_fallback = true;
  }
  M->exchange = use_rccl ? 1 : (M->exchange_fallback ? 2 : 0);
  M->rccl_ranks = use_rccl ? n : 0;
  std::vector<uint32_t*> S(n), h(n, nullptr);
  std::vector<char*> res(n);
  // phase 1 — every device: inputs, the four z-MSMs over its slices on the MSM streams, its part of the witness map
  for (int k = 0; k < n; k++) {
    zkp_ctx* ctx = root->devs[k];
    zkp_groth16_pk* pk = M->pk[k];
    ZKP_HIP(hipSetDevice(ctx->device));
    ctx->cur = &ctx->lanes[0];
    ctx->cur_idx = 0;
    hipStream_t st = ctx->cur->stream;
    zkp_groth16_pk::PerLane& PL = pk->lane[0];
    S[k] = PL.S.as<uint32_t>((pk->nz + 4) * 8);
    uint32_t* rs = PL.proof.as<uint32_t>(1024);
    res[k] = reinterpret_cast<char*>(PL.results.get(6 * slot));
    const uint64_t* zk = z_on_device ? z[k] : z[0];
    ZKP_HIP(hipMemcpyAsync(S[k], zk, pk->nz * 32, z_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    ZKP_HIP(hipMemcpyAsync(rs, r, 32, hipMemcpyHostToDevice, st));
    ZKP_HIP(hipMemcpyAsync(rs + 8, s, 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(scalar_tail_kernel<FrP>, dim3(1), dim3(64), 0, st, S[k] + pk->nz * 8, rs);
    ZKP_HIP(hipEventRecord(ctx->cur->ev_fork, st));
    for (int w = 1; w < zkp_ctx::N_WS; w++) ZKP_HIP(hipStreamWaitEvent(ctx->cur->ws[w].stream, ctx->cur->ev_fork, 0));
    if (!split) h[k] = witness_map_dev<FrP>(ctx, pk, S[k], !pk->h_lagrange);
    else if (k < (pk->c_folded ? 2 : 3)) {                           // C folded into the L query: no c chain
      witness_chain<FrP>(ctx, pk, S[k], k);
      ZKP_HIP(hipEventRecord(M->ev_chain[k], st));
    }
    const uint64_t* Sd = reinterpret_cast<const uint64_t*>(S[k]);
    auto run = [&](int idx, uint64_t handle, int w, int sort_src, int l1_src) {
      msm_run(ctx, handle, 0, Sd + 4 * pk->q_lo[idx], pk->q_n[idx], true, nullptr, res[k] + idx * slot, nullptr, nullptr, w,
              sort_src, nullptr, l1_src);
    };
    const int l1 = pk->share_l1 ? 1 : -1;
    const bool b_l1 = pk->share_l1 && pk->b_in_l1;
    if (!b_l1) run(2, pk->hB2, 2, -1, -1);
    run(0, pk->hA, 1, -1, -1);
    if (b_l1) run(2, pk->hB2, 2, -1, l1);
    run(1, pk->hB1, 3, pk->share_b_sort ? 2 : -1, -1);
    run(4, pk->hL, 1, pk->share_al_sort ? 1 : -1, l1);
  }
  // phase 2 (task-split witness map) — device 0 gathers b and c, finishes h, every device fetches its slice of h
  if (split) {
    zkp_ctx* c0 = root->devs[0];
    zkp_groth16_pk* p0 = M->pk[0];
    ZKP_HIP(hipSetDevice(c0->device));
    hipStream_t st0 = c0->cur->stream;
    uint32_t* abc0 = p0->lane[0].abc.as<uint32_t>(3 * p0->N * 8);
    const int nch = p0->c_folded ? 2 : 3;
    for (int k = 1; k < nch; k++) {
      zkp_groth16_pk* pk = M->pk[k];
      const uint32_t* src = pk->lane[0].abc.as<uint32_t>(3 * pk->N * 8) + (size_t)k * pk->N * 8;
      ZKP_HIP(hipStreamWaitEvent(st0, M->ev_chain[k], 0));
      copy_between(abc0 + (size_t)k * p0->N * 8, c0->device, src, root->devs[k]->device, p0->N * 32, st0);
    }
    h[0] = witness_tail<FrP>(c0, p0);
    ZKP_HIP(hipEventRecord(M->ev_h, st0));
    for (int k = 1; k < n; k++) {
      zkp_ctx* ctx = root->devs[k];
      zkp_groth16_pk* pk = M->pk[k];
      ZKP_HIP(hipSetDevice(ctx->device));
      hipStream_t st = ctx->cur->stream;
      h[k] = pk->lane[0].abc.as<uint32_t>(3 * pk->N * 8);        // region a of this device (its own chain, if any, is consumed)
      ZKP_HIP(hipStreamWaitEvent(st, M->ev_h, 0));
      if (k < nch) ZKP_HIP(hipStreamWaitEvent(st, M->ev_chain[k], 0));
      copy_between(h[k] + pk->q_lo[3] * 8, ctx->device, h[0] + pk->q_lo[3] * 8, c0->device, pk->q_n[3] * 32, st);
      ZKP_HIP(hipEventRecord(M->ev_h_read[k], st));
    }
  }
  // phase 3 — every device: H MSM over its slice of h, join, partial sums
  for (int k = 0; k < n; k++) {
    zkp_ctx* ctx = root->devs[k];
    zkp_groth16_pk* pk = M->pk[k];
    ZKP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->cur->stream;
    msm_run(ctx, pk->hH, 0, reinterpret_cast<const uint64_t*>(h[k]) + 4 * pk->q_lo[3], pk->q_n[3], true, nullptr,
            res[k] + 3 * slot, nullptr, nullptr, 0);
    for (int w = 1; w < zkp_ctx::N_WS; w++) {
      ZKP_HIP(hipEventRecord(ctx->cur->ws[w].done, ctx->cur->ws[w].stream));
      ZKP_HIP(hipStreamWaitEvent(st, ctx->cur->ws[w].done, 0));
    }
    ZKP_HIP(hipMemcpyAsync(M->partial[k], res[k], pb, hipMemcpyDeviceToDevice, st));
    ZKP_HIP(hipEventRecord(M->ev_done[k], st));
  }
  // phase 4 — exchange (n x 1.25 KiB for BN254: latency-bound on any topology) + fold + assembly on device 0
  zkp_ctx* c0 = root->devs[0];
  const void* gathered = M->gathered;
  if (use_rccl) {
    RcclApi& R = rccl();
    // two roots over the same devices share these communicators (two prover threads of one process): their groups are enqueued one
    // after the other, in the same order on every rank
    std::lock_guard<std::mutex> lk(R.mu);
    ZKP_REQUIRE(R.f.GroupStart() == 0, ZKP_ERR_DEVICE);
    for (int k = 0; k < n; k++) {
      ZKP_HIP(hipSetDevice(root->devs[k]->device));
      ZKP_REQUIRE(R.f.AllGather(M->partial[k], M->gathered_all[k], pb, /*ncclUint8*/ 1, (*rccl_comms)[k], root->devs[k]->cur->stream) == 0,
                  ZKP_ERR_DEVICE);
    }
    ZKP_REQUIRE(R.f.GroupEnd() == 0, ZKP_ERR_DEVICE);
    gathered = M->gathered_all[0];
    ZKP_HIP(hipSetDevice(c0->device));
  } else {
    ZKP_HIP(hipSetDevice(c0->device));
    hipStream_t st0 = c0->cur->stream;
    for (int k = 0; k < n; k++) {
      if (k) ZKP_HIP(hipStreamWaitEvent(st0, M->ev_done[k], 0));
      copy_between((char*)M->gathered + (size_t)k * pb, c0->device, M->partial[k], root->devs[k]->device, pb, st0);
    }
  }
  if (split)                                     // h on device 0 must outlive the slice reads of the other devices
    for (int k = 1; k < n; k++) ZKP_HIP(hipStreamWaitEvent(c0->cur->stream, M->ev_h_read[k], 0));
  groth16_fold_assemble(c0, M->curve, gathered, n, r, s, proof_out, inf_out);        // synchronises device 0's stream
  for (int k = 1; k < n; k++) {                  // the next call reuses every device's buffers
    ZKP_HIP(hipSetDevice(root->devs[k]->device));
    ZKP_HIP(hipStreamSynchronize(root->devs[k]->cur->stream));
  }
  ZKP_HIP(hipSetDevice(c0->device));
  if (n >= 3 && M->split_choice < 0 && call >= 2) {
    M->ms_variant[split ? 1 : 0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    if (call >= 3) M->split_choice = M->ms_variant[1] < M->ms_variant[0] ? 1 : 0;
  }
}
}  // namespace

zkp_groth16_pk_multi* groth16_pk_upload_multi(zkp_ctx* root, const zkp_groth16_pk_desc* d, int mode) {
  ZKP_REQUIRE(!root->devs.empty() && (mode == 0 || mode == 1), ZKP_ERR_BAD_ARG);
  const int n = (int)root->devs.size();
  std::unique_ptr<zkp_groth16_pk_multi> M(new zkp_groth16_pk_multi());
  M->mode = mode;
  M->curve = d->curve;
  struct KeepLagrange {
    KeepLagrange() { lagrange_keep_cache = true; }
    ~KeepLagrange() {
      lagrange_keep_cache = false;
      lagrange_cache.clear();
      lagrange_cache_g.clear();
    }
  } keep_lagrange;
  try {
    for (int k = 0; k < n; k++) {
      zkp_ctx* ctx = root->devs[k];
      ZKP_HIP(hipSetDevice(ctx->device));
      M->pk.push_back(mode == 0 ? groth16_pk_upload(ctx, d, k, n) : groth16_pk_upload(ctx, d, 0, 0));
      hipEvent_t e;
      ZKP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      M->ev_chain.push_back(e);
      ZKP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      M->ev_done.push_back(e);
      ZKP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      M->ev_h_read.push_back(e);
      void* p = nullptr;
      if (hipMalloc(&p, groth16_partials_bytes(d->curve)) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
      M->partial.push_back(p);
      p = nullptr;
      if (hipMalloc(&p, (size_t)n * groth16_partials_bytes(d->curve)) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
      M->gathered_all.push_back(p);
    }
    ZKP_HIP(hipSetDevice(root->device));
    ZKP_HIP(hipEventCreateWithFlags(&M->ev_h, hipEventDisableTiming));
    M->gathered = M->gathered_all[0];
  } catch (...) {
    groth16_pk_multi_free(root, M.release());
    throw;
  }
  return M.release();
}

void groth16_pk_multi_free(zkp_ctx* root, zkp_groth16_pk_multi* M) {
  for (size_t k = 0; k < M->pk.size(); k++) {
    (void)hipSetDevice(root->devs[k]->device);
    if (M->pk[k]) groth16_pk_free(root->devs[k], M->pk[k]);
  }
  for (size_t k = 0; k < M->partial.size(); k++) {
    (void)hipSetDevice(root->devs[k]->device);
    if (M->partial[k]) (void)hipFree(M->partial[k]);
  }
  for (size_t k = 0; k < M->gathered_all.size(); k++) {
    (void)hipSetDevice(root->devs[k]->device);
    if (M->gathered_all[k]) (void)hipFree(M->gathered_all[k]);
  }
  for (auto& v : {M->ev_chain, M->ev_done, M->ev_h_read})
    for (hipEvent_t e : v)
      if (e) (void)hipEventDestroy(e);
  if (M->ev_h) (void)hipEventDestroy(M->ev_h);
  (void)hipSetDevice(root->device);
  delete M;
}

// info[0] = exchange of the last proof (0 peer copies, 1 RCCL all-gather), [1] = RCCL ranks, [2] = witness map (0 replicated, 1 split
// over devices 0..2, 2 = still measuring), [3] / [4] = microseconds of the timed proof with the replicated / split map, [5] = devices
void groth16_multi_info(zkp_ctx* root, zkp_groth16_pk_multi* M, uint64_t info[6]) {
  info[0] = (uint64_t)M->exchange;
  info[1] = (uint64_t)M->rccl_ranks;
  info[2] = M->split_choice < 0 ? 2 : (uint64_t)M->split_choice;
  info[3] = (uint64_t)(M->ms_variant[0] * 1e3);
  info[4] = (uint64_t)(M->ms_variant[1] * 1e3);
  info[5] = (uint64_t)root->devs.size();
}

void groth16_prove_multi(zkp_ctx* root, zkp_groth16_pk_multi* M, const uint64_t* const* z, bool z_on_device,
                         const uint64_t* r, const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out) {
  ZKP_REQUIRE(M->mode == 0 && M->pk.size() == root->devs.size() && !root->devs.empty(), ZKP_ERR_BAD_ARG);
  try {
    if (M->curve == ZKP_BN254) prove_multi_t<Bn254Fr>(root, M, z, z_on_device, r, s, proof_out, inf_out);
    else prove_multi_t<Bls381Fr>(root, M, z, z_on_device, r, s, proof_out, inf_out);
  } catch (...) {
    for (zkp_ctx* c : root->devs) {
      (void)hipSetDevice(c->device);
      (void)hipDeviceSynchronize();
    }
    (void)hipSetDevice(root->device);
    throw;
  }
}

// Throughput mode on every device of the root context: proof i runs on device i % n (its witness pointer, when
// z_on_device, lives there); one host thread per device drives that device's lanes exactly as zkp_groth16_prove_batch does.
void groth16_prove_batch_multi(zkp_ctx* root, zkp_groth16_pk_multi* M, size_t count, const uint64_t* const* z,
                               bool z_on_device, const uint64_t* r, const uint64_t* s, uint64_t* proofs_out,
                               uint8_t* inf_out) {
  ZKP_REQUIRE(M->mode == 1 && M->pk.size() == root->devs.size() && !root->devs.empty(), ZKP_ERR_BAD_ARG);
  const int n = (int)root->devs.size();
  const MsmVtbl* v1 = msm_vtbl(M->curve, 1);
  const MsmVtbl* v2 = msm_vtbl(M->curve, 2);
  const size_t pw64 = (4 * (size_t)v1->fN + 2 * (size_t)v2->fN) / 2;
  std::vector<int32_t> status(n, ZKP_OK);
  std::vector<std::thread> th;
  for (int k = 0; k < n; k++) {
    th.emplace_back([&, k] {
      try {
        zkp_ctx* ctx = root->devs[k];
        ZKP_HIP(hipSetDevice(ctx->device));
        std::vector<const uint64_t*> zk;
        std::vector<uint64_t> rk, sk;
        for (size_t i = (size_t)k; i < count; i += (size_t)n) {
          zk.push_back(z[i]);
          rk.insert(rk.end(), r + 4 * i, r + 4 * i + 4);
          sk.insert(sk.end(), s + 4 * i, s + 4 * i + 4);
        }
        if (zk.empty()) return;
        std::vector<uint64_t> po(zk.size() * pw64);
        std::vector<uint8_t> io(zk.size() * 3);
        groth16_prove_batch(ctx, M->pk[k], zk.size(), zk.data(), rk.data(), sk.data(), po.data(), io.data(), z_on_device);
        size_t j = 0;
        for (size_t i = (size_t)k; i < count; i += (size_t)n, j++) {
          memcpy(proofs_out + i * pw64, po.data() + j * pw64, pw64 * 8);
          memcpy(inf_out + i * 3, io.data() + j * 3, 3);
        }
      } catch (const StatusError& e) {
        status[k] = e.status;
      } catch (const HipError& e) {
        fprintf(stderr, "[zkp_accel] HIP error %s at line %d: %s (device rank %d)\n", hipGetErrorString(e.e), e.line, e.what, k);
        status[k] = e.e == hipErrorOutOfMemory ? ZKP_ERR_OOM : ZKP_ERR_DEVICE;
      } catch (...) {
        status[k] = ZKP_ERR_DEVICE;
      }
    });
  }
  for (auto& t : th) t.join();
  (void)hipSetDevice(root->device);
  for (int32_t st : status)
    if (st != ZKP_OK) throw StatusError{st};
}

}  // namespace zkp
