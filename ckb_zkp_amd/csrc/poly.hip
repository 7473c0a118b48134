// Fr polynomial / vector primitives for the Marlin prover's polynomial pipeline and KZG10 openings.
//
// Replaces (device side) the ark-poly / ark-ff helpers that /root/reference/marlin/src calls around its MSMs and NTTs:
//   * `p / (X - z)`  — witness polynomial of KZG10::open          marlin/src/pc/kzg10.rs:211-226
//   * `p.evaluate(z)` — Horner evaluations                          marlin/src/lib.rs:147-156, pc/kzg10.rs:142
//   * `fields::batch_inversion`                                     marlin/src/ahp/prover.rs:357-367, ahp/arithmetic.rs:32
//   * element-wise products / sums / axpy of evaluation vectors     marlin/src/ahp/prover.rs:248-252,298-305,399-411
// All vectors: Fr Montgomery, AoS 32 B/element, device memory.
//
// Synthetic division and evaluation are first-order linear recurrences (q_{i-1} = p_i + z q_i); a single lane needs
// ~1 us per Montgomery product, so they are evaluated as a three-phase blocked scan: per-chunk local Horner,
// a short scan over chunk heads with z^CHUNK, then a per-chunk replay — 3 products per coefficient, log depth.
#include <algorithm>

#include <vector>

#include "field_dev.hpp"
#include "internal.hpp"

namespace zkp {

constexpr int POLY_CHUNK = 32;        // coefficients per lane

template <class P>
__global__ __launch_bounds__(256) void vec_op_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                     const uint32_t* __restrict__ k, uint32_t* __restrict__ out,
                                                     size_t n, int op) {
  using F = Fp<P>;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F x = F::load(a + i * 8);
  F r;
  switch (op) {
    case 0: r = x * F::load(b + i * 8); break;                       // mul
    case 1: r = x + F::load(b + i * 8); break;                       // add
    case 2: r = x - F::load(b + i * 8); break;                       // sub
    case 3: r = x * F::load(k); break;                               // scale by constant
    case 5: r = x + F::load(k); break;                               // add constant
    default: r = x + F::load(b + i * 8) * F::load(k); break;         // axpy: a + k*b
  }
  r.store(out + i * 8);
}

// batch inversion, Montgomery's trick per lane over a strided chunk (zeros are left untouched, as ark does)
template <class P>
__global__ __launch_bounds__(256) void batch_inverse_kernel(uint32_t* __restrict__ v, size_t n, size_t lanes) {
  using F = Fp<P>;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  // elements t, t+lanes, t+2*lanes, ... (coalesced across lanes)
  F pref[POLY_CHUNK];
  F acc = F::one();
  int cnt = 0;
  for (size_t i = t; i < n && cnt < POLY_CHUNK; i += lanes, cnt++) {
    F x = F::load(v + i * 8);
    pref[cnt] = acc;
    if (!x.is_zero()) acc = acc * x;
  }
  F inv = acc.inv();
  for (int c = cnt - 1; c >= 0; c--) {
    size_t i = t + (size_t)c * lanes;
    F x = F::load(v + i * 8);
    if (x.is_zero()) continue;
    (inv * pref[c]).store(v + i * 8);
    inv = inv * x;
  }
}

// phase 1: chunk c covers coefficients [c*CH, min(n,(c+1)*CH)); head[c] = sum_j p_j z^(j - c*CH)
template <class P>
__device__ __forceinline__ void horner_chunk_body(const uint32_t* __restrict__ p, size_t n, const uint32_t* __restrict__ z,
                                                  uint32_t* __restrict__ head, size_t chunks) {
  using F = Fp<P>;
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= chunks) return;
  F zz = F::load(z);
  size_t lo = c * POLY_CHUNK, hi = lo + POLY_CHUNK < n ? lo + POLY_CHUNK : n;
  F acc = F::zero();
  for (size_t j = hi; j-- > lo;) acc = acc * zz + F::load(p + j * 8);
  acc.store(head + c * 8);
}
template <class P>
__global__ __launch_bounds__(256) void horner_chunk_kernel(const uint32_t* __restrict__ p, size_t n,
                                                           const uint32_t* __restrict__ z, uint32_t* __restrict__ head,
                                                           size_t chunks) {
  horner_chunk_body<P>(p, n, z, head, chunks);
}
// one evaluation of a batch (poly_evaluate_batch): blockIdx.y selects the job, so the 21 evaluations of a Marlin proof are three
// launches instead of 63 latency-bound ones
struct HornerJob {
  const uint32_t* p;
  size_t n, chunks, tiles;
  const uint32_t* z;
  uint32_t *head, *tile_sum, *ev;
};
template <class P>
__global__ __launch_bounds__(256) void horner_chunk_batch_kernel(const HornerJob* __restrict__ jobs) {
  const HornerJob j = jobs[blockIdx.y];
  horner_chunk_body<P>(j.p, j.n, j.z, j.head, j.chunks);
}

// phase 2: suffix scan over chunk heads, S[c] = head[c] + zc * S[c+1] with zc = z^CHUNK, in three steps:
//   (a) every block scans one tile of 256 heads locally (Hillis-Steele on affine maps x -> val + mul * x in LDS) and
//       emits the tile's own suffix sum;  (b) one block scans the tile sums with multiplier zc^256;
//   (c) every chunk adds zc^(distance to its tile top) * S(first chunk of the tile above).
// scan_tile: 256 (value, multiplier) maps, element e = 0 is the HIGHEST index; returns the composed map for element e.
template <class F>
__device__ __forceinline__ void scan_tile(uint32_t* sv, uint32_t* sm, int t) {
  for (int d = 1; d < 256; d <<= 1) {
    F v2, m2;
    const bool act = t >= d;
    if (act) {
      F vlo = F::load(sv + (t - d) * 8), mlo = F::load(sm + (t - d) * 8);
      F vhi = F::load(sv + t * 8), mhi = F::load(sm + t * 8);
      v2 = vhi + mhi * vlo;                               // apply the lower-index (higher chunk) map first
      m2 = mhi * mlo;
    }
    __syncthreads();
    if (act) {
      v2.store(sv + t * 8);
      m2.store(sm + t * 8);
    }
    __syncthreads();
  }
}
template <class P>
__device__ __forceinline__ void horner_tile_body(uint32_t* __restrict__ head, size_t chunks, const uint32_t* __restrict__ z,
                                                 uint32_t* __restrict__ tile_sum, uint32_t* sv, uint32_t* sm) {
  using F = Fp<P>;
  const int t = threadIdx.x;
  F zc = F::load(z);
  for (int i = 0; i < 5; i++) zc = zc.sqr();          // z^32 == z^POLY_CHUNK
  static_assert(POLY_CHUNK == 32, "zc exponent");
  const size_t c = (size_t)blockIdx.x * 256 + t;
  (c < chunks ? F::load(head + c * 8) : F::zero()).store(sv + (255 - t) * 8);
  (c < chunks ? zc : F::one()).store(sm + (255 - t) * 8);
  __syncthreads();
  scan_tile<F>(sv, sm, t);
  if (c < chunks) F::load(sv + (255 - t) * 8).store(head + c * 8);     // local S (as if nothing lay above the tile)
  if (t == 0) F::load(sv + 255 * 8).store(tile_sum + (size_t)blockIdx.x * 8);
}
template <class P>
__global__ __launch_bounds__(256) void horner_tile_kernel(uint32_t* __restrict__ head, size_t chunks,
                                                          const uint32_t* __restrict__ z, uint32_t* __restrict__ tile_sum) {
  __shared__ uint32_t sv[256 * 8], sm[256 * 8];
  horner_tile_body<P>(head, chunks, z, tile_sum, sv, sm);
}
template <class P>
__global__ __launch_bounds__(256) void horner_tile_batch_kernel(const HornerJob* __restrict__ jobs) {
  __shared__ uint32_t sv[256 * 8], sm[256 * 8];
  const HornerJob j = jobs[blockIdx.y];
  if (blockIdx.x >= j.tiles) return;                   // uniform per workgroup
  horner_tile_body<P>(j.head, j.chunks, j.z, j.tile_sum, sv, sm);
}
// one block: suffix scan of `count` values with a constant multiplier z^(32 * 256) per step, tiles of 256 from the top
template <class P>
__device__ __forceinline__ void horner_scan_body(uint32_t* __restrict__ head, size_t count, const uint32_t* __restrict__ z,
                                                 uint32_t* __restrict__ total, uint32_t* sv, uint32_t* sm) {
  using F = Fp<P>;
  const int t = threadIdx.x;
  F zc = F::load(z);
  for (int i = 0; i < 13; i++) zc = zc.sqr();         // z^(POLY_CHUNK * 256)
  F carry = F::zero();                                  // S of the element just above the current tile
  const size_t tiles = (count + 255) / 256;
  for (size_t tile = tiles; tile-- > 0;) {
    size_t c = tile * 256 + t;
    (c < count ? F::load(head + c * 8) : F::zero()).store(sv + (255 - t) * 8);
    (c < count ? zc : F::one()).store(sm + (255 - t) * 8);
    __syncthreads();
    scan_tile<F>(sv, sm, t);
    F v = F::load(sv + (255 - t) * 8), m = F::load(sm + (255 - t) * 8);
    F s = v + m * carry;
    if (c < count) s.store(head + c * 8);
    __syncthreads();
    F v0 = F::load(sv + 255 * 8), m0 = F::load(sm + 255 * 8);
    carry = v0 + m0 * carry;
    __syncthreads();
  }
  if (t == 0 && total) carry.store(total);               // S[0] = p(z)
}
template <class P>
__global__ __launch_bounds__(256) void horner_scan_kernel(uint32_t* __restrict__ head, size_t count,
                                                          const uint32_t* __restrict__ z, uint32_t* __restrict__ total) {
  __shared__ uint32_t sv[256 * 8], sm[256 * 8];
  horner_scan_body<P>(head, count, z, total, sv, sm);
}
template <class P>
__global__ __launch_bounds__(256) void horner_scan_batch_kernel(const HornerJob* __restrict__ jobs) {
  __shared__ uint32_t sv[256 * 8], sm[256 * 8];
  const HornerJob j = jobs[blockIdx.y];
  horner_scan_body<P>(j.tile_sum, j.tiles, j.z, j.ev, sv, sm);
}
template <class P>
__global__ __launch_bounds__(256) void horner_fix_kernel(uint32_t* __restrict__ head, size_t chunks,
                                                         const uint32_t* __restrict__ z, const uint32_t* __restrict__ tile_sum,
                                                         size_t tiles) {
  using F = Fp<P>;
  const size_t tile = blockIdx.x, c = tile * 256 + threadIdx.x;
  if (tile + 1 >= tiles || c >= chunks) return;
  F zc = F::load(z);
  for (int i = 0; i < 5; i++) zc = zc.sqr();
  F m = F::one();                                        // zc^(256 - t)
  for (uint32_t e = 256 - threadIdx.x, bit = 256; bit; bit >>= 1) {
    m = m.sqr();
    if (e & bit) m = m * zc;
  }
  (F::load(head + c * 8) + m * F::load(tile_sum + (tile + 1) * 8)).store(head + c * 8);
}

// phase 3: q_{i-1} = S_i for i = 1..n-1 (quotient of p by (X - z)); S_i = local Horner from the chunk top + z^k S_top
template <class P>
__global__ __launch_bounds__(256) void horner_replay_kernel(const uint32_t* __restrict__ p, size_t n,
                                                            const uint32_t* __restrict__ z,
                                                            const uint32_t* __restrict__ head, size_t chunks,
                                                            uint32_t* __restrict__ q) {
  using F = Fp<P>;
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= chunks) return;
  F zz = F::load(z);
  size_t lo = c * POLY_CHUNK, hi = lo + POLY_CHUNK < n ? lo + POLY_CHUNK : n;
  F acc = c + 1 < chunks ? F::load(head + (c + 1) * 8) : F::zero();      // S_hi
  for (size_t j = hi; j-- > lo;) {
    acc = acc * zz + F::load(p + j * 8);                                // S_j
    if (j >= 1) acc.store(q + (j - 1) * 8);
  }
}

template <class P>
static void poly_div_linear_t(zkp_ctx* ctx, const uint32_t* p, size_t n, const uint32_t* z_dev, uint32_t* q,
                              uint32_t* eval_dev) {
  hipStream_t st = ctx->cur->stream;
  size_t chunks = (n + POLY_CHUNK - 1) / POLY_CHUNK;
  const size_t tiles = (chunks + 255) / 256;
  uint32_t* head = ctx->poly_tmp.as<uint32_t>((chunks + tiles + 2) * 8);
  uint32_t* tile_sum = head + (chunks + 1) * 8;
  hipLaunchKernelGGL(horner_chunk_kernel<P>, dim3((chunks + 255) / 256), dim3(256), 0, st, p, n, z_dev, head, chunks);
  hipLaunchKernelGGL(horner_tile_kernel<P>, dim3(tiles), dim3(256), 0, st, head, chunks, z_dev, tile_sum);
  hipLaunchKernelGGL(horner_scan_kernel<P>, dim3(1), dim3(256), 0, st, tile_sum, tiles, z_dev, eval_dev);
  if (q && tiles > 1) hipLaunchKernelGGL(horner_fix_kernel<P>, dim3(tiles), dim3(256), 0, st, head, chunks, z_dev, (const uint32_t*)tile_sum, tiles);
  if (q) hipLaunchKernelGGL(horner_replay_kernel<P>, dim3((chunks + 255) / 256), dim3(256), 0, st, p, n, z_dev, head, chunks, q);
  ZKP_HIP(hipGetLastError());
}

void fr_vec_op(zkp_ctx* ctx, int curve, int op, const uint64_t* a, const uint64_t* b, const uint64_t* k_host,
               uint64_t* out, size_t n) {
  ZKP_REQUIRE(op >= 0 && op <= 5, ZKP_ERR_BAD_ARG);
  uint32_t* kd = ctx->poly_consts.as<uint32_t>(128);
  if (k_host) ZKP_HIP(hipMemcpyAsync(kd, k_host, 32, hipMemcpyHostToDevice, ctx->cur->stream));
  if (n == 0) return;
  auto launch = [&](auto tag) {
    using P = decltype(tag);
    hipLaunchKernelGGL(vec_op_kernel<P>, dim3((n + 255) / 256), dim3(256), 0, ctx->cur->stream,
                       reinterpret_cast<const uint32_t*>(a), reinterpret_cast<const uint32_t*>(b), kd,
                       reinterpret_cast<uint32_t*>(out), n, op);
  };
  if (curve == ZKP_BN254) launch(Bn254Fr{});
  else if (curve == ZKP_BLS12_381) launch(Bls381Fr{});
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  ZKP_HIP(hipGetLastError());
}

// Marlin, second round (ahp/prover.rs:246-305) on the product domain D:   out = r_alpha * (eta_c z_a z_b + eta_a z_a + eta_b z_b) - t * z
// from the evaluations over D of r_alpha, z_a, z_b, t, z (round 4: m(X) = eta_c z_a z_b + ... is formed pointwise instead of as
// coefficients, which needed a product of its own: three transforms of size |D| / 2 and eight element-wise launches less).
// k: 3 Fr (Montgomery) = eta_a, eta_b, eta_c
template <class P>
__global__ __launch_bounds__(256) void marlin_round2_prod_kernel(const uint32_t* __restrict__ ra, const uint32_t* __restrict__ za,
                                                                 const uint32_t* __restrict__ zb, const uint32_t* __restrict__ t,
                                                                 const uint32_t* __restrict__ z, const uint32_t* __restrict__ k,
                                                                 uint32_t* __restrict__ out, size_t n) {
  using F = Fp<P>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F a = F::load(za + i * 8), b = F::load(zb + i * 8);
  const F m = F::load(k + 16) * (a * b) + F::load(k) * a + F::load(k + 8) * b;
  (F::load(ra + i * 8) * m - F::load(t + i * 8) * F::load(z + i * 8)).store(out + i * 8);
}
void marlin_round2_prod(zkp_ctx* ctx, int curve, const uint64_t* ra, const uint64_t* za, const uint64_t* zb, const uint64_t* t,
                        const uint64_t* z, const uint64_t* k_host, uint64_t* out, size_t n) {
  if (n == 0) return;
  hipStream_t st = ctx->cur->stream;
  uint32_t* kd = ctx->poly_consts.as<uint32_t>(128) + 64;
  ZKP_HIP(hipMemcpyAsync(kd, k_host, 3 * 32, hipMemcpyHostToDevice, st));
  auto launch = [&](auto tag) {
    using P = decltype(tag);
    hipLaunchKernelGGL(marlin_round2_prod_kernel<P>, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(ra),
                       reinterpret_cast<const uint32_t*>(za), reinterpret_cast<const uint32_t*>(zb), reinterpret_cast<const uint32_t*>(t),
                       reinterpret_cast<const uint32_t*>(z), (const uint32_t*)kd, reinterpret_cast<uint32_t*>(out), n);
  };
  if (curve == ZKP_BN254) launch(Bn254Fr{});
  else if (curve == ZKP_BLS12_381) launch(Bls381Fr{});
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  ZKP_HIP(hipGetLastError());
}

// Marlin, third round (ahp/prover.rs:340-377): the evaluations over K of   v_H(alpha) v_H(beta) sum_m eta_m val_m / ((beta - row_m)(alpha - col_m))
// in ONE kernel (round 4; was three batch inversions + ~12 element-wise launches over |K|-element vectors).  Montgomery's trick per lane
// over a strided chunk of the TRIPLE products d_0 d_1 d_2 (one Fermat inversion per 3 * POLY_CHUNK denominators instead of per
// POLY_CHUNK); a zero denominator contributes a zero term, exactly as ark's batch_inversion leaves zeros in place.
// k: 5 Fr (Montgomery) = alpha, beta, eta_m * v_H(alpha) v_H(beta) for m = 0, 1, 2
struct MarlinOnK {
  const uint32_t* v[9];       // [m][row, col, val]
};
template <class P>
__global__ __launch_bounds__(256) void marlin_t3_evals_kernel(MarlinOnK in, const uint32_t* __restrict__ k, uint32_t* __restrict__ out,
                                                              size_t n, size_t lanes) {
  using F = Fp<P>;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  const F alpha = F::load(k), beta = F::load(k + 8);
  auto dens = [&](size_t i, F* e, bool* z) {
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const F d = (beta - F::load(in.v[3 * m] + i * 8)) * (alpha - F::load(in.v[3 * m + 1] + i * 8));
      z[m] = d.is_zero();
      e[m] = z[m] ? F::one() : d;
    }
  };
  F pref[POLY_CHUNK];
  F acc = F::one();
  int cnt = 0;
  for (size_t i = t; i < n && cnt < POLY_CHUNK; i += lanes, cnt++) {
    F e[3];
    bool z[3];
    dens(i, e, z);
    pref[cnt] = acc;
    acc = acc * (e[0] * e[1] * e[2]);
  }
  F inv = acc.inv();
  for (int c = cnt - 1; c >= 0; c--) {
    const size_t i = t + (size_t)c * lanes;
    F e[3];
    bool z[3];
    dens(i, e, z);
    const F p12 = e[1] * e[2], p02 = e[0] * e[2], p01 = e[0] * e[1];
    const F di = inv * pref[c];                                        // 1 / (e0 e1 e2)
    inv = inv * (e[0] * p12);
    F r = F::zero();
    if (!z[0]) r = r + F::load(k + 16) * (F::load(in.v[2] + i * 8) * (di * p12));
    if (!z[1]) r = r + F::load(k + 24) * (F::load(in.v[5] + i * 8) * (di * p02));
    if (!z[2]) r = r + F::load(k + 32) * (F::load(in.v[8] + i * 8) * (di * p01));
    r.store(out + i * 8);
  }
}
void marlin_t3_evals(zkp_ctx* ctx, int curve, const uint64_t* const* on_k, const uint64_t* k_host, uint64_t* out, size_t n) {
  if (n == 0) return;
  hipStream_t st = ctx->cur->stream;
  uint32_t* kd = ctx->poly_consts.as<uint32_t>(128) + 64;
  ZKP_HIP(hipMemcpyAsync(kd, k_host, 5 * 32, hipMemcpyHostToDevice, st));
  MarlinOnK in;
  for (int j = 0; j < 9; j++) in.v[j] = reinterpret_cast<const uint32_t*>(on_k[j]);
  const size_t lanes = (n + POLY_CHUNK - 1) / POLY_CHUNK;
  auto launch = [&](auto tag) {
    using P = decltype(tag);
    hipLaunchKernelGGL(marlin_t3_evals_kernel<P>, dim3((lanes + 255) / 256), dim3(256), 0, st, in, (const uint32_t*)kd,
                       reinterpret_cast<uint32_t*>(out), n, lanes);
  };
  if (curve == ZKP_BN254) launch(Bn254Fr{});
  else if (curve == ZKP_BLS12_381) launch(Bls381Fr{});
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  ZKP_HIP(hipGetLastError());
}

// Marlin, third round (ahp/prover.rs:380-417), evaluations over the domain B in ONE pass (round 4; was 24 element-wise launches over
// 2^23-element vectors):   den_m = row_col_m - alpha row_m - beta col_m + alpha beta,
//     a = v_H(alpha) v_H(beta) sum_m eta_m val_m den_{m+1} den_{m+2},   b = den_0 den_1 den_2,   out = a - b * t
// with t = the evaluations of t(X) over B: (a - b t)(X) has degree < |B|, so its interpolation IS a_poly - b_poly * t_poly.
// k: 6 Fr (Montgomery) = alpha, beta, alpha*beta, eta_m * v_H(alpha) v_H(beta) for m = 0, 1, 2
struct MarlinOnB {
  const uint32_t* v[12];      // [m][row, col, val, row_col]
};
template <class P>
__global__ __launch_bounds__(256) void marlin_h2_numerator_kernel(MarlinOnB in, const uint32_t* __restrict__ t, const uint32_t* __restrict__ k,
                                                                  uint32_t* __restrict__ out, size_t n) {
  using F = Fp<P>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F alpha = F::load(k), beta = F::load(k + 8), ab = F::load(k + 16);
  F den[3];
#pragma unroll
  for (int m = 0; m < 3; m++)
    den[m] = F::load(in.v[4 * m + 3] + i * 8) - alpha * F::load(in.v[4 * m] + i * 8) - beta * F::load(in.v[4 * m + 1] + i * 8) + ab;
  const F p01 = den[0] * den[1], p12 = den[1] * den[2], p20 = den[2] * den[0];
  F a = F::load(k + 24) * (F::load(in.v[2] + i * 8) * p12);
  a = a + F::load(k + 32) * (F::load(in.v[6] + i * 8) * p20);
  a = a + F::load(k + 40) * (F::load(in.v[10] + i * 8) * p01);
  (a - (p01 * den[2]) * F::load(t + i * 8)).store(out + i * 8);
}
void marlin_h2_numerator(zkp_ctx* ctx, int curve, const uint64_t* const* on_b, const uint64_t* t, const uint64_t* k_host, uint64_t* out,
                         size_t n) {
  if (n == 0) return;
  hipStream_t st = ctx->cur->stream;
  uint32_t* kd = ctx->poly_consts.as<uint32_t>(128) + 64;
  ZKP_HIP(hipMemcpyAsync(kd, k_host, 6 * 32, hipMemcpyHostToDevice, st));
  MarlinOnB in;
  for (int j = 0; j < 12; j++) in.v[j] = reinterpret_cast<const uint32_t*>(on_b[j]);
  auto launch = [&](auto tag) {
    using P = decltype(tag);
    hipLaunchKernelGGL(marlin_h2_numerator_kernel<P>, dim3((n + 255) / 256), dim3(256), 0, st, in, reinterpret_cast<const uint32_t*>(t),
                       (const uint32_t*)kd, reinterpret_cast<uint32_t*>(out), n);
  };
  if (curve == ZKP_BN254) launch(Bn254Fr{});
  else if (curve == ZKP_BLS12_381) launch(Bls381Fr{});
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  ZKP_HIP(hipGetLastError());
}

// out[i] = sum_k coeff[k] * x[col[k]], k in [row_ptr[i], row_ptr[i+1])   (z_a = A z, t = (eta_a A + ...)^T r_alpha, ...)
// One lane per row; rows longer than SPMV_LONG (the column of the constant `one` in a transposed R1CS matrix holds a
// constant fraction of all entries) are queued in long_list = [count, row ids...] and reduced by a whole block each.
constexpr uint32_t SPMV_LONG = 128, SPMV_LIST_CAP = 1u << 16;
template <class P>
__device__ __forceinline__ Fp<P> spmv_term(const uint32_t* __restrict__ col, const uint32_t* __restrict__ coeff,
                                           const uint32_t* __restrict__ x, uint32_t k, const Fp<P>& one) {
  using F = Fp<P>;
  F v = F::load(x + (size_t)col[k] * 8), cf = F::load(coeff + (size_t)k * 8);
  return cf == one ? v : v * cf;
}
template <class P>
__global__ __launch_bounds__(256) void spmv_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                   const uint32_t* __restrict__ coeff, const uint32_t* __restrict__ x,
                                                   size_t nrows, uint32_t* __restrict__ out, uint32_t* __restrict__ long_list) {
  using F = Fp<P>;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const uint32_t b = row_ptr[i], e = row_ptr[i + 1];
  if (e - b > SPMV_LONG) {
    uint32_t slot = atomicAdd(long_list, 1u);
    if (slot < SPMV_LIST_CAP) {
      long_list[1 + slot] = (uint32_t)i;
      return;
    }
  }
  F acc = F::zero();
  const F one = F::one();
  for (uint32_t k = b; k < e; k++) acc = acc + spmv_term<P>(col, coeff, x, k, one);
  acc.store(out + i * 8);
}
// Long rows in two steps so that ONE huge row (the column of the constant `one` holds ~40 % of the entries of a transposed
// R1CS matrix: 10^6 terms) is spread over the whole chip instead of one workgroup:
//   partial: (row, chunk of SPMV_CHUNK terms) pairs are dealt round-robin to the workgroups; every workgroup walks the row
//            list with the same running partial-slot counter, so no index structure is needed;
//   reduce:  one workgroup per row sums that row's partials.
// Rows whose partial slots would not fit SPMV_PART_CAP are summed whole by their reduce workgroup.
constexpr uint32_t SPMV_CHUNK = 8192, SPMV_PART_CAP = 1u << 16;
template <class P>
__device__ __forceinline__ Fp<P> block_sum(Fp<P> acc, uint32_t* red, uint32_t tid) {
  using F = Fp<P>;
  acc.store(red + tid * 8);
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      acc = acc + F::load(red + (tid + s) * 8);
      acc.store(red + tid * 8);
    }
    __syncthreads();
  }
  return acc;                                              // valid in lane 0
}
template <class P>
__global__ __launch_bounds__(256) void spmv_long_partial_kernel(const uint32_t* __restrict__ row_ptr,
                                                                const uint32_t* __restrict__ col,
                                                                const uint32_t* __restrict__ coeff,
                                                                const uint32_t* __restrict__ x,
                                                                const uint32_t* __restrict__ long_list,
                                                                uint32_t* __restrict__ partial) {
  using F = Fp<P>;
  __shared__ uint32_t red[256 * 8];
  const uint32_t count = min(long_list[0], SPMV_LIST_CAP), tid = threadIdx.x;
  const F one = F::one();
  uint32_t slot = 0;                                       // first partial slot of the current row
  for (uint32_t r = 0; r < count; r++) {
    const uint32_t i = long_list[1 + r], b = row_ptr[i], e = row_ptr[i + 1];
    const uint32_t nch = (e - b + SPMV_CHUNK - 1) / SPMV_CHUNK;
    if (slot + nch > SPMV_PART_CAP) break;                 // the rest is summed whole in the reduce kernel
    for (uint32_t c = blockIdx.x; c < nch; c += gridDim.x) {
      const uint32_t lo = b + c * SPMV_CHUNK, hi = min(e, lo + SPMV_CHUNK);
      F acc = F::zero();
      for (uint32_t k = lo + tid; k < hi; k += 256) acc = acc + spmv_term<P>(col, coeff, x, k, one);
      acc = block_sum<P>(acc, red, tid);
      if (tid == 0) acc.store(partial + (size_t)(slot + c) * 8);
      __syncthreads();
    }
    slot += nch;
  }
}
template <class P>
__global__ __launch_bounds__(256) void spmv_long_reduce_kernel(const uint32_t* __restrict__ row_ptr,
                                                               const uint32_t* __restrict__ col,
                                                               const uint32_t* __restrict__ coeff,
                                                               const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                               const uint32_t* __restrict__ long_list,
                                                               const uint32_t* __restrict__ partial) {
  using F = Fp<P>;
  __shared__ uint32_t red[256 * 8];
  const uint32_t count = min(long_list[0], SPMV_LIST_CAP), tid = threadIdx.x;
  const F one = F::one();
  uint32_t slot = 0;
  bool fits = true;
  for (uint32_t r = 0; r < count; r++) {
    const uint32_t i = long_list[1 + r], b = row_ptr[i], e = row_ptr[i + 1];
    const uint32_t nch = (e - b + SPMV_CHUNK - 1) / SPMV_CHUNK;
    fits = fits && (slot + nch <= SPMV_PART_CAP);
    if (r % gridDim.x == blockIdx.x) {
      F acc = F::zero();
      if (fits) {
        for (uint32_t c = tid; c < nch; c += 256) acc = acc + F::load(partial + (size_t)(slot + c) * 8);
      } else {
        for (uint32_t k = b + tid; k < e; k += 256) acc = acc + spmv_term<P>(col, coeff, x, k, one);
      }
      acc = block_sum<P>(acc, red, tid);
      if (tid == 0) acc.store(out + (size_t)i * 8);
      __syncthreads();
    }
    if (fits) slot += nch;
  }
}
// out[i] = idx[i] < 0 ? 0 : in[idx[i]]
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ in, const int32_t* __restrict__ idx, size_t n,
                                                     uint4* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t j = idx[i];
  uint4 z = make_uint4(0, 0, 0, 0);
  out[2 * i] = j < 0 ? z : in[2 * (size_t)j];
  out[2 * i + 1] = j < 0 ? z : in[2 * (size_t)j + 1];
}
// p = q (X^n - 1) + rem.  With s[i] = sum_{k>=0} p[i + k n] (strided suffix sums): q[i] = s[i + n], rem[i] = s[i].
// View p as R = ceil(len/n) rows of n residues; rows are cut into chunks of C rows.  Pass 1 sums each (chunk, residue),
// pass 2 turns the chunk sums into exclusive suffix carries per residue, pass 3 replays each chunk top-down from its
// carry and writes q / rem.  When a residue has few rows (the folds by |H|, |K|: R <= 4) only pass 3 runs.
template <class P>
__global__ __launch_bounds__(256) void vfold_sum_kernel(const uint32_t* __restrict__ p, size_t len, size_t n, size_t C,
                                                        size_t T, uint32_t* __restrict__ part) {
  using F = Fp<P>;
  size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= T * n) return;
  const size_t t = id / n, j = id % n;
  F acc = F::zero();
  for (size_t k = t * C; k < (t + 1) * C; k++) {
    size_t e = k * n + j;
    if (e < len) acc = acc + F::load(p + e * 8);
  }
  acc.store(part + id * 8);
}
template <class P>
__global__ __launch_bounds__(256) void vfold_carry_kernel(uint32_t* __restrict__ part, size_t n, size_t T) {
  using F = Fp<P>;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  F acc = F::zero();
  for (size_t t = T; t-- > 0;) {
    F v = F::load(part + (t * n + j) * 8);
    acc.store(part + (t * n + j) * 8);
    acc = acc + v;
  }
}
template <class P>
__global__ __launch_bounds__(256) void vfold_replay_kernel(const uint32_t* __restrict__ p, size_t len, size_t n, size_t C,
                                                           size_t T, const uint32_t* __restrict__ carry,
                                                           uint32_t* __restrict__ q, uint32_t* __restrict__ rem) {
  using F = Fp<P>;
  size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= T * n) return;
  const size_t t = id / n, j = id % n;
  F acc = carry ? F::load(carry + id * 8) : F::zero();
  for (size_t k = (t + 1) * C; k-- > t * C;) {
    size_t e = k * n + j;
    if (e < len) acc = acc + F::load(p + e * 8);
    if (k == 0) {
      if (rem) acc.store(rem + j * 8);
    } else if (q && e < len) {
      acc.store(q + (e - n) * 8);
    }
  }
}

void fr_spmv(zkp_ctx* ctx, int curve, const uint32_t* row_ptr, const uint32_t* col, const uint64_t* coeff, size_t nrows,
             const uint64_t* x, uint64_t* out) {
  if (nrows == 0) return;
  hipStream_t st = ctx->cur->stream;
  uint32_t* list = ctx->spmv_list.as<uint32_t>(1 + SPMV_LIST_CAP + (size_t)SPMV_PART_CAP * 8);
  uint32_t* partial = list + 1 + SPMV_LIST_CAP;
  ZKP_HIP(hipMemsetAsync(list, 0, sizeof(uint32_t), st));
  auto launch = [&](auto tag) {
    using P = decltype(tag);
    hipLaunchKernelGGL(spmv_kernel<P>, dim3((nrows + 255) / 256), dim3(256), 0, st, row_ptr, col,
                       reinterpret_cast<const uint32_t*>(coeff), reinterpret_cast<const uint32_t*>(x), nrows,
                       reinterpret_cast<uint32_t*>(out), list);
    hipLaunchKernelGGL(spmv_long_partial_kernel<P>, dim3(512), dim3(256), 0, st, row_ptr, col,
                       reinterpret_cast<const uint32_t*>(coeff), reinterpret_cast<const uint32_t*>(x), list, partial);
    hipLaunchKernelGGL(spmv_long_reduce_kernel<P>, dim3(256), dim3(256), 0, st, row_ptr, col,
                       reinterpret_cast<const uint32_t*>(coeff), reinterpret_cast<const uint32_t*>(x),
                       reinterpret_cast<uint32_t*>(out), list, partial);
  };
  if (curve == ZKP_BN254) launch(Bn254Fr{});
  else if (curve == ZKP_BLS12_381) launch(Bls381Fr{});
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  ZKP_HIP(hipGetLastError());
}
void fr_gather(zkp_ctx* ctx, const uint64_t* in, const int32_t* idx, size_t n, uint64_t* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->cur->stream,
                     reinterpret_cast<const uint4*>(in), idx, n, reinterpret_cast<uint4*>(out));
  ZKP_HIP(hipGetLastError());
}
void poly_vanishing_fold(zkp_ctx* ctx, int curve, const uint64_t* p, size_t len, size_t n, uint64_t* q, uint64_t* rem) {
  ZKP_REQUIRE(n > 0, ZKP_ERR_BAD_ARG);
  hipStream_t st = ctx->cur->stream;
  if (n == 1 && len > 4096 && q && rem) {
    // X^1 - 1 (Marlin's division by v_X with one public input): a running sum over the WHOLE vector.  The strided-suffix-sum
    // kernels below would give it 2 sqrt(len) lanes with sqrt(len) serial additions each (1.5 ms for 2^20 coefficients); it is the
    // division by (X - 1): blocked-scan Horner with z = 1, quotient and p(1) = the remainder in one go.
    auto go = [&](auto tag) {
      using P = decltype(tag);
      uint32_t* zd = ctx->poly_consts.as<uint32_t>(128) + 32;           // (words 0..23 belong to fr_vec_op / poly_div_linear)
      uint32_t one[8];
      for (int i = 0; i < 8; i++) one[i] = P::ONE[i];
      ZKP_HIP(hipMemcpyAsync(zd, one, 32, hipMemcpyHostToDevice, st));   // pageable source: staged before the call returns
      poly_div_linear_t<P>(ctx, reinterpret_cast<const uint32_t*>(p), len, zd, reinterpret_cast<uint32_t*>(q),
                           reinterpret_cast<uint32_t*>(rem));
    };
    if (curve == ZKP_BN254) go(Bn254Fr{});
    else if (curve == ZKP_BLS12_381) go(Bls381Fr{});
    else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
    return;
  }
  const size_t R = std::max<size_t>((len + n - 1) / n, 1);
  size_t C = R;
  if (R > 8) {
    C = 8;
    while (C * C < R) C++;
  }
  const size_t T = (R + C - 1) / C, work = T * n;
  auto launch = [&](auto tag) {
    using P = decltype(tag);
    const uint32_t* pp = reinterpret_cast<const uint32_t*>(p);
    uint32_t* carry = nullptr;
    if (T > 1) {
      carry = ctx->poly_tmp.as<uint32_t>(work * 8);
      hipLaunchKernelGGL(vfold_sum_kernel<P>, dim3((work + 255) / 256), dim3(256), 0, st, pp, len, n, C, T, carry);
      hipLaunchKernelGGL(vfold_carry_kernel<P>, dim3((n + 255) / 256), dim3(256), 0, st, carry, n, T);
    }
    hipLaunchKernelGGL(vfold_replay_kernel<P>, dim3((work + 255) / 256), dim3(256), 0, st, pp, len, n, C, T,
                       (const uint32_t*)carry, reinterpret_cast<uint32_t*>(q), reinterpret_cast<uint32_t*>(rem));
  };
  if (curve == ZKP_BN254) launch(Bn254Fr{});
  else if (curve == ZKP_BLS12_381) launch(Bls381Fr{});
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  ZKP_HIP(hipGetLastError());
}

void fr_batch_inverse(zkp_ctx* ctx, int curve, uint64_t* v, size_t n) {
  if (n == 0) return;
  size_t lanes = (n + POLY_CHUNK - 1) / POLY_CHUNK;
  auto launch = [&](auto tag) {
    using P = decltype(tag);
    hipLaunchKernelGGL(batch_inverse_kernel<P>, dim3((lanes + 255) / 256), dim3(256), 0, ctx->cur->stream,
                       reinterpret_cast<uint32_t*>(v), n, lanes);
  };
  if (curve == ZKP_BN254) launch(Bn254Fr{});
  else if (curve == ZKP_BLS12_381) launch(Bls381Fr{});
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  ZKP_HIP(hipGetLastError());
}

// out[k] = p_k(z_k) for `count` polynomials (the 21 evaluations of lib.rs:147-156): every Horner chain is enqueued before the single
// read-back, instead of one host round trip per polynomial.  z_host / out_host: count x 4 u64 (Montgomery).
void poly_evaluate_batch(zkp_ctx* ctx, int curve, size_t count, const uint64_t* const* p, const size_t* n, const uint64_t* z_host,
                         uint64_t* out_host) {
  if (count == 0) return;
  ZKP_REQUIRE(curve == ZKP_BN254 || curve == ZKP_BLS12_381, ZKP_ERR_UNSUPPORTED_CURVE);
  hipStream_t st = ctx->cur->stream;
  size_t words = 16 * count;                                          // z_k and p_k(z_k)
  std::vector<size_t> off(count);
  for (size_t k = 0; k < count; k++) {
    const size_t chunks = (n[k] + POLY_CHUNK - 1) / POLY_CHUNK, tiles = (chunks + 255) / 256;
    off[k] = words;
    words += (chunks + tiles + 2) * 8;
  }
  const size_t job_words = (count * sizeof(HornerJob) + 3) / 4;
  uint32_t* buf = ctx->poly_tmp.as<uint32_t>(words + job_words + 8);
  uint32_t* zd = buf;
  uint32_t* ev = buf + 8 * count;
  HornerJob* jobs_dev = reinterpret_cast<HornerJob*>(buf + ((words + 3) & ~(size_t)3));
  ZKP_HIP(hipMemcpyAsync(zd, z_host, 32 * count, hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemsetAsync(ev, 0, 32 * count, st));                     // an empty polynomial evaluates to zero
  std::vector<HornerJob> jobs;
  size_t max_chunks = 0, max_tiles = 0;
  for (size_t k = 0; k < count; k++) {
    if (n[k] == 0) continue;
    const size_t chunks = (n[k] + POLY_CHUNK - 1) / POLY_CHUNK, tiles = (chunks + 255) / 256;
    uint32_t* head = buf + off[k];
    jobs.push_back(HornerJob{reinterpret_cast<const uint32_t*>(p[k]), n[k], chunks, tiles, zd + 8 * k, head, head + (chunks + 1) * 8,
                             ev + 8 * k});
    max_chunks = std::max(max_chunks, chunks);
    max_tiles = std::max(max_tiles, tiles);
  }
  if (!jobs.empty()) {
    ZKP_HIP(hipMemcpyAsync(jobs_dev, jobs.data(), jobs.size() * sizeof(HornerJob), hipMemcpyHostToDevice, st));
    ZKP_HIP(hipStreamSynchronize(st));                                // `jobs` (pageable) is consumed; nothing is in flight before it anyway
    auto go = [&](auto tag) {
      using P = decltype(tag);
      hipLaunchKernelGGL(horner_chunk_batch_kernel<P>, dim3((max_chunks + 255) / 256, jobs.size()), dim3(256), 0, st, jobs_dev);
      hipLaunchKernelGGL(horner_tile_batch_kernel<P>, dim3(max_tiles, jobs.size()), dim3(256), 0, st, jobs_dev);
      hipLaunchKernelGGL(horner_scan_batch_kernel<P>, dim3(1, jobs.size()), dim3(256), 0, st, jobs_dev);
    };
    if (curve == ZKP_BN254) go(Bn254Fr{});
    else go(Bls381Fr{});
  }
  ZKP_HIP(hipGetLastError());
  ZKP_HIP(hipMemcpyAsync(out_host, ev, 32 * count, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
}

// q (n-1 coeffs, may be nullptr) = p / (X - z) ; eval_out (host, may be nullptr) = p(z)
void poly_div_linear(zkp_ctx* ctx, int curve, const uint64_t* p, size_t n, const uint64_t* z_host, uint64_t* q,
                     uint64_t* eval_out_host) {
  uint32_t* zd = ctx->poly_consts.as<uint32_t>(128);
  ZKP_HIP(hipMemcpyAsync(zd, z_host, 32, hipMemcpyHostToDevice, ctx->cur->stream));
  uint32_t* ev = zd + 16;
  if (n == 0) {
    ZKP_HIP(hipMemsetAsync(ev, 0, 32, ctx->cur->stream));
  } else if (curve == ZKP_BN254) {
    poly_div_linear_t<Bn254Fr>(ctx, reinterpret_cast<const uint32_t*>(p), n, zd, reinterpret_cast<uint32_t*>(q), ev);
  } else if (curve == ZKP_BLS12_381) {
    poly_div_linear_t<Bls381Fr>(ctx, reinterpret_cast<const uint32_t*>(p), n, zd, reinterpret_cast<uint32_t*>(q), ev);
  } else {
    throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
  }
  if (eval_out_host) {
    ZKP_HIP(hipMemcpyAsync(eval_out_host, ev, 32, hipMemcpyDeviceToHost, ctx->cur->stream));
    ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  }
}

}  // namespace zkp
