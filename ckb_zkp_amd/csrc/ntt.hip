// Radix-2 NTT over Fr for gfx950: multi-pass Stockham (autosort) with LDS tiles.
//
// Replaces ark-poly 0.2 `GeneralEvaluationDomain::<Fr>::{fft,ifft,coset_fft,coset_ifft}_in_place`
// (call sites: /root/reference/groth16/src/r1cs_to_qap.rs:144-148,161-162,169 and every
// interpolate/fft/evaluate_over_domain in /root/reference/marlin/src/ahp/prover.rs).
// Contract: natural order in, natural order out, out[i] = sum_j in[j] w^(ij); inverse uses w^-1 and
// multiplies by 1/N; coset_fft scales coefficient j by g^j first, coset_ifft scales by g^-j last.
//
// Decomposition (derived in DESIGN.md §NTT): log N = S_1 + ... + S_P, S_p <= 7.  State before pass p is
// "B interleaved sub-transforms of size M" (B*M = N, B = 2^(S_1+..+S_{p-1})):  in[b + B*j].
// Pass p with R = 2^S:   j = j1*(M/R) + j',   k = k1 + R*k'
//     Y_b[k1][j'] = w_M^(j' k1) * sum_{j1} in[b + B*(j1*(M/R) + j')] * w_R^(j1 k1)
// stored at  b + B*k1 + B*R*j'   (new batch index b + B*k1, new B' = B*R).  Reads are runs of C
// consecutive elements (the element index b + B*j' is a *contiguous* range [0, N/R) for every j1),
// writes are runs of >= min(C, B) consecutive elements -> every pass is coalesced in both directions and
// the last pass lands in natural order: no bit-reversal pass, no transposes.
//
// One workgroup = one LDS tile of R rows x C columns (R*C = 1024 elements, 32 KiB, limb-major SoA so that
// ds_read/ds_write_b32 of consecutive columns hit consecutive banks).  The R-point column transforms are
// S radix-2 DIF stages in LDS with the w_R twiddle tile in LDS; the inter-pass twiddle w_N^(B j' k1) comes
// from a two-level table (w^lo, w^(hi<<h)), both L2-resident (<= 256 KiB at 2^24).
//
// Arithmetic is 256-bit Montgomery on the integer VALU (v_mad_u64_u32); there is no MFMA in this path.
#include <algorithm>
#include <cstdlib>

#include "ctx.hpp"
#include "field_dev.hpp"
#include "unsat_dev.hpp"

namespace zkp {

constexpr int NTT_SMAX = 9;            // max radix bits per pass (default plan)
// Round 5 (profiles/r05_ntt_two_pass.txt): a pass takes up to 9 radix bits on the same 1024-element tile (C = 1024 >> S columns per
// row) — 2^22 .. 2^24 in three passes instead of four (-5 .. -8 % per transform), 2^18 in two; 2^20 / 2^21 keep 7+7+6 / 7+7+7.
// ZKP_NTT_SMAX=7 restores the round-4 plans; 10 gives 2^20 in two passes (10 + 10, one column per row): measured +14 %, killed.
// Read once per process (the cached inter-pass tables depend on the plan).
static int ntt_smax() {
  static const int v = [] {
    const char* e = getenv("ZKP_NTT_SMAX");
    int s = e ? atoi(e) : NTT_SMAX;
    return s < 4 ? 4 : (s > 10 ? 10 : s);
  }();
  return v;
}
#ifndef ZKP_NTT_TILE_LOG
#define ZKP_NTT_TILE_LOG 10
#endif
#ifndef ZKP_NTT_THREADS
#define ZKP_NTT_THREADS 256
#endif
// (round 3, stand-alone fft at 2^20 / 2^22 / 2^23 — tools/ntt_time.py: 1024 elements x 256 threads 0.164 / 0.640 / 1.225 ms; 512 x 128:
//  0.195 / 0.668 / 1.250; 512 x 256: 0.175 / 0.688 / 1.314; 1024 x 512: 0.165 / 0.657 / 1.259; 2048 x 512: 0.181 / 0.770 / 1.497.
//  At 2^23 a pass runs at 74 % of its VALU-instruction bound, at 2^20 at 50 %: per-launch ramp, not the tile shape, is what is left.)
constexpr int NTT_TILE_LOG = ZKP_NTT_TILE_LOG;       // elements per LDS tile (1024 * 32 B = 32 KiB)
constexpr int NTT_THREADS = ZKP_NTT_THREADS;
constexpr int NTT_SUB_LOG = 10;        // sub-FFT twiddle table covers R <= 2^10 (the default plan uses R <= 2^7: every 8th entry)
constexpr int NTT_FULL_MAX_LOG = 24;   // full-size twiddle / coset tables up to this domain size (32 B * N each)
constexpr int NTT_MAX_BATCH = 4;       // independent transforms per launch (blockIdx.y)

// ------------------------------------------------------------------------------------------- tables
template <class P>
__global__ void ntt_setup_kernel(uint32_t* consts, int log_n, int h) {
  // consts[k*8..]: 0 w, 1 w^-1, 2 w^(2^h), 3 w^-(2^h), 4 g^(2^h), 5 g^-(2^h), 6 1/N, 7 w_sub, 8 w_sub^-1
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  using F = Fp<P>;
  F w, wi, g, gi;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    w.v[i] = P::ROOT[i];
    wi.v[i] = P::ROOT_INV[i];
    g.v[i] = P::GEN[i];
    gi.v[i] = P::GEN_INV[i];
  }
  F wsub = w, wsubi = wi;
  for (int i = 0; i < P::TWO_ADICITY - log_n; i++) {
    w = w.sqr();
    wi = wi.sqr();
  }
  int sub = log_n < NTT_SUB_LOG ? log_n : NTT_SUB_LOG;
  for (int i = 0; i < P::TWO_ADICITY - sub; i++) {
    wsub = wsub.sqr();
    wsubi = wsubi.sqr();
  }
  F wh = w, wih = wi, gh = g, gih = gi;
  for (int i = 0; i < h; i++) {
    wh = wh.sqr();
    wih = wih.sqr();
    gh = gh.sqr();
    gih = gih.sqr();
  }
  F n = F::zero();
  n.v[log_n / 32] = 1u << (log_n % 32);
  F ninv = n.to_mont().inv();
  w.store(consts + 0 * 8);
  wi.store(consts + 1 * 8);
  wh.store(consts + 2 * 8);
  wih.store(consts + 3 * 8);
  gh.store(consts + 4 * 8);
  gih.store(consts + 5 * 8);
  ninv.store(consts + 6 * 8);
  wsub.store(consts + 7 * 8);
  wsubi.store(consts + 8 * 8);
  g.store(consts + 9 * 8);
  gi.store(consts + 10 * 8);
}

// out[i] = base^i * (mult ? *mult : 1)
template <class P>
__global__ void pow_table_kernel(uint32_t* out, const uint32_t* base, const uint32_t* mult, uint32_t count) {
  using F = Fp<P>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  F b = F::load(base);
  F r = b.pow_u64(i);
  if (mult) r = r * F::load(mult);
  r.store(out + (size_t)i * 8);
}

template <class P>
static NttTables& get_tables(zkp_ctx* ctx, int curve, int log_n) {
  auto key = std::make_pair(curve, log_n);
  auto it = ctx->ntt_tables.find(key);
  if (it != ctx->ntt_tables.end()) return it->second;
  NttTables t;
  t.log_n = log_n;
  t.h = (log_n + 1) / 2;
  size_t lo = (size_t)1 << t.h, hi = (size_t)1 << (log_n - t.h);
  int sub = std::min(log_n, NTT_SUB_LOG);
  size_t subn = sub > 0 ? ((size_t)1 << sub) / 2 : 1;
  if (subn == 0) subn = 1;
  size_t words = (4 * (lo + hi) + hi + 2 * subn + 16) * 8;
  uint32_t* blk;
  if (hipMalloc(&blk, words * 4) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  t.block = blk;
  uint32_t* p = blk;
  uint32_t* consts = p; p += 16 * 8;
  t.w_lo = p; p += lo * 8;
  t.w_hi = p; p += hi * 8;
  t.wi_lo = p; p += lo * 8;
  t.wi_hi = p; p += hi * 8;
  t.g_lo = p; p += lo * 8;
  t.g_hi = p; p += hi * 8;
  t.gi_lo = p; p += lo * 8;
  t.gi_hi = p; p += hi * 8;
  t.g_hi_n = p; p += hi * 8;
  t.sub_fwd = p; p += subn * 8;
  t.sub_inv = p; p += subn * 8;
  t.n_inv = consts + 6 * 8;
  hipStream_t s = ctx->cur->stream;
  hipLaunchKernelGGL(ntt_setup_kernel<P>, dim3(1), dim3(64), 0, s, consts, log_n, t.h);
  auto gen = [&](uint32_t* out, uint32_t* base, uint32_t* mult, size_t cnt) {
    hipLaunchKernelGGL(pow_table_kernel<P>, dim3((cnt + 255) / 256), dim3(256), 0, s, out, base, mult, (uint32_t)cnt);
  };
  gen(t.w_lo, consts + 0 * 8, nullptr, lo);
  gen(t.w_hi, consts + 2 * 8, nullptr, hi);
  gen(t.wi_lo, consts + 1 * 8, nullptr, lo);
  gen(t.wi_hi, consts + 3 * 8, nullptr, hi);
  gen(t.g_lo, consts + 9 * 8, nullptr, lo);
  gen(t.g_hi, consts + 4 * 8, nullptr, hi);
  gen(t.gi_lo, consts + 10 * 8, nullptr, lo);
  gen(t.gi_hi, consts + 5 * 8, consts + 6 * 8, hi);   // g^-(i<<h) / N
  gen(t.g_hi_n, consts + 4 * 8, consts + 6 * 8, hi);  // g^(i<<h) / N
  gen(t.sub_fwd, consts + 7 * 8, nullptr, subn);
  gen(t.sub_inv, consts + 8 * 8, nullptr, subn);
  ZKP_HIP(hipGetLastError());
  ZKP_HIP(hipStreamSynchronize(s));      // one-time: the tables are shared by every lane / stream afterwards
  ctx->ntt_tables[key] = t;
  return ctx->ntt_tables[key];
}

// ------------------------------------------------------------------------------------------- pass
struct NttPassArgs {
  int log_n, S, logB, logC, sub_log;
  int last;                 // M == R: no inter-pass twiddle
  int h;
  const uint32_t* tw_sub;   // w_Rsub^k (Rsub = 2^sub_log), k < Rsub/2
  const uint32_t* tw_lo;    // two-level w_N
  const uint32_t* tw_hi;
  const uint32_t* pre_lo;   // optional: multiply input j by pre_hi[j>>h]*pre_lo[j&mask]
  const uint32_t* pre_hi;
  const uint32_t* post_lo;  // optional: multiply output k likewise
  const uint32_t* post_hi;
  const uint32_t* post_const;  // optional: multiply every output by a constant (1/N)
  const uint32_t* tw_full;     // optional full tables (indexed by output / input / output position): replace the
  const uint32_t* pre_full;    // two-level lookups above by one load + one product
  const uint32_t* post_full;
  const uint32_t* in[NTT_MAX_BATCH];
  uint32_t* out[NTT_MAX_BATCH];
  // optional (first pass of the witness map's coset_ifft): the input element is (in * fuse_b - fuse_c) * fuse_k, i.e. the
  // pointwise step (a*b - c) / Z(g) of r1cs_to_qap.rs:150,164-168 fused into the load
  const uint32_t* fuse_b;
  const uint32_t* fuse_c;
  const uint32_t* fuse_k;
};

// full tables: out[i] = hi[e >> h] * lo[e & mask] with e = i (coset factors) or the inter-pass exponent of output i
template <class P>
__global__ __launch_bounds__(256) void ntt_full_table_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ lo,
                                                             const uint32_t* __restrict__ hi, int h, uint32_t n,
                                                             int S, int logB, int twiddle, int rp) {
  using F = Fp<P>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t e = i;
  if (twiddle) {
    uint32_t k1 = (i >> logB) & ((1u << S) - 1);
    uint32_t bj = (i >> (logB + S)) << logB;
    e = bj * k1;
  }
  F r = F::load(hi + (size_t)(e >> h) * 8) * F::load(lo + (size_t)(e & ((1u << h) - 1)) * 8);
  // rp: the entry as a canonical R'-form factor (t * 2^(L*B) mod p instead of t * 2^(32N)): SHIFT modular doublings.  A pass of
  // the unsaturated kernel below repacks it into limbs and has a factor < p, not < 2^SHIFT * p
  if (rp)
    for (int k = 0; k < Fu<P>::SHIFT; k++) r = r + r;
  r.store(out + (size_t)i * 8);
}

// Round 3: the products of a pass run on the UNSATURATED multiplier (unsat_dev.hpp: 9 x 29-bit limbs, one v_mad_u64_u32 per
// partial product, 205 instead of ~330 VALU instructions) while the data stay saturated (canonical Montgomery words) in LDS and
// HBM, so additions / subtractions keep their carry-chain forms: x enters by a plain limb repack (value < p), the table factor t
// (a twiddle / coset factor, canonical t * 2^256) as from_sat(t) = t * R' (< 32p: 1 * 32 <= floor(R' / r), also for the 255-bit
// BLS12-381 scalar field), the product x t * 2^256 < 2p leaves by a repack and one conditional subtraction.  The butterfly
// twiddle tile is converted once per workgroup (9 words per twiddle in LDS).
template <class P>
ZKP_DEV Fp<P> ntt_mul(const Fp<P>& x, const Fu<P>& t) {
  Fu<P> r = Fu<P>::mul(Fu<P>::from_words(x.v, 0), t);
  Fp<P> o;
  r.to_words(o.v);
  return Fp<P>::reduce_once(o);
}
template <class P>
ZKP_DEV Fu<P> ntt_factor(const uint32_t* p) { return Fu<P>::from_sat(Fp<P>::load(p)); }

// 129 VGPRs -> 3 workgroups per CU.  Forcing 128 (launch bounds (256, 4): 3 spilled registers, 4 workgroups per CU) was
// measured slower: 0.181 vs 0.174 ms per 2^20 transform, 110-112 vs 113-115 proofs/s.
template <class P>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassArgs a) {
  using F = Fp<P>;
  using U = Fu<P>;
  constexpr int UL = U::L;
  // (s_setprio for these waves — the witness map gates the H MSM — was measured in round 3: 137.1 vs 137.0 proofs/s, not kept;
  //  the same hint on the bucket-sort kernels gained 1.4 %, msm.hip ZKP_SORT_PRIO)
  const uint32_t* __restrict__ in = a.in[blockIdx.y];
  uint32_t* __restrict__ out = a.out[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int S = a.S, logC = a.logC;
  const int R = 1 << S, C = 1 << logC;
  const int TILE = R << logC;
  uint32_t* tw_l = lds + 8 * TILE;                 // R/2 twiddles as product factors (from_sat: UL words each), AoS (broadcast reads)
  const uint32_t ncols = 1u << (a.log_n - S);
  const uint32_t c0 = blockIdx.x << logC;
  const int tid = threadIdx.x;
  const uint32_t hmask = (1u << a.h) - 1;

  // sub-FFT twiddle tile -> LDS
  {
    const int shift = a.sub_log - S;               // w_R^k = w_Rsub^(k << shift)
    for (int e = tid; e < (R >> 1); e += NTT_THREADS) {
      const U t = ntt_factor<P>(a.tw_sub + ((size_t)(e << shift)) * 8);
#pragma unroll
      for (int l = 0; l < UL; l++) tw_l[e * UL + l] = t.v[l];
    }
  }
  // load tile (coalesced runs of C elements), optional coset pre-scale
  for (int e = tid; e < TILE; e += NTT_THREADS) {
    uint32_t j1 = e >> logC, cc = e & (C - 1);
    uint32_t gidx = j1 * ncols + c0 + cc;
    F x = F::load(in + (size_t)gidx * 8);
    if (a.fuse_b)
      x = ntt_mul(ntt_mul(x, ntt_factor<P>(a.fuse_b + (size_t)gidx * 8)) - F::load(a.fuse_c + (size_t)gidx * 8), ntt_factor<P>(a.fuse_k));
    if (a.pre_full) {
      x = ntt_mul(x, ntt_factor<P>(a.pre_full + (size_t)gidx * 8));
    } else if (a.pre_lo) {
      F s = ntt_mul(F::load(a.pre_hi + (size_t)(gidx >> a.h) * 8), ntt_factor<P>(a.pre_lo + (size_t)(gidx & hmask) * 8));
      x = ntt_mul(x, U::from_sat(s));
    }
#pragma unroll
    for (int l = 0; l < 8; l++) lds[l * TILE + e] = x.v[l];
  }
  // S radix-2 DIF stages over rows, two stages per LDS round trip: a lane holds the four rows
  // {base, base + q, base + 2q, base + 3q} (q = quarter of the current group) and runs both butterfly layers in registers
  // — half the LDS traffic and barriers of one stage per round trip; an odd S ends with a single radix-2 stage.
  int s = 0;
  for (; s + 1 < S; s += 2) {
    __syncthreads();
    const int half = R >> (s + 1), quarter = half >> 1;
    for (int qd = tid; qd < (TILE >> 2); qd += NTT_THREADS) {
      int cc = qd & (C - 1);
      int u = qd >> logC;
      int pos = u & (quarter - 1);
      int grp = u >> (S - 2 - s);                          // u / quarter
      int p0 = (((grp * 2 * half) + pos) << logC) + cc;
      int p1 = p0 + (quarter << logC), p2 = p0 + (half << logC), p3 = p2 + (quarter << logC);
      F x0, x1, x2, x3;
#pragma unroll
      for (int l = 0; l < 8; l++) {
        x0.v[l] = lds[l * TILE + p0];
        x1.v[l] = lds[l * TILE + p1];
        x2.v[l] = lds[l * TILE + p2];
        x3.v[l] = lds[l * TILE + p3];
      }
      U ta, tb;
#pragma unroll
      for (int l = 0; l < UL; l++) {
        ta.v[l] = tw_l[(pos << s) * UL + l];
        tb.v[l] = tw_l[((pos + quarter) << s) * UL + l];
      }
      F y0 = x0 + x2, y2 = ntt_mul(x0 - x2, ta);
      F y1 = x1 + x3, y3 = ntt_mul(x1 - x3, tb);
      F z0 = y0 + y1, z1 = y0 - y1, z2 = y2 + y3, z3 = y2 - y3;
      if (s + 2 < S) {                                     // the last stage's twiddle is w^0
        U tc;
#pragma unroll
        for (int l = 0; l < UL; l++) tc.v[l] = tw_l[(pos << (s + 1)) * UL + l];
        z1 = ntt_mul(z1, tc);
        z3 = ntt_mul(z3, tc);
      }
#pragma unroll
      for (int l = 0; l < 8; l++) {
        lds[l * TILE + p0] = z0.v[l];
        lds[l * TILE + p1] = z1.v[l];
        lds[l * TILE + p2] = z2.v[l];
        lds[l * TILE + p3] = z3.v[l];
      }
    }
  }
  for (; s < S; s++) {
    __syncthreads();
    const int half = R >> (s + 1);
    for (int bf = tid; bf < (TILE >> 1); bf += NTT_THREADS) {
      int cc = bf & (C - 1);
      int u = bf >> logC;
      int pos = u & (half - 1);
      int grp = u >> (S - 1 - s);
      int p0 = (((grp * 2 * half) + pos) << logC) + cc;
      int p1 = p0 + (half << logC);
      F x0, x1;
#pragma unroll
      for (int l = 0; l < 8; l++) {
        x0.v[l] = lds[l * TILE + p0];
        x1.v[l] = lds[l * TILE + p1];
      }
      F y0 = x0 + x1;
      F y1 = x0 - x1;
      if (s + 1 < S) {                                   // the last stage's twiddle is w^0
        U tw;
#pragma unroll
        for (int l = 0; l < UL; l++) tw.v[l] = tw_l[(pos << s) * UL + l];
        y1 = ntt_mul(y1, tw);
      }
#pragma unroll
      for (int l = 0; l < 8; l++) {
        lds[l * TILE + p0] = y0.v[l];
        lds[l * TILE + p1] = y1.v[l];
      }
    }
  }
  __syncthreads();
  // store: rows come out bit-reversed in LDS; enumerate outputs in global-address order
  const int logB = a.logB;
  const int logBc = logB < logC ? logB : logC;
  for (int o = tid; o < TILE; o += NTT_THREADS) {
    uint32_t bb = o & ((1u << logBc) - 1);
    uint32_t k1 = (o >> logBc) & (R - 1);
    uint32_t jj = o >> (logBc + S);
    uint32_t cc = bb + (jj << logBc);
    uint32_t c = c0 + cc;
    uint32_t b = c & ((1u << logB) - 1);
    uint32_t bj = c - b;                              // B * j'
    uint32_t oidx = b + (k1 << logB) + (bj << S);
    uint32_t row = S ? (__brev(k1) >> (32 - S)) : 0;
    int pos = (row << logC) + cc;
    F x;
#pragma unroll
    for (int l = 0; l < 8; l++) x.v[l] = lds[l * TILE + pos];
    if (!a.last) {
      if (a.tw_full) {
        x = ntt_mul(x, ntt_factor<P>(a.tw_full + (size_t)oidx * 8));
      } else {
        uint32_t e = bj * k1;                           // < N
        if (e) {
          F t = ntt_mul(F::load(a.tw_hi + (size_t)(e >> a.h) * 8), ntt_factor<P>(a.tw_lo + (size_t)(e & hmask) * 8));
          x = ntt_mul(x, U::from_sat(t));
        }
      }
    }
    if (a.post_full) {
      x = ntt_mul(x, ntt_factor<P>(a.post_full + (size_t)oidx * 8));
    } else if (a.post_lo) {
      F t = ntt_mul(F::load(a.post_hi + (size_t)(oidx >> a.h) * 8), ntt_factor<P>(a.post_lo + (size_t)(oidx & hmask) * 8));
      x = ntt_mul(x, U::from_sat(t));
    } else if (a.post_const) {
      x = ntt_mul(x, ntt_factor<P>(a.post_const));
    }
    x.store(out + (size_t)oidx * 8);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3, second half: the pass with the tile UNSATURATED in LDS (ZKP_NTT_V2=0 restores the kernel above).
//
// The kernel above converts at every product (repack in, repack + conditional subtraction out: ~80 of the ~285 VALU
// instructions of a butterfly's product) and adds / subtracts on carry chains with a conditional subtraction.  Here an element
// enters the tile once as 9 x 29-bit limbs and stays that way through all S stages; the in-tile transform is decimation in TIME
// (rows placed bit-reversed at the load, natural order at the store):   a' = a + w b,  b' = a + 2p - w b   with w b < 2p from the
// multiplier, so the value bound grows by 2p per stage — LINEARLY (<= (2S + 1) p = 15p after seven stages; decimation in
// frequency doubles the bound of the never-multiplied sum path per stage) — and no stage reduces anything: a butterfly is one
// product (205 instructions) + a normalising add + a normalising subtract (27 each).  Twiddles of the tile are brought below 2p
// once per workgroup (data bound x twiddle bound = 30 <= floor(R' / r) = 64 for the 255-bit BLS12-381 scalar field too); the
// per-element tables (inter-pass twiddles, coset factors) are generated in canonical R' form (ntt_full_table_kernel, rp), so
// the product that applies them also brings the element back below 2p, and one repack + conditional subtraction per element
// and pass returns the canonical words HBM holds between passes.  DIT also makes the trivial twiddles fall in the FIRST stages:
// 2.75 products per element in a 7-stage tile instead of 3.
// LDS: limb planes of PLANE = TILE + TILE/8 words (8 pad words per 64: the four rows a lane touches in the early stages are
// 4h rows apart, which would otherwise land in the same banks).
// (an XOR swizzle without padding — 39 KB, four workgroups per CU instead of three — measured the same: 142.7 vs 142.9 proofs/s)
// Round 4: no padding but an XOR swizzle of the bank bits — position bits [4:3] ^= bits [6:5] ^ bits [9:8] — which separates the
// same two conflict patterns (the rows a lane group touches in the first stage pair are 4 rows apart = bits [6:5] with 8 columns;
// the bit-reversed rows of the load phase differ in bits [9:8]) at 36.9 + 2.3 KB per workgroup: FOUR workgroups per CU instead of
// three, i.e. the 1024 workgroups of a 2^20 pass are all resident at once instead of 768 + a tail of 256.
// -DZKP_NTT_PAD restores the padded layout.
#ifdef ZKP_NTT_PAD
#define NTT2_PLANE(tile) ((tile) + ((tile) >> 3))
ZKP_DEV int ntt2_pad(int a) { return a + ((a >> 6) << 3); }
#else
#define NTT2_PLANE(tile) (tile)
ZKP_DEV int ntt2_pad(int a) { return a ^ ((((a >> 5) ^ (a >> 8)) & 3) << 3); }
#endif
template <class P>
ZKP_DEV Fu<P> ntt2_lds_load(const uint32_t* lds, int plane, int pos) {
  Fu<P> r;
#pragma unroll
  for (int l = 0; l < Fu<P>::L; l++) r.v[l] = lds[l * plane + pos];
  return r;
}
template <class P>
ZKP_DEV void ntt2_lds_store(uint32_t* lds, int plane, int pos, const Fu<P>& x) {
#pragma unroll
  for (int l = 0; l < Fu<P>::L; l++) lds[l * plane + pos] = x.v[l];
}
template <class P>
ZKP_DEV Fu<P> ntt2_tw(const uint32_t* tw_l, int e) {
  Fu<P> r;
#pragma unroll
  for (int l = 0; l < Fu<P>::L; l++) r.v[l] = tw_l[e * Fu<P>::L + l];
  return r;
}
// a table entry in canonical R' form (< p)
template <class P>
ZKP_DEV Fu<P> ntt2_factor(const uint32_t* p) {
  const Fp<P> t = Fp<P>::load(p);
  return Fu<P>::from_words(t.v, 0);
}

// TILE_C: the tile size as a compile-time constant (every domain >= 2^NTT_TILE_LOG has full tiles) so that the limb-plane offsets
// l * PLANE fold into the ds_read / ds_write offset fields instead of one v_add per limb and access; 0 = read it from the arguments
template <class P, int TILE_C>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass2_kernel(NttPassArgs a) {
  using F = Fp<P>;
  using U = Fu<P>;
  constexpr int UL = U::L;
  const uint32_t* __restrict__ in = a.in[blockIdx.y];
  uint32_t* __restrict__ out = a.out[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int S = a.S, logC = a.logC;
  const int R = 1 << S, C = 1 << logC;
  const int TILE = TILE_C ? TILE_C : (R << logC);
  const int PLANE = NTT2_PLANE(TILE);
  uint32_t* tw_l = lds + UL * PLANE;               // R/2 twiddles, < 2p, AoS (broadcast reads)
  const uint32_t ncols = 1u << (a.log_n - S);
  const uint32_t c0 = blockIdx.x << logC;
  const int tid = threadIdx.x;
  const uint32_t hmask = (1u << a.h) - 1;
  {
    const int shift = a.sub_log - S;               // w_R^k = w_Rsub^(k << shift)
    for (int e = tid; e < (R >> 1); e += NTT_THREADS) {
      const U t = U::from_sat_reduced(F::load(a.tw_sub + ((size_t)(e << shift)) * 8));
#pragma unroll
      for (int l = 0; l < UL; l++) tw_l[e * UL + l] = t.v[l];
    }
  }
  // load (coalesced runs of C elements); the optional pointwise step / coset pre-scale works on canonical values as above
  for (int e = tid; e < TILE; e += NTT_THREADS) {
    uint32_t j1 = e >> logC, cc = e & (C - 1);
    uint32_t gidx = j1 * ncols + c0 + cc;
    F x = F::load(in + (size_t)gidx * 8);
    if (a.fuse_b)
      x = ntt_mul(ntt_mul(x, ntt_factor<P>(a.fuse_b + (size_t)gidx * 8)) - F::load(a.fuse_c + (size_t)gidx * 8), ntt_factor<P>(a.fuse_k));
    if (a.pre_full) {
      x = ntt_mul(x, ntt2_factor<P>(a.pre_full + (size_t)gidx * 8));
    } else if (a.pre_lo) {
      F sc = ntt_mul(F::load(a.pre_hi + (size_t)(gidx >> a.h) * 8), ntt_factor<P>(a.pre_lo + (size_t)(gidx & hmask) * 8));
      x = ntt_mul(x, U::from_sat(sc));
    }
    const uint32_t row = S ? (__brev(j1) >> (32 - S)) : 0;
    ntt2_lds_store<P>(lds, PLANE, ntt2_pad((int)((row << logC) + cc)), U::from_words(x.v, 0));
  }
  // S radix-2 DIT stages over rows, two per LDS round trip: a lane holds rows {b, b + h, b + 2h, b + 3h} (h = 2^s)
  int s = 0;
  for (; s + 1 < S; s += 2) {
    __syncthreads();
    const int h = 1 << s;
    for (int qd = tid; qd < (TILE >> 2); qd += NTT_THREADS) {
      const int cc = qd & (C - 1);
      const int u = qd >> logC;
      const int pos = u & (h - 1);
      const int b = ((u >> s) << (s + 2)) + pos;
      const int p0 = ntt2_pad((b << logC) + cc), p1 = ntt2_pad(((b + h) << logC) + cc);
      const int p2 = ntt2_pad(((b + 2 * h) << logC) + cc), p3 = ntt2_pad(((b + 3 * h) << logC) + cc);
      U r0 = ntt2_lds_load<P>(lds, PLANE, p0), r1 = ntt2_lds_load<P>(lds, PLANE, p1);
      U r2 = ntt2_lds_load<P>(lds, PLANE, p2), r3 = ntt2_lds_load<P>(lds, PLANE, p3);
      if (s > 0) {                                         // stage s: w_{2h}^pos for both pairs (w^0 in the first stage)
        const U ta = ntt2_tw<P>(tw_l, pos << (S - 1 - s));
        r1 = U::mul(r1, ta);
        r3 = U::mul(r3, ta);
      }
      const U y0 = U::add(r0, r1), y1 = U::template sub<2>(r0, r1);
      U y2 = U::add(r2, r3), y3 = U::template sub<2>(r2, r3);
      if (s > 0) y2 = U::mul(y2, ntt2_tw<P>(tw_l, pos << (S - 2 - s)));           // stage s + 1: w_{4h}^pos ...
      y3 = U::mul(y3, ntt2_tw<P>(tw_l, (pos + h) << (S - 2 - s)));                // ... and w_{4h}^(pos + h)
      ntt2_lds_store<P>(lds, PLANE, p0, U::add(y0, y2));
      ntt2_lds_store<P>(lds, PLANE, p2, U::template sub<2>(y0, y2));
      ntt2_lds_store<P>(lds, PLANE, p1, U::add(y1, y3));
      ntt2_lds_store<P>(lds, PLANE, p3, U::template sub<2>(y1, y3));
    }
  }
  for (; s < S; s++) {
    __syncthreads();
    const int h = 1 << s;
    for (int bf = tid; bf < (TILE >> 1); bf += NTT_THREADS) {
      const int cc = bf & (C - 1);
      const int u = bf >> logC;
      const int pos = u & (h - 1);
      const int b = ((u >> s) << (s + 1)) + pos;
      const int p0 = ntt2_pad((b << logC) + cc), p1 = ntt2_pad(((b + h) << logC) + cc);
      const U r0 = ntt2_lds_load<P>(lds, PLANE, p0);
      U r1 = ntt2_lds_load<P>(lds, PLANE, p1);
      if (s > 0) r1 = U::mul(r1, ntt2_tw<P>(tw_l, pos << (S - 1 - s)));
      ntt2_lds_store<P>(lds, PLANE, p0, U::add(r0, r1));
      ntt2_lds_store<P>(lds, PLANE, p1, U::template sub<2>(r0, r1));
    }
  }
  __syncthreads();
  // store: rows are in natural order; enumerate outputs in global-address order.  Values < (2S + 1) p leave through a product
  // (inter-pass twiddle / coset factor / 1/N, or R' mod p when the pass has none), a repack and one conditional subtraction
  const int logB = a.logB;
  const int logBc = logB < logC ? logB : logC;
  U pc = U::zero();
  if (a.post_const) pc = U::from_sat_reduced(F::load(a.post_const));
  for (int o = tid; o < TILE; o += NTT_THREADS) {
    uint32_t bb = o & ((1u << logBc) - 1);
    uint32_t k1 = (o >> logBc) & (R - 1);
    uint32_t jj = o >> (logBc + S);
    uint32_t cc = bb + (jj << logBc);
    uint32_t c = c0 + cc;
    uint32_t b = c & ((1u << logB) - 1);
    uint32_t bj = c - b;                              // B * j'
    uint32_t oidx = b + (k1 << logB) + (bj << S);
    U x = ntt2_lds_load<P>(lds, PLANE, ntt2_pad((int)((k1 << logC) + cc)));
    bool reduced = false;
    if (!a.last) {
      if (a.tw_full) {
        x = U::mul(x, ntt2_factor<P>(a.tw_full + (size_t)oidx * 8));
        reduced = true;
      } else {
        uint32_t e = bj * k1;                           // < N
        if (e) {
          F t = ntt_mul(F::load(a.tw_hi + (size_t)(e >> a.h) * 8), ntt_factor<P>(a.tw_lo + (size_t)(e & hmask) * 8));
          x = U::mul(x, U::from_sat_reduced(t));
          reduced = true;
        }
      }
    }
    if (a.post_full) {
      x = U::mul(x, ntt2_factor<P>(a.post_full + (size_t)oidx * 8));
      reduced = true;
    } else if (a.post_lo) {
      F t = ntt_mul(F::load(a.post_hi + (size_t)(oidx >> a.h) * 8), ntt_factor<P>(a.post_lo + (size_t)(oidx & hmask) * 8));
      x = U::mul(x, U::from_sat_reduced(t));
      reduced = true;
    } else if (a.post_const) {
      x = U::mul(x, pc);
      reduced = true;
    }
    if (!reduced) x = U::mul(x, U::one());
    F r;
    x.to_words(r.v);
    F::reduce_once(r).store(out + (size_t)oidx * 8);
  }
}
static bool ntt_v2() {
  static const bool on = !(getenv("ZKP_NTT_V2") && atoi(getenv("ZKP_NTT_V2")) == 0);
  return on;
}
// one pass: the kernel and the LDS size that goes with it
template <class P>
static void ntt_launch(hipStream_t st, const NttPassArgs& a, uint32_t grid, int count) {
  const size_t tile = (size_t)1 << (a.S + a.logC);
  const size_t tw_bytes = ((size_t)1 << a.S) / 2 * 4 * Fu<P>::L + 32;
  if (ntt_v2() && tile == ((size_t)1 << NTT_TILE_LOG))
    hipLaunchKernelGGL((ntt_pass2_kernel<P, (1 << NTT_TILE_LOG)>), dim3(grid, count), dim3(NTT_THREADS),
                       NTT2_PLANE(tile) * 4 * Fu<P>::L + tw_bytes, st, a);
  else if (ntt_v2())
    hipLaunchKernelGGL((ntt_pass2_kernel<P, 0>), dim3(grid, count), dim3(NTT_THREADS), NTT2_PLANE(tile) * 4 * Fu<P>::L + tw_bytes, st, a);
  else
    hipLaunchKernelGGL(ntt_pass_kernel<P>, dim3(grid, count), dim3(NTT_THREADS), tile * 32 + tw_bytes, st, a);
}

static void ntt_plan(int log_n, int* S, int* P) {
  if (log_n == 0) {
    *P = 0;
    return;
  }
  const int smax = ntt_smax();
  int p = (log_n + smax - 1) / smax;
  int base = log_n / p, rem = log_n % p;
  for (int i = 0; i < p; i++) S[i] = base + (i < rem ? 1 : 0);
  *P = p;
}

// lazily built full table (one-time per (curve, log_n, kind)); returns nullptr above NTT_FULL_MAX_LOG
template <class P>
static uint32_t* full_table(zkp_ctx* ctx, NttTables& t, uint32_t** slot, const uint32_t* lo, const uint32_t* hi, int S,
                            int logB, int twiddle) {
  static const bool enabled = !(getenv("ZKP_NTT_FULL") && atoi(getenv("ZKP_NTT_FULL")) == 0);   // A/B switch
  if (!enabled || t.log_n > NTT_FULL_MAX_LOG || t.log_n < 2) return nullptr;
  if (*slot) return *slot;
  const size_t N = (size_t)1 << t.log_n;
  uint32_t* blk;
  if (hipMalloc(&blk, N * 32) != hipSuccess) return nullptr;        // fall back to the two-level lookup
  hipStream_t s = ctx->cur->stream;
  hipLaunchKernelGGL(ntt_full_table_kernel<P>, dim3((N + 255) / 256), dim3(256), 0, s, blk, lo, hi, t.h, (uint32_t)N, S,
                     logB, twiddle, ntt_v2() ? 1 : 0);
  ZKP_HIP(hipGetLastError());
  ZKP_HIP(hipStreamSynchronize(s));       // shared by every lane / stream afterwards
  t.extra.push_back(blk);
  *slot = blk;
  return blk;
}

// data[k]: N elements each (device), k < count <= NTT_MAX_BATCH independent transforms done by the same launches.
// In place from the caller's view; uses ctx->cur->ntt_scratch.
template <class P>
void ntt_run_t(zkp_ctx* ctx, int curve, uint32_t* const* data, int count, int log_n, int op) {
  ZKP_REQUIRE(log_n <= P::TWO_ADICITY, ZKP_ERR_DOMAIN_TOO_LARGE);
  ZKP_REQUIRE(count >= 1 && count <= NTT_MAX_BATCH, ZKP_ERR_BAD_ARG);
  if (log_n == 0) return;  // size-1 transform is the identity (coset scale by g^0, 1/N = 1)
  NttTables& t = get_tables<P>(ctx, curve, log_n);
  const size_t N = (size_t)1 << log_n;
  int S[8], np;
  ntt_plan(log_n, S, &np);
  // buffers: an even pass count ping-pongs data <-> scratch and ends in `data`; an odd one would end in scratch and pay a copy
  // (13 us of a 164 us transform at 2^20: three passes) — with a second scratch region its passes go data -> s0 -> s1 -> ... ->
  // data instead (round 4)
  const bool odd = (np & 1) && np > 1;
  uint32_t* scratch = ctx->cur->ntt_scratch.as<uint32_t>((size_t)count * N * 8 * (odd ? 2 : 1));
  uint32_t* scratch2 = scratch + (size_t)count * N * 8;
  const bool inverse = (op == ZKP_NTT_IFFT || op == ZKP_NTT_COSET_IFFT);
  bool in_scratch = false;
  int logB = 0;
  const uint32_t* src_base = nullptr;                  // nullptr = the caller's vectors
  for (int p = 0; p < np; p++) {
    // destination of pass p: np == 1: scratch (then copied); even np: alternate so that the last lands in data; odd np >= 3:
    // s0, s1, s0, ..., data (the last two passes use different scratch halves by construction: np - 2 is odd)
    uint32_t* dst_base;
    if (np == 1) dst_base = scratch;
    else if (p == np - 1) dst_base = nullptr;
    else if (!odd) dst_base = ((np - 1 - p) & 1) ? scratch : nullptr;
    else dst_base = (p & 1) ? scratch2 : scratch;
    NttPassArgs a{};
    for (int k = 0; k < count; k++) {
      a.in[k] = src_base ? src_base + (size_t)k * N * 8 : data[k];
      a.out[k] = dst_base ? dst_base + (size_t)k * N * 8 : data[k];
    }
    in_scratch = dst_base != nullptr;
    a.log_n = log_n;
    a.S = S[p];
    a.logB = logB;
    int logC = std::min(NTT_TILE_LOG - S[p], log_n - S[p]);
    a.logC = logC;
    a.sub_log = std::min(log_n, NTT_SUB_LOG);
    a.last = (p == np - 1);
    a.h = t.h;
    a.tw_sub = inverse ? t.sub_inv : t.sub_fwd;
    a.tw_lo = inverse ? t.wi_lo : t.w_lo;
    a.tw_hi = inverse ? t.wi_hi : t.w_hi;
    if (!a.last)
      a.tw_full = full_table<P>(ctx, t, inverse ? &t.full_inv[p] : &t.full_fwd[p], a.tw_lo, a.tw_hi, S[p], logB, 1);
    if (p == 0 && op == ZKP_NTT_COSET_FFT) {
      a.pre_lo = t.g_lo;
      a.pre_hi = t.g_hi;
      a.pre_full = full_table<P>(ctx, t, &t.full_g, t.g_lo, t.g_hi, 0, 0, 0);
    }
    if (p == np - 1) {
      if (op == ZKP_NTT_COSET_IFFT) {
        a.post_lo = t.gi_lo;
        a.post_hi = t.gi_hi;                          // includes 1/N
        a.post_full = full_table<P>(ctx, t, &t.full_gi, t.gi_lo, t.gi_hi, 0, 0, 0);
      } else if (op == ZKP_NTT_IFFT) {
        a.post_const = t.n_inv;
      }
    }
    ntt_launch<P>(ctx->cur->stream, a, (uint32_t)(N >> (S[p] + logC)), count);
    src_base = dst_base;
    logB += S[p];
  }
  ZKP_HIP(hipGetLastError());
  if (in_scratch)
    for (int k = 0; k < count; k++)
      ZKP_HIP(hipMemcpyAsync(data[k], scratch + (size_t)k * N * 8, N * 32, hipMemcpyDeviceToDevice, ctx->cur->stream));
}

// One pass launch shared by the fused runners below.
template <class P>
static void launch_pass(zkp_ctx* ctx, NttTables& t, int log_n, const int* S, int np, int p, int logB, bool inverse,
                        const uint32_t* in, uint32_t* out, NttPassArgs a) {
  a.in[0] = in;
  a.out[0] = out;
  a.log_n = log_n;
  a.S = S[p];
  a.logB = logB;
  const int logC = std::min(NTT_TILE_LOG - S[p], log_n - S[p]);
  a.logC = logC;
  a.sub_log = std::min(log_n, NTT_SUB_LOG);
  a.last = (p == np - 1);
  a.h = t.h;
  a.tw_sub = inverse ? t.sub_inv : t.sub_fwd;
  a.tw_lo = inverse ? t.wi_lo : t.w_lo;
  a.tw_hi = inverse ? t.wi_hi : t.w_hi;
  if (!a.last) a.tw_full = full_table<P>(ctx, t, inverse ? &t.full_inv[p] : &t.full_fwd[p], a.tw_lo, a.tw_hi, S[p], logB, 1);
  ntt_launch<P>(ctx->cur->stream, a, (uint32_t)(((size_t)1 << log_n) >> (S[p] + logC)), 1);
}

// Witness map, first half (r1cs_to_qap.rs:144-148,161-162): ifft_in_place followed by coset_fft_in_place of the same vector
// as ONE chain of 2*np passes: the inverse transform's 1/N is folded into the coset pre-scale table (g^j / N), the buffers
// ping-pong data <-> scratch so that the result lands in `data` without the copy an odd pass count costs twice otherwise.
template <class P>
static bool ntt_ifft_coset_fft_t(zkp_ctx* ctx, int curve, uint32_t* data, int log_n) {
  if (log_n < 2 || log_n > NTT_FULL_MAX_LOG) return false;
  NttTables& t = get_tables<P>(ctx, curve, log_n);
  uint32_t* pre = full_table<P>(ctx, t, &t.full_g_n, t.g_lo, t.g_hi_n, 0, 0, 0);
  if (!pre) return false;
  const size_t N = (size_t)1 << log_n;
  uint32_t* scratch = ctx->cur->ntt_scratch.as<uint32_t>(N * 8);
  int S[8], np;
  ntt_plan(log_n, S, &np);
  uint32_t* buf[2] = {data, scratch};
  int cur = 0;
  for (int half = 0; half < 2; half++) {
    int logB = 0;
    for (int p = 0; p < np; p++) {
      NttPassArgs a{};
      if (half == 1 && p == 0) a.pre_full = pre;                 // coset shift and 1/N in one product
      launch_pass<P>(ctx, t, log_n, S, np, p, logB, half == 0, buf[cur], buf[cur ^ 1], a);
      cur ^= 1;
      logB += S[p];
    }
  }
  ZKP_HIP(hipGetLastError());                                    // 2*np passes: back in `data`
  return true;
}

// Witness map, second half (r1cs_to_qap.rs:150,164-169): h = coset_ifft((a*b - c) / Z(g)) with the pointwise step fused
// into the first pass's load.  a, b, c: N elements each (destroyed).  Returns the buffer that holds h (a or b).
template <class P>
static uint32_t* ntt_qap_coset_ifft_t(zkp_ctx* ctx, int curve, uint32_t* a_, uint32_t* b_, uint32_t* c_, const uint32_t* zinv,
                                      int log_n) {
  if (log_n < 2 || log_n > NTT_FULL_MAX_LOG) return nullptr;
  NttTables& t = get_tables<P>(ctx, curve, log_n);
  uint32_t* post = full_table<P>(ctx, t, &t.full_gi, t.gi_lo, t.gi_hi, 0, 0, 0);
  if (!post) return nullptr;
  const size_t N = (size_t)1 << log_n;
  uint32_t* scratch = ctx->cur->ntt_scratch.as<uint32_t>(N * 8);
  int S[8], np;
  ntt_plan(log_n, S, &np);
  if (np < 2) return nullptr;
  // pass 0: (a, b, c) -> scratch; then scratch <-> a; an odd pass count ends in b (free once pass 0 has read it)
  const uint32_t* in = a_;
  uint32_t* out = scratch;
  int logB = 0;
  for (int p = 0; p < np; p++) {
    NttPassArgs a{};
    if (p == 0) {
      a.fuse_b = b_;
      a.fuse_c = c_;
      a.fuse_k = zinv;
    }
    if (p == np - 1) {
      a.post_lo = t.gi_lo;
      a.post_hi = t.gi_hi;
      a.post_full = post;
      if (out == scratch) out = b_;                              // odd np: do not finish in scratch
    }
    launch_pass<P>(ctx, t, log_n, S, np, p, logB, true, in, out, a);
    in = out;
    out = (out == scratch) ? a_ : scratch;
    logB += S[p];
  }
  ZKP_HIP(hipGetLastError());
  return const_cast<uint32_t*>(in);
}

bool ntt_ifft_coset_fft(zkp_ctx* ctx, int curve, uint32_t* data, int log_n) {
  static const bool on = !(getenv("ZKP_NTT_FUSE") && atoi(getenv("ZKP_NTT_FUSE")) == 0);
  if (!on) return false;
  if (curve == ZKP_BN254) return ntt_ifft_coset_fft_t<Bn254Fr>(ctx, curve, data, log_n);
  if (curve == ZKP_BLS12_381) return ntt_ifft_coset_fft_t<Bls381Fr>(ctx, curve, data, log_n);
  throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
}
uint32_t* ntt_qap_coset_ifft(zkp_ctx* ctx, int curve, uint32_t* a, uint32_t* b, uint32_t* c, const uint32_t* zinv, int log_n) {
  static const bool on = !(getenv("ZKP_NTT_FUSE") && atoi(getenv("ZKP_NTT_FUSE")) == 0);
  if (!on) return nullptr;
  if (curve == ZKP_BN254) return ntt_qap_coset_ifft_t<Bn254Fr>(ctx, curve, a, b, c, zinv, log_n);
  if (curve == ZKP_BLS12_381) return ntt_qap_coset_ifft_t<Bls381Fr>(ctx, curve, a, b, c, zinv, log_n);
  throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
}

void ntt_run_batch(zkp_ctx* ctx, int curve, uint32_t* const* data, int count, int log_n, int op) {
  ZKP_REQUIRE(op >= 0 && op <= 3, ZKP_ERR_BAD_ARG);
  if (curve == ZKP_BN254) ntt_run_t<Bn254Fr>(ctx, curve, data, count, log_n, op);
  else if (curve == ZKP_BLS12_381) ntt_run_t<Bls381Fr>(ctx, curve, data, count, log_n, op);
  else throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
}

void ntt_run(zkp_ctx* ctx, int curve, uint32_t* data, int log_n, int op) {
  uint32_t* one[1] = {data};
  ntt_run_batch(ctx, curve, one, 1, log_n, op);
}

void ntt_free_tables(zkp_ctx* ctx) {
  for (auto& kv : ctx->ntt_tables) {
    if (kv.second.block) (void)hipFree(kv.second.block);
    for (void* e : kv.second.extra) (void)hipFree(e);
  }
  ctx->ntt_tables.clear();
}

}  // namespace zkp
