// Multi-scalar multiplication over G1/G2 for gfx950: fixed-base-precomputed, single-window, sort-based
// Pippenger.
//
// Replaces ark-ec 0.2 `VariableBaseMSM::multi_scalar_mul(bases, scalars)` (reference call sites:
// /root/reference/groth16/src/prover.rs:187,190,220; /root/reference/marlin/src/pc/kzg10.rs:109,118,137,146;
// /root/reference/curve/src/lib.rs:44).  ark's CPU algorithm is W = ceil(bits/c) independent windows of
// 2^c-1 buckets each, one rayon task per window, then a Horner combine with c doublings per window.
// A GPU lane is ~1000x slower than the chip in aggregate, so every serial tail (running sums, the Horner
// chain of ~250 doublings) is poison here; and MI355X has 288 GB of HBM.  Design, MI355X-first:
//
//   upload (once per key):  T[w][i] = 2^(c*w) * P_i  for w < W, affine, resident in HBM (W ~ 13 copies).
//   per MSM:
//     K5  digit scan      scalar i -> W signed c-bit digits d_w in [-2^(c-1), 2^(c-1)];  entry
//                         (bucket |d_w|-1, point w*n+i, sign) — all windows share ONE bucket set because
//                         the window weight already lives in T.  Fused into both level-1 passes of the sort:
//                         coalesced 32 B/scalar reads, one 8 B store per entry; identity bases / zero digits
//                         emit nothing.
//     K6  group by bucket hand-written two-level counting sort (LDS histograms + LDS-atomic scatter); also yields
//                         the start/end of every bucket.
//     K7  accumulate      one lane per bucket walks its run of the sorted list: gather T[val] (64 B,
//                         negate y on sign) and mixed-add into an XYZZ accumulator held in VGPRs.
//     K8  reduce          sum_b (b+1) B_b  with log-depth kernels only: pairwise-sum pyramid A^(l+1)_k =
//                         A^l_2k + A^l_2k+1; O_l = sum of odd entries of level l;  result =
//                         sum_l 2^l O_l + root.  No running sums, no Horner over windows.
//
// The result is a group element: any correct schedule is bit-identical after `into_affine()`.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <memory>
#include <mutex>
#include <vector>

#include "field_dev.hpp"
#include "internal.hpp"
#include "msm_vtbl.hpp"

namespace zkp {

// memory- / latency-bound kernels that gate the next accumulate launch raise their wave priority (A/B: -DZKP_NO_SORT_PRIO)
#if defined(ZKP_NO_SORT_PRIO)
#define ZKP_SORT_PRIO() do { } while (0)
#else
#define ZKP_SORT_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

const MsmVtbl* msm_vtbl_c01();
const MsmVtbl* msm_vtbl_c02();
const MsmVtbl* msm_vtbl_c11();
const MsmVtbl* msm_vtbl_c12();

const MsmVtbl* msm_vtbl(int curve, int group) {
  if (curve == ZKP_BN254 && group == 1) return msm_vtbl_c01();
  if (curve == ZKP_BN254 && group == 2) return msm_vtbl_c02();
  if (curve == ZKP_BLS12_381 && group == 1) return msm_vtbl_c11();
  if (curve == ZKP_BLS12_381 && group == 2) return msm_vtbl_c12();
  throw StatusError{ZKP_ERR_UNSUPPORTED_CURVE};
}

struct BasesEntry {
  int curve = 0, group = 1;
  const MsmVtbl* vt = nullptr;
  size_t n = 0;
  int c = 0, W = 0;            // bucket bits (= width of the wide windows), number of windows
  int wide = 0;                // the first `wide` windows are c bits wide, the remaining W - wide are c - 1 (balanced windows)
  uint32_t cap = MSM_TASK_CAP; // entries per task (<= MSM_TASK_CAP): smaller = more lanes for MSMs with few, long buckets
  char* table = nullptr;       // W * n affine points: T[w][i] = 2^(c*w) * P_i (J copies with window groups, lgk below)
  size_t table_bytes = 0;
  uint8_t* inf = nullptr;      // n identity flags (device) or nullptr
  // Variable-base mode (zkp_msm_g*_var): no window tables — `table` holds the n points as uploaded, the digit scan emits
  // entry (bucket w * 2^(c-1) + |d| - 1, point i) into W SEPARATE bucket sets, and the reduction weights window w by 2^(c*w).
  bool var = false;
  // Window GROUPS (round 3: graceful path when the W window tables do not fit).  k = 2^lgk consecutive windows share ONE table copy:
  // window w = k*j + m uses T[j] = 2^(c*k*j) * P and bucket set m (weight 2^(c*m), applied by the reduction with c*m doublings), so
  // only J = ceil(W / k) copies are resident.  k = 1: the fixed-base plan above (var == false); k > 1: var == true with k bucket sets
  // of 2^(c-1) buckets; k >= W (J = 1): the variable-base plan.  Equal window widths (wide == W) whenever k > 1.
  int lgk = 0;
  bool owns = true;            // false: table / inf live in context scratch
  // Optional flags used ONLY by the digit scan when this entry owns a bucket sort that other MSMs reuse (Groth16: A, B1, B2
  // and L share one sort of z): a point is skipped by the scan only if it is the identity in EVERY sharing query; each
  // MSM's own identities are (0, 0) in its table and are skipped by its accumulate kernel.
  uint8_t* sort_inf = nullptr;
  // Shared level-1 pass (bases_set_group): on the OWNER of the pass, group_flags[i] bit k = base i is the identity in member k
  // of the group although the group's scan keeps it; the scatter stores the three bits in the top of each 8-byte entry.  On
  // every member, filter_bit = its k: its level-2 sort drops the entries whose bit k is set.
  uint8_t* group_flags = nullptr;
  int filter_bit = -1;
  ~BasesEntry() {
    if (owns && table) (void)hipFree(table);
    if (owns && inf) (void)hipFree(inf);
    if (sort_inf) (void)hipFree(sort_inf);
    if (group_flags) (void)hipFree(group_flags);
  }
};

static int pick_window_bits(const zkp_cfg& cfg, size_t n, int group, bool lone) {
  if (group == 2 && cfg.msm_c_g2 >= 2 && cfg.msm_c_g2 <= 22) return cfg.msm_c_g2;       // zkp_ctx_config.msm_window_bits_g2 / ZKP_MSM_C_G2
  if (cfg.msm_c >= 2 && cfg.msm_c <= 22) return cfg.msm_c;                                // zkp_ctx_config.msm_window_bits / ZKP_MSM_C
  int lg = 0;
  while (((size_t)2 << lg) <= n) lg++;          // floor(log2 n)
  if (lg < 63 && (double)n >= 1.41421356 * (double)((size_t)1 << lg)) lg++;   // round(log2 n): 2^20 - 1 -> 20
  // Round 6 (profiles/r06_msm_window_sweep.txt, r06_widen_ab.txt): chosen per call site.  A LONE MSM below 2^20 points (a base vector
  // uploaded through zkp_bases_upload_*: KZG10 powers, VariableBaseMSM callers) is latency-bound — ~20 dependent launches, 0.55-0.9 ms
  // whatever n is — and every window fewer is one table gather and one bucket addition per point less, while the reduction of 2^(c-1)
  // buckets hides in the same launches.  Measured optimum on resident tables: c = lg + 3 up to 2^14 (0.886 -> 0.573 ms at 2^12,
  // 0.951 -> 0.609 at 2^14), lg + 2 up to 2^17 (0.82 -> 0.705 at 2^16), lg + 1 at 2^18 / 2^19 (1.038 -> 0.957 at 2^18), lg from 2^20
  // on (c = 20 stays the optimum there and above: 19 and 21 both lose).  The queries of a Groth16 key keep round(log2 n): the
  // pipelined prover is VALU-bound, where twice the buckets is twice the reduction work (2^18 circuit: 495 -> 433 proofs/s widened).
  static const bool widen = !(getenv("ZKP_MSM_WIDEN") && atoi(getenv("ZKP_MSM_WIDEN")) == 0);      // A/B: 0 = round(log2 n) everywhere
  if (lone && widen && lg >= 10) lg += lg <= 14 ? 3 : lg <= 17 ? 2 : lg <= 19 ? 1 : 0;
  return std::min(20, std::max(4, lg));
}

// Window plan of a resident base vector at group size k = 2^lgk: c window bits, W windows, J resident table copies.
struct WindowPlan {
  int c, W, wide, lgk, J;
};
static WindowPlan window_plan(const zkp_cfg& cfg, int scalar_bits, size_t n, int group, int c_hint, int lgk) {
  static const bool balanced = !(getenv("ZKP_MSM_BALANCED") && atoi(getenv("ZKP_MSM_BALANCED")) == 0);
  const int T = scalar_bits + 1, c0 = c_hint >= 2 && c_hint <= 22 ? c_hint : pick_window_bits(cfg, n, group, /*lone=*/c_hint == -1);
  WindowPlan p{};
  if (lgk <= 0) {
    // Balanced windows: T = scalar_bits + 1 (one spare bit absorbs the last signed-digit carry) is spread over W = ceil(T / c)
    // windows of ceil(T / W) and floor(T / W) bits instead of W - 1 full windows and a thin top one.  With c = 20 on a 254-bit
    // field the top window was 14 bits wide: its n digits fell on 1/64 of the buckets (6x the average load: 16 of the 1024
    // level-1 sort bins oversized, three tasks per hot bucket and a combine step); now 8 windows are 20 and 5 are 19 bits wide.
    // ZKP_MSM_BALANCED=0 restores equal widths.
    p.W = (T + c0 - 1) / c0;
    p.c = balanced ? (T + p.W - 1) / p.W : c0;
    p.wide = balanced ? T - p.W * (p.c - 1) : p.W;
    p.lgk = 0;
    p.J = p.W;
    return p;
  }
  // window groups: k bucket sets keep the bucket count at 2^(c0 - 1) (c = c0 - lgk), equal widths, J = ceil(W / k) table copies
  p.c = std::max(4, c0 - lgk);
  p.W = (T + p.c - 1) / p.c;
  p.wide = p.W;
  p.lgk = lgk;
  if ((1 << lgk) >= p.W) {                                     // one copy: k = the next power of two >= W
    p.lgk = 0;
    while ((1 << p.lgk) < p.W) p.lgk++;
  }
  p.J = (p.W + (1 << p.lgk) - 1) >> p.lgk;
  return p;
}
size_t bases_table_bytes(zkp_ctx* ctx, int curve, int group, size_t n, int lgk) {
  const MsmVtbl* vt = msm_vtbl(curve, group);
  return std::max<size_t>(1, n) * (size_t)window_plan(ctx->cfg, vt->scalar_bits, n, group, 0, lgk).J * vt->aff_bytes;
}
// Budget for resident window tables: ZKP_TABLE_BUDGET_GB (whole context) if set, else the free device memory minus a quarter of
// the device for MSM / NTT scratch; what the context already holds is subtracted.
static double table_budget_left(zkp_ctx* ctx) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 1e30;
  double left = (double)free_b - 0.25 * (double)total_b;
  if (ctx->cfg.table_budget_gb > 0.0) {                          // zkp_ctx_config.table_budget_gb / ZKP_TABLE_BUDGET_GB
    const double b = ctx->cfg.table_budget_gb * 1073741824.0 - (double)ctx->table_bytes;
    left = std::min(left + 0.25 * (double)total_b - 1073741824.0, b);     // an explicit budget only keeps 1 GiB back
  }
  return left;
}
int bases_plan_lgk(zkp_ctx* ctx, int curve, const int* groups, const size_t* ns, int count) {
  if (const char* e = getenv("ZKP_TABLE_K")) {                   // forced group size (tests)
    int k = atoi(e), lg = 0;
    while ((1 << lg) < k) lg++;
    return std::min(lg, 6);
  }
  const double left = table_budget_left(ctx);
  for (int lgk = 0; lgk <= 6; lgk++) {
    double need = 0;
    for (int q = 0; q < count; q++) need += (double)bases_table_bytes(ctx, curve, groups[q], ns[q], lgk);
    if (need <= left) return lgk;
  }
  return 6;                                                      // one copy per query: let the allocation decide
}

uint64_t bases_upload(zkp_ctx* ctx, int curve, int group, const uint64_t* xy, const uint8_t* inf, size_t n, int c_hint,
                      int cap_hint, int lgk_hint) {
  auto e = std::make_shared<BasesEntry>();
  e->curve = curve;
  e->group = group;
  e->vt = msm_vtbl(curve, group);
  e->n = n;
  // Graceful path when the window tables do not fit (round 3): instead of ZKP_ERR_OOM the plan degrades to window groups —
  // k = 2, 4, 8, ... consecutive windows share one table copy and use k bucket sets (BasesEntry::lgk) — down to ONE copy, the
  // variable-base plan.  Cost at 2^20 points: tables 13 / 7 / 4 / 2 / 1 copies, entries x1.00 / 1.08 / 1.15 / 1.15 / 1.23, and a
  // reduction tail of (c k - 1) doublings.
  int lgk = lgk_hint;
  if (lgk < 0) lgk = c_hint > 0 ? 0 : bases_plan_lgk(ctx, curve, &group, &n, 1);
  const WindowPlan wp = window_plan(ctx->cfg, e->vt->scalar_bits, n, group, c_hint, lgk);
  e->c = wp.c;
  e->W = wp.W;
  e->wide = wp.wide;
  e->lgk = wp.lgk;
  e->var = wp.lgk > 0;
  {
    const char* env = group == 2 ? getenv("ZKP_TASK_CAP_G2") : nullptr;
    if (!env) env = getenv("ZKP_TASK_CAP");
    if (cap_hint >= 4) e->cap = std::min<uint32_t>(MSM_TASK_CAP, (uint32_t)cap_hint);
    if (env && atoi(env) >= 4) e->cap = std::min<uint32_t>(MSM_TASK_CAP, (uint32_t)atoi(env));
  }
  ZKP_REQUIRE((double)n * e->W < 2147483000.0, ZKP_ERR_BAD_ARG);
  const size_t ab = e->vt->aff_bytes;
  size_t bytes = std::max<size_t>(1, n) * wp.J * ab;
  if (hipMalloc(&e->table, bytes) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  e->table_bytes = bytes;
  ctx->table_bytes += bytes;
  if (n) {
    ZKP_HIP(hipMemcpyAsync(e->table, xy, n * ab, hipMemcpyHostToDevice, ctx->cur->stream));
    if (inf) {
      if (hipMalloc(&e->inf, n) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
      ZKP_HIP(hipMemcpyAsync(e->inf, inf, n, hipMemcpyHostToDevice, ctx->cur->stream));
      e->vt->ingest(ctx->cur->stream, e->table, e->inf, n);
    }
    // copy j = 2^(c k j) * P: the precompute kernel sees J "windows" of c*k bits (all of them full width) when k > 1
    if (wp.lgk > 0) e->vt->precompute(ctx->cur->stream, e->table, n, e->c << wp.lgk, wp.J, wp.J);
    else e->vt->precompute(ctx->cur->stream, e->table, n, e->c, e->W, e->wide);
    ZKP_HIP(hipGetLastError());
    ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  }
  if (getenv("ZKP_DEBUG_MSM"))
    fprintf(stderr, "[msm] bases group=%d n=%zu: c=%d W=%d wide=%d k=%d copies=%d (%.2f GiB)\n", group, n, e->c, e->W, e->wide,
            1 << e->lgk, wp.J, (double)bytes / 1073741824.0);
  uint64_t h = ctx->next_handle++;
  ctx->bases[h] = e;
  return h;
}

static std::shared_ptr<BasesEntry> get_bases(zkp_ctx* ctx, uint64_t handle) {
  auto it = ctx->bases.find(handle);
  if (it == ctx->bases.end()) throw StatusError{ZKP_ERR_BAD_HANDLE};
  return it->second;
}
void bases_free(zkp_ctx* ctx, uint64_t handle) {
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  auto it = ctx->bases.find(handle);
  if (it == ctx->bases.end()) throw StatusError{ZKP_ERR_BAD_HANDLE};
  if (it->second.use_count() == 1) ctx->table_bytes -= std::min(ctx->table_bytes, it->second->table_bytes);
  ctx->bases.erase(it);
}
size_t bases_len(zkp_ctx* ctx, uint64_t handle) { return get_bases(ctx, handle)->n; }
int bases_group(zkp_ctx* ctx, uint64_t handle) { return get_bases(ctx, handle)->group; }
void bases_info(zkp_ctx* ctx, uint64_t handle, uint64_t info[5]) {
  auto e = get_bases(ctx, handle);
  info[0] = (uint64_t)e->c;
  info[1] = (uint64_t)e->W;
  info[2] = (uint64_t)1 << e->lgk;
  info[3] = e->lgk ? (uint64_t)((e->W + (1 << e->lgk) - 1) >> e->lgk) : (uint64_t)e->W;
  info[4] = e->table_bytes;
}
void bases_drop(zkp_ctx* ctx, uint64_t handle) {              // like bases_free, for owners that tear a key down (no throw)
  auto it = ctx->bases.find(handle);
  if (it == ctx->bases.end()) return;
  if (it->second.use_count() == 1) ctx->table_bytes -= std::min(ctx->table_bytes, it->second->table_bytes);
  ctx->bases.erase(it);
}
void msm_free_all(zkp_ctx* ctx) {
  ctx->bases.clear();
  ctx->table_bytes = 0;
  ctx->var_plans.clear();
}
uint64_t bases_share(zkp_ctx* dst, zkp_ctx* src, uint64_t handle) {
  auto e = get_bases(src, handle);
  ZKP_REQUIRE(dst->device == src->device, ZKP_ERR_BAD_ARG);
  uint64_t h = dst->next_handle++;
  dst->bases[h] = e;
  return h;
}
void bases_set_sort_flags(zkp_ctx* ctx, uint64_t handle, const uint8_t* flags_host, size_t n) {
  auto e = get_bases(ctx, handle);
  ZKP_REQUIRE(n == e->n, ZKP_ERR_BAD_ARG);
  if (e->sort_inf) (void)hipFree(e->sort_inf);
  e->sort_inf = nullptr;
  if (!flags_host || n == 0) return;
  if (hipMalloc(&e->sort_inf, n) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  ZKP_HIP(hipMemcpyAsync(e->sort_inf, flags_host, n, hipMemcpyHostToDevice, ctx->cur->stream));
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
}
void bases_set_group(zkp_ctx* ctx, uint64_t owner, const uint8_t* member_flags_host, size_t n) {
  auto e = get_bases(ctx, owner);
  ZKP_REQUIRE(n == e->n, ZKP_ERR_BAD_ARG);
  if (e->group_flags) (void)hipFree(e->group_flags);
  e->group_flags = nullptr;
  if (!member_flags_host || n == 0) return;
  if (hipMalloc(&e->group_flags, n) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  ZKP_HIP(hipMemcpyAsync(e->group_flags, member_flags_host, n, hipMemcpyHostToDevice, ctx->cur->stream));
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
}
void bases_set_filter_bit(zkp_ctx* ctx, uint64_t handle, int bit) {
  ZKP_REQUIRE(bit >= -1 && bit < 3, ZKP_ERR_BAD_ARG);
  get_bases(ctx, handle)->filter_bit = bit;
}
bool bases_same_shape(zkp_ctx* ctx, uint64_t h1, uint64_t h2) {
  auto a = get_bases(ctx, h1), b = get_bases(ctx, h2);
  return a->curve == b->curve && a->n == b->n && a->c == b->c && a->W == b->W && a->wide == b->wide && a->cap == b->cap && a->lgk == b->lgk;
}

// ------------------------------------------------------------------------------------------- K5 digit scan
// Signed c-bit digits of one scalar, produced on the fly inside BOTH level-1 sort passes (histogram and scatter):
// the (bucket, point) entries are never materialised unsorted, so the scan reads 32 B per scalar (twice) and writes
// one 8-B (low key, point | sign) word per entry.  ark's `into_repr()` (prover.rs:150-161) is the fused from_mont().
struct DigitIter {
  uint32_t v[8];
  uint32_t carry;
  // window w: the first `wide` windows are c bits wide, the others c - 1 (BasesEntry::wide); nb = 2^(c-1)
  __device__ __forceinline__ void next(int w, int c, int wide, uint32_t nb, uint32_t& key, uint32_t& neg) {
    const int cw = w < wide ? c : c - 1;
    const int bit = w < wide ? w * c : wide * c + (w - wide) * (c - 1);
    const int limb = bit >> 5, sh = bit & 31;
    uint32_t d = 0;
    if (limb < 8) {
      uint64_t two = v[limb];
      if (limb + 1 < 8) two |= (uint64_t)v[limb + 1] << 32;
      d = (uint32_t)(two >> sh) & ((1u << cw) - 1);
    }
    d += carry;
    neg = 0;
    if (d > (1u << (cw - 1))) {
      d = (1u << cw) - d;
      neg = 1;
      carry = 1;
    } else {
      carry = 0;
    }
    key = d == 0 ? nb : d - 1;                      // nb == sentinel (zero digit)
  }
};
template <class FrP>
__device__ __forceinline__ DigitIter load_scalar(const uint32_t* __restrict__ scalars, size_t i, int montgomery) {
  Fp<FrP> s = Fp<FrP>::load(scalars + i * 8);
  if (montgomery) s = s.from_mont();
  DigitIter it;
#pragma unroll
  for (int l = 0; l < 8; l++) it.v[l] = s.v[l];
  it.carry = 0;
  return it;
}

// ------------------------------------------------------------------------------------------- exclusive scan (u32)
// three launches: per-2048-chunk sums -> one block scans the chunk sums (tiles of 1024 with a running carry) ->
// per-chunk exclusive scan + base.  Used for the sort's (bin, block) histogram and for the task offsets.
constexpr int SCAN_CHUNK = 2048;
__global__ __launch_bounds__(256) void scan_reduce_kernel(const uint32_t* __restrict__ in, size_t n,
                                                          uint32_t* __restrict__ sums) {
  ZKP_SORT_PRIO();
  __shared__ uint32_t red[256];
  size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
  uint32_t acc = 0;
  for (int k = 0; k < SCAN_CHUNK / 256; k++) {
    size_t i = base + (size_t)k * 256 + threadIdx.x;
    if (i < n) acc += in[i];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* __restrict__ sums, size_t m) {
  ZKP_SORT_PRIO();
  __shared__ uint32_t buf[1024];
  uint32_t carry = 0;
  for (size_t t0 = 0; t0 < m; t0 += 1024) {
    size_t i = t0 + threadIdx.x;
    uint32_t v = i < m ? sums[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      uint32_t add = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < m) sums[i] = carry + buf[threadIdx.x] - v;       // exclusive
    uint32_t total = buf[1023];
    __syncthreads();
    carry += total;
  }
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t* __restrict__ in, size_t n,
                                                         const uint32_t* __restrict__ sums,
                                                         uint32_t* __restrict__ out) {
  ZKP_SORT_PRIO();
  __shared__ uint32_t part[256];
  size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
  constexpr int PER = SCAN_CHUNK / 256;                         // 8 consecutive elements per thread
  uint32_t v[PER];
  uint32_t acc = 0;
  for (int k = 0; k < PER; k++) {
    size_t i = base + (size_t)threadIdx.x * PER + k;
    v[k] = i < n ? in[i] : 0;
    acc += v[k];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + part[threadIdx.x] - acc;
  for (int k = 0; k < PER; k++) {
    size_t i = base + (size_t)threadIdx.x * PER + k;
    if (i < n) out[i] = run;
    run += v[k];
  }
}
static void exclusive_scan_u32(hipStream_t st, const uint32_t* in, uint32_t* out, size_t n, DevBuf& tmp) {
  if (n == 0) return;
  size_t chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  uint32_t* sums = tmp.as<uint32_t>(chunks + 1);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(chunks), dim3(256), 0, st, in, n, sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, chunks);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(chunks), dim3(256), 0, st, in, n, sums, out);
}

// ------------------------------------------------------------------------------------------- K6 bucket sort
// Hand-written two-level counting sort of the (bucket, point) entries — it only has to GROUP entries by bucket
// (order inside a bucket is irrelevant: EC addition commutes), which is cheaper than a general radix sort:
//   level 1  bins = top <= 10 bits of the bucket id.  Per 8192-entry tile: LDS histogram (one pass over the
//            keys), global exclusive scan of the (bin, tile) counts, then a scatter pass in which an LDS atomic
//            hands every entry its slot inside its bin's region.
//   level 2  one workgroup per bin (~16 K entries, <= 1024 distinct low keys): LDS histogram, LDS scan — which
//            directly yields start/end of every bucket of the bin — and an LDS-atomic scatter of the values.
// LDS atomics resolve same-bucket conflicts inside a wave in hardware; no global atomics on the data path.
constexpr int SORT_H1_MAX = 13;        // level-1 bins <= 8192 (static LDS histogram, 32 KiB)
constexpr uint32_t SORT_BIN_TARGET = 16384;   // entries per level-1 bin the level-2 kernel stages in LDS
constexpr int SORT_L_MAX = 13;         // level-2 keys per bin <= 8192 (dynamic LDS)
constexpr int SORT_SCALARS = 2048;    // default; larger MSMs use larger tiles (runtime `tile`) to keep the tile count ~1200
constexpr int SORT_SCALARS_UNUSED_ = 0;     // scalars per workgroup in the level-1 passes (8 per lane): larger tiles = smaller (bin x tile) count matrix and longer contiguous runs per bin in the scatter (512 -> 2048: +2 % proofs/s)
// NT = threads per workgroup.  Round 6: all workgroups of the level-1 passes are resident at once, so a pass lasts as long as ONE
// workgroup; a tile's scalars are spread over more lanes (profiles/r06_sort_stage_ab.txt).
template <class FrP, int NT>
__global__ __launch_bounds__(NT) void sort_hist_kernel(const uint32_t* __restrict__ scalars, size_t n, size_t offset,
                                                        const uint8_t* __restrict__ inf, int montgomery, int c, int W, int wide,
                                                        uint32_t nb, int L, uint32_t nbins1,
                                                        uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t tile,
                                                        int lgk) {
  ZKP_SORT_PRIO();
  const uint32_t gmask = (1u << lgk) - 1;                          // window w -> bucket set w & gmask, table copy w >> lgk
  __shared__ uint32_t cnt[(1 << SORT_H1_MAX) + 1];
  for (uint32_t i = threadIdx.x; i <= nbins1; i += NT) cnt[i] = 0;
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < tile; j += NT) {          // (tile is a multiple of 256, not necessarily of NT)
    size_t i = (size_t)blockIdx.x * tile + j;
    if (i < n && !(inf && inf[offset + i])) {
      DigitIter it = load_scalar<FrP>(scalars, i, montgomery);
      for (int w = 0; w < W; w++) {
        uint32_t key, neg;
        it.next(w, c, wide, nb, key, neg);
        if (key < nb) atomicAdd(&cnt[((((uint32_t)w & gmask) << (c - 1)) | key) >> L], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < nbins1; i += NT) hist[(size_t)i * nblocks + blockIdx.x] = cnt[i];
}
template <class FrP>
__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint32_t* __restrict__ scalars, size_t n,
                                                           size_t offset, const uint8_t* __restrict__ inf,
                                                           int montgomery, size_t ntab, int c, int W, int wide, uint32_t nb,
                                                           int L, uint32_t nbins1, const uint32_t* __restrict__ offs,
                                                           uint32_t nblocks, uint32_t tile,
                                                           uint64_t* __restrict__ kv, int lgk,
                                                           const uint8_t* __restrict__ group_flags) {   // (low key << 32) | val
  ZKP_SORT_PRIO();
  const uint32_t gmask = (1u << lgk) - 1;
  __shared__ uint32_t cur[(1 << SORT_H1_MAX) + 1];
  for (uint32_t i = threadIdx.x; i < nbins1; i += 256) cur[i] = offs[(size_t)i * nblocks + blockIdx.x];
  __syncthreads();
  const uint32_t lmask = (1u << L) - 1;
  for (uint32_t rep = 0; rep < tile / 256; rep++) {
    size_t i = (size_t)blockIdx.x * tile + rep * 256 + threadIdx.x;
    if (i < n && !(inf && inf[offset + i])) {
      DigitIter it = load_scalar<FrP>(scalars, i, montgomery);
      // shared level-1 pass: bit 61 + k of every entry of this base = "identity in member k of the group" (sort_bin_kernel)
      const uint64_t gbits = group_flags ? (uint64_t)(group_flags[offset + i] & 7u) << 61 : 0;
      for (int w = 0; w < W; w++) {
        uint32_t key, neg;
        it.next(w, c, wide, nb, key, neg);
        if (key < nb) {                                // zero digits are dropped here
          const uint32_t fk = (((uint32_t)w & gmask) << (c - 1)) | key;     // window w = k*j + m: bucket set m, table copy j
          uint32_t pos = atomicAdd(&cur[fk >> L], 1u);
          uint32_t val = (uint32_t)((size_t)(w >> lgk) * ntab + offset + i) | (neg << 31);
          kv[pos] = gbits | ((uint64_t)(fk & lmask) << 32) | val;   // one 8-B store per entry
        }
      }
    }
  }
}
// Exclusive prefix sum of one value per thread over a workgroup of NT threads: wave scan by lane shuffles, the NT / 64 wave totals
// through LDS (tmp: >= NT / 64 + 1 words), three barriers instead of the 2 log2(NT) of a Hillis-Steele pass over LDS (round 6: the
// level-1 / level-2 sort kernels are latency chains of such phases).  Returns the exclusive prefix; *total = the sum over the workgroup.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t* total) {
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  if (lane == 63u) tmp[wave] = incl;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int w = 0; w < NT / 64; w++) {
      const uint32_t x = tmp[w];
      tmp[w] = run;
      run += x;
    }
    tmp[NT / 64] = run;
  }
  __syncthreads();
  const uint32_t res = tmp[wave] + incl - v;
  *total = tmp[NT / 64];
  __syncthreads();                                              // tmp may be reused by the caller's next scan
  return res;
}

// ------------------------------------------------------------------------------------------- staged level-1 scatter (round 3)
// Counter increment for a wave in which many lanes may hit the SAME counter (skewed scalars: a boolean witness sends every
// non-zero digit of window 0 to bucket 0, equal scalars collide in every window): the lanes that share the first active lane's
// counter are found with a ballot, ranked with mbcnt and served by ONE LDS atomic; the rest fall through to individual atomics.
// Returns the value of the counter before this lane's increment.  Must be called by all lanes of the wave.
__device__ __forceinline__ uint32_t wave_agg_inc(uint32_t* cnt, uint32_t idx, bool active) {
  const uint64_t act = __ballot(active);
  uint32_t res = 0;
  if (act == 0) return 0;                                            // wave-uniform
  const int leader = __ffsll((unsigned long long)act) - 1;
  const uint32_t lidx = (uint32_t)__builtin_amdgcn_readlane((int)idx, leader);
  const bool same = active && idx == lidx;
  const uint64_t m = __ballot(same);
  const uint32_t pc = (uint32_t)__popcll(m);
  if (pc >= 8) {                                                     // wave-uniform: a heavy hitter
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&cnt[lidx], pc);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (same) res = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    else if (active) res = atomicAdd(&cnt[idx], 1u);
  } else if (active) {
    res = atomicAdd(&cnt[idx], 1u);
  }
  return res;
}

// Level-1 scatter with the entries of a sub-round (`sub` scalars, <= SORT_STAGE_BYTES of entries) grouped by bin in LDS first:
// count per bin (LDS) -> exclusive scan -> entries placed bin by bin in the LDS stage -> copied out so that consecutive lanes
// write consecutive entries of one (bin, tile) run.  The one-entry-per-lane version above turns every 8-byte store into a
// 32-byte fabric write (PMC: 9.0 M 32-byte + 2.3 M 64-byte write requests = 434 MB for 125 MB of entries); here a run of k
// entries costs ceil(8k / 32) + 1 requests at most.  The bin id rides in bits 45..57 of the staged word (level 2 ignores them).
#ifndef ZKP_SORT_STAGE_BYTES              // A/B builds (ZKP_BUILD_DEFS_msm): 28 KiB = five workgroups per CU instead of two
#define ZKP_SORT_STAGE_BYTES (56 * 1024)
#endif
constexpr uint32_t SORT_STAGE_BYTES = ZKP_SORT_STAGE_BYTES;
template <class FrP, int NT>
__global__ __launch_bounds__(NT) void sort_scatter_staged_kernel(const uint32_t* __restrict__ scalars, size_t n,
                                                                  size_t offset, const uint8_t* __restrict__ inf,
                                                                  int montgomery, size_t ntab, int c, int W, int wide, uint32_t nb,
                                                                  int L, uint32_t nbins1, const uint32_t* __restrict__ offs,
                                                                  uint32_t nblocks, uint32_t tile, uint64_t* __restrict__ kv,
                                                                  int lgk, const uint8_t* __restrict__ group_flags, uint32_t sub) {
  ZKP_SORT_PRIO();
  const uint32_t gmask = (1u << lgk) - 1;
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  __shared__ uint32_t pre[NT];
  uint32_t* cur = sm;                                     // [nbins1] next free slot of (bin, this tile) in kv
  uint32_t* cnt = cur + nbins1;                           // [nbins1] counts -> cursors inside the stage
  uint64_t* stage = reinterpret_cast<uint64_t*>(cnt + ((nbins1 + 3) & ~3u));
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < nbins1; i += NT) cur[i] = offs[(size_t)i * nblocks + blockIdx.x];
  const uint32_t lmask = (1u << L) - 1;
  const uint32_t per = (nbins1 + NT - 1) / NT;
  for (uint32_t s0 = 0; s0 < tile; s0 += sub) {
    for (uint32_t i = tid; i < nbins1; i += NT) cnt[i] = 0;
    __syncthreads();
    // phase A: counts per bin
    for (uint32_t rep = 0; rep * NT < sub; rep++) {               // every lane of a wave runs the loop (wave_agg_inc): `live` masks
      const uint32_t j = rep * NT + tid;
      const size_t i = (size_t)blockIdx.x * tile + s0 + j;
      const bool live = j < sub && s0 + j < tile && i < n && !(inf && inf[offset + i]);
      DigitIter it = load_scalar<FrP>(scalars, live ? i : 0, montgomery);
      for (int w = 0; w < W; w++) {
        uint32_t key, neg;
        it.next(w, c, wide, nb, key, neg);
        const uint32_t fk = (((uint32_t)w & gmask) << (c - 1)) | key;
        (void)wave_agg_inc(cnt, fk >> L, live && key < nb);
      }
    }
    __syncthreads();
    // exclusive scan of the counts, in place
    uint32_t total;
    {
      uint32_t acc = 0;
      for (uint32_t k = 0; k < per; k++) {
        const uint32_t idx = tid * per + k;
        if (idx < nbins1) acc += cnt[idx];
      }
      uint32_t run = block_excl_scan<NT>(acc, pre, &total);
      for (uint32_t k = 0; k < per; k++) {
        const uint32_t idx = tid * per + k;
        if (idx < nbins1) {
          const uint32_t v = cnt[idx];
          cnt[idx] = run;
          run += v;
        }
      }
    }
    __syncthreads();
    // phase B: entries into the stage, bin by bin
    for (uint32_t rep = 0; rep * NT < sub; rep++) {               // every lane of a wave runs the loop (wave_agg_inc): `live` masks
      const uint32_t j = rep * NT + tid;
      const size_t i = (size_t)blockIdx.x * tile + s0 + j;
      const bool live = j < sub && s0 + j < tile && i < n && !(inf && inf[offset + i]);
      DigitIter it = load_scalar<FrP>(scalars, live ? i : 0, montgomery);
      const uint64_t gbits = live && group_flags ? (uint64_t)(group_flags[offset + i] & 7u) << 61 : 0;
      for (int w = 0; w < W; w++) {
        uint32_t key, neg;
        it.next(w, c, wide, nb, key, neg);
        const uint32_t fk = (((uint32_t)w & gmask) << (c - 1)) | key;
        const bool act = live && key < nb;
        const uint32_t pos = wave_agg_inc(cnt, fk >> L, act);
        if (act) {
          const uint32_t val = (uint32_t)((size_t)(w >> lgk) * ntab + offset + i) | (neg << 31);
          stage[pos] = gbits | ((uint64_t)(fk >> L) << 45) | ((uint64_t)(fk & lmask) << 32) | val;
        }
      }
    }
    __syncthreads();
    // copy out: cnt[b] is now the END of bin b inside the stage, i.e. the start of bin b + 1
    for (uint32_t j = tid; j < total; j += NT) {
      const uint64_t e = stage[j];
      const uint32_t b = (uint32_t)(e >> 45) & 0x1fffu;
      const uint32_t lo = b ? cnt[b - 1] : 0;
      kv[cur[b] + (j - lo)] = e;
    }
    __syncthreads();
    for (uint32_t b = tid; b < nbins1; b += NT) cur[b] += cnt[b] - (b ? cnt[b - 1] : 0);
    __syncthreads();
  }
}
// (round 2: level 1 as ONE kernel per tile — LDS histogram, LDS scan, entries scattered into the tile's own contiguous region
//  of kv so that the 8-byte stores merge in L2, level 2 gathering one run per tile — was built, bit-exact, and measured:
//  the level-1 kernel alone took 0.26 ms, as much as histogram + count scan + scatter together before (the passes are bound
//  by the LDS atomics on ~1000 random bins and the digit extraction, not by the store amplification), and level 2 got slower
//  reading 1200 short runs per bin: A-query MSM 2.66 vs 2.37 ms, 106-108 vs 113 proofs/s.  Reverted; commit 'Level-1 bucket
//  sort as ONE kernel per tile' holds the code.)
constexpr uint32_t SORT_BIN_THREADS = 1024;
#ifndef ZKP_SORT_BIN_STAGE                      // A/B builds: 18432 (72 KiB) lets two workgroups share a CU
#define ZKP_SORT_BIN_STAGE 20480
#endif
constexpr uint32_t SORT_BIN_STAGE = ZKP_SORT_BIN_STAGE;      // values staged in LDS (80 KiB) so the output is written fully coalesced
__global__ __launch_bounds__(SORT_BIN_THREADS) void sort_bin_kernel(const uint64_t* __restrict__ kv,
                                                                    const uint32_t* __restrict__ offs,
                                                                    uint32_t nblocks, int L,
                                                                    uint32_t* __restrict__ vout,
                                                                    uint32_t* __restrict__ start,
                                                                    uint32_t* __restrict__ end, int drop_bit) {
  ZKP_SORT_PRIO();
  extern __shared__ uint32_t sm[];                     // [nk] counters / cursors, then [SORT_BIN_STAGE] staged values
  __shared__ uint32_t pre[SORT_BIN_THREADS];
  const uint32_t nk = 1u << L;
  uint32_t* cnt = sm;
  uint32_t* stage = sm + nk;
  const uint32_t b = blockIdx.x, T = SORT_BIN_THREADS, t = threadIdx.x;
  const uint32_t lo = offs[(size_t)b * nblocks], hi = offs[(size_t)(b + 1) * nblocks];
  const bool staged = hi - lo <= SORT_BIN_STAGE;       // block-uniform
  for (uint32_t i = t; i < nk; i += T) cnt[i] = 0;
  __syncthreads();
  // drop_bit >= 0: the level-1 pass was shared by a group of queries; entries whose base is the identity in THIS query carry
  // bit 61 + drop_bit (sort_scatter_kernel) and are dropped here.  The low key sits in bits 32 .. 32 + L - 1.
  const uint64_t dmask = drop_bit >= 0 ? (uint64_t)1 << (61 + drop_bit) : 0;
  const uint32_t kmask = nk - 1;
  for (uint32_t i = lo + t; i < hi; i += T) {
    const uint64_t x = kv[i];
    if (!(x & dmask)) atomicAdd(&cnt[(uint32_t)(x >> 32) & kmask], 1u);
  }
  __syncthreads();
  // exclusive scan of the nk counters: `per` consecutive counters per thread + a workgroup scan of the partials
  {
    const uint32_t per = (nk + T - 1) / T;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per; k++) {
      uint32_t idx = t * per + k;
      if (idx < nk) acc += cnt[idx];
    }
    uint32_t total_unused;
    uint32_t run = block_excl_scan<(int)SORT_BIN_THREADS>(acc, pre, &total_unused);
    for (uint32_t k = 0; k < per; k++) {
      uint32_t idx = t * per + k;
      if (idx < nk) {
        uint32_t v = cnt[idx];
        start[((size_t)b << L) + idx] = lo + run;
        end[((size_t)b << L) + idx] = lo + run + v;
        cnt[idx] = run;                                 // cursor, relative to lo
        run += v;
      }
    }
  }
  __syncthreads();
  for (uint32_t i = lo + t; i < hi; i += T) {
    const uint64_t x = kv[i];
    if (x & dmask) continue;
    uint32_t pos = atomicAdd(&cnt[(uint32_t)(x >> 32) & kmask], 1u);
    if (staged) stage[pos] = (uint32_t)x;
    else vout[lo + pos] = (uint32_t)x;                  // oversized bin (skewed scalars): direct scatter
  }
  if (staged) {
    __syncthreads();
    for (uint32_t i = t; i < hi - lo; i += T) vout[lo + i] = stage[i];
  }
}

// ------------------------------------------------------------------------------------------- task scheduling
// A bucket of len entries becomes ceil(len / CAP) tasks.  Tasks are then ordered by length (counting sort on
// CAP+1 bins) so that the 64 lanes of a wave walk runs of (almost) equal length: with Poisson-distributed
// bucket sizes this turns ~65 % lane utilisation into > 95 %, and it bounds the serial chain of any lane by CAP
// even for adversarial inputs (all scalars equal) or the thinly populated top window.
// tmeta layout (uint32): [0] n_long  [1..CAP+1] histogram by length  [256..256+CAP] cursors
constexpr int TM_NLONG = 0, TM_HIST = 1, TM_CUR = 256, TM_WORDS = 512;

__global__ void task_count_kernel(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, uint32_t nb,
                                  uint32_t* __restrict__ tcount, uint32_t cap) {
  ZKP_SORT_PRIO();
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nb) return;
  tcount[b] = b < nb ? (end[b] - start[b] + cap - 1) / cap : 0;
}

// Empty buckets get the identity (all-zero bytes: zz == 0) here instead of a memset of the whole bucket array in front of every
// accumulate launch (75 / 151 MB of writes per 2^20 G1 / G2 MSM, 0.45 GB per proof): every non-empty bucket is written by the
// accumulate, redo or combine kernels.
__global__ __launch_bounds__(256) void zero_empty_buckets_kernel(const uint32_t* __restrict__ start,
                                                                 const uint32_t* __restrict__ end, uint32_t nb,
                                                                 char* __restrict__ buckets, uint32_t xb) {
  ZKP_SORT_PRIO();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb || end[b] != start[b]) return;
  uint4* p = reinterpret_cast<uint4*>(buckets + (size_t)b * xb);
  for (uint32_t k = 0; k < xb / 16; k++) p[k] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(1024) void task_fill_kernel(const uint32_t* __restrict__ start,
                                                        const uint32_t* __restrict__ end,
                                                        const uint32_t* __restrict__ toff, uint32_t nb,
                                                        uint32_t* __restrict__ task_start,
                                                        uint32_t* __restrict__ task_len,
                                                        uint32_t* __restrict__ task_dst,
                                                        uint32_t* __restrict__ long_list,
                                                        uint32_t* __restrict__ tmeta, uint32_t cap) {
  ZKP_SORT_PRIO();
  __shared__ uint32_t hist[MSM_TASK_CAP + 1];
  for (int i = threadIdx.x; i <= (int)MSM_TASK_CAP; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) {
    const uint32_t s = start[b], len = end[b] - s;
    const uint32_t t0 = toff[b], nt = toff[b + 1] - t0;
    if (nt > 1) long_list[atomicAdd(&tmeta[TM_NLONG], 1u)] = b;
    for (uint32_t j = 0; j < nt; j++) {
      uint32_t l = min(cap, len - j * cap);
      task_start[t0 + j] = s + j * cap;
      task_len[t0 + j] = l;
      task_dst[t0 + j] = nt == 1 ? b : (0x80000000u | (t0 + j));
      atomicAdd(&hist[l], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= (int)MSM_TASK_CAP; i += blockDim.x)
    if (hist[i]) atomicAdd(&tmeta[TM_HIST + i], hist[i]);
}

// cursors: longest tasks first
__global__ void task_cursor_kernel(uint32_t* tmeta) {
  ZKP_SORT_PRIO();
  if (threadIdx.x || blockIdx.x) return;
  uint32_t acc = 0;
  for (int l = MSM_TASK_CAP; l >= 0; l--) {
    tmeta[TM_CUR + l] = acc;
    acc += tmeta[TM_HIST + l];
  }
}

__global__ __launch_bounds__(1024) void task_order_kernel(const uint32_t* __restrict__ task_start,
                                                         const uint32_t* __restrict__ task_len,
                                                         const uint32_t* __restrict__ task_dst,
                                                         const uint32_t* __restrict__ n_tasks_dev,
                                                         uint32_t* __restrict__ tmeta, uint4* __restrict__ desc) {
  ZKP_SORT_PRIO();
  // counting-sort scatter: LDS histogram gives each task its rank among the block's tasks of equal length;
  // one global atomic per (block, length) reserves the block's slice of that length's output range.
  __shared__ uint32_t cnt[MSM_TASK_CAP + 1];
  __shared__ uint32_t base[MSM_TASK_CAP + 1];
  for (int i = threadIdx.x; i <= (int)MSM_TASK_CAP; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = t < *n_tasks_dev;
  uint32_t l = 0, rank = 0;
  if (live) {
    l = task_len[t];
    rank = atomicAdd(&cnt[l], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= (int)MSM_TASK_CAP; i += blockDim.x)
    if (cnt[i]) base[i] = atomicAdd(&tmeta[TM_CUR + i], cnt[i]);
  __syncthreads();
  // the schedule holds the task DESCRIPTORS in order (not indices into the task arrays): the accumulate kernel reads them
  // coalesced — round 3: 1.9 M random 128-B fetches less per 2^20 MSM (PMC: TCC_EA0_RDREQ)
  if (live) desc[base[l] + rank] = make_uint4(task_start[t], l, task_dst[t], t);
}

// ZKP_DEBUG_MSM=1: consistency check of the task schedule (one line per MSM on stdout), with a device sync either side
__global__ void sched_check_kernel(const uint32_t* toff, const uint32_t* long_list, const uint32_t* tmeta, uint32_t nb,
                                   uint32_t max_tasks, uint32_t* rep) {
  const uint32_t n_long = tmeta[TM_NLONG], n_tasks = toff[nb];
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    if (toff[b + 1] < toff[b]) atomicAdd(&rep[0], 1u);
    atomicMax(&rep[1], toff[b + 1] - toff[b]);
  }
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_long; w += gridDim.x * blockDim.x)
    if (long_list[w] >= nb) atomicAdd(&rep[2], 1u);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rep[3] = n_long;
    rep[4] = n_tasks;
    rep[5] = n_tasks > max_tasks;
  }
}

static void msm_run_entry(zkp_ctx* ctx, const BasesEntry* be, size_t offset, const uint64_t* scalars_dev, size_t n,
                          bool montgomery, uint64_t* out_xyz_host, void* out_dev_xyzz, float* ms_accumulate,
                          uint64_t* n_entries, int ws_idx, int sort_src, float* ms_scan, int l1_src = -1);
void msm_run(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* scalars_dev, size_t n, bool montgomery,
             uint64_t* out_xyz_host, void* out_dev_xyzz, float* ms_accumulate, uint64_t* n_entries, int ws_idx,
             int sort_src, float* ms_scan, int l1_src) {
  struct ChainReset {                // one-shot requests: never leak into a later MSM, whatever this call throws (bad handle, ...)
    zkp_ctx* c;
    ~ChainReset() {
      c->msm_defer_reduce = false;
      c->msm_acc_into = -1;
    }
  } chain_reset{ctx};
  auto be = get_bases(ctx, handle);
  // Chunked MSM (round 4): a large MSM that runs on its own (Marlin's h_2 and opening witnesses: 6.3 M points) spent 3 ms in its
  // bucket sort — memory-bound kernels with the vector ALUs idle — before 5.6 ms of VALU-bound accumulation.  Split by index into
  // chunks that share ONE bucket array (bucket chaining: every chunk but the last defers the reduction, every chunk but the first
  // accumulates on top of the stored buckets), alternating between this workspace and its partner (ws ^ 2: own stream and sort
  // scratch), so that the sort of chunk k + 1 runs under the accumulation of chunk k.  The result is the same group element.
  // ZKP_MSM_CHUNK=<points per chunk> (0 = off); not for MSMs that share a sort, belong to a sort group, are profiled, or are
  // part of a caller's own bucket chain.
  const size_t chunk_pts = (size_t)std::max<long long>(0, ctx->cfg.msm_chunk);        // zkp_ctx_config.msm_chunk_points / ZKP_MSM_CHUNK
  const bool chunkable = chunk_pts >= 1024 && n >= 2 * chunk_pts && be->group == 1 && !be->var && sort_src < 0 && l1_src < 0 &&
                         !be->group_flags && !ctx->profiling && !ms_accumulate && !ms_scan && !n_entries &&
                         !ctx->msm_defer_reduce && ctx->msm_acc_into < 0 && ctx->msm_bucket_ws < 0 && !ctx->batch_mode;
  if (!chunkable) {
    msm_run_entry(ctx, be.get(), offset, scalars_dev, n, montgomery, out_xyz_host, out_dev_xyzz, ms_accumulate, n_entries,
                  ws_idx, sort_src, ms_scan, l1_src);
    return;
  }
  // (per is rounded up to a multiple of 256, so the chunk count is recomputed from it: with a ZKP_MSM_CHUNK that is not a multiple
  //  of 256 the last of the original nch chunks could come out empty — ADVICE r4)
  const size_t nch0 = (n + chunk_pts - 1) / chunk_pts, per = ((n + nch0 - 1) / nch0 + 255) & ~(size_t)255;
  // Round 5: a SHORT first chunk.  Nothing overlaps the bucket sort of chunk 0 — a lone MSM (Marlin's h_2 commitment and the two
  // opening witnesses) shows 1-2 ms of sort kernels with the vector ALUs idle before its first accumulate launch (kernel trace,
  // profiles/r05_marlin_trace.txt) — so chunk 0 is per / ZKP_MSM_CHUNK_FIRST points (default 2: 57.7 -> 56.8 ms per Marlin proof, profiles/r05_marlin_ab.txt; 1 = equal chunks)
  // and the rest is split evenly.
  static const size_t first_div = [] { const char* e = getenv("ZKP_MSM_CHUNK_FIRST"); long v = e ? atol(e) : 2; return (size_t)(v < 1 ? 1 : v); }();
  const size_t first = first_div > 1 ? std::max<size_t>(((per / first_div) + 255) & ~(size_t)255, 1024) : 0;
  const size_t rest = first && first < n ? n - first : n;
  const size_t nrest = (rest + per - 1) / per, per_rest = ((rest + nrest - 1) / nrest + 255) & ~(size_t)255;
  const size_t nch = (first && first < n ? 1 : 0) + (rest + per_rest - 1) / per_rest;
  const int wa = ws_idx, wb = ws_idx ^ 2;                               // partner: 0 <-> 2, 1 <-> 3
  hipStream_t sa = wa == 0 ? ctx->cur->stream : ctx->cur->ws[wa].stream, sb = wb == 0 ? ctx->cur->stream : ctx->cur->ws[wb].stream;
  // the partner's stream joins behind everything this MSM's stream has seen (scalars complete, bucket array of `wa` free)
  ZKP_HIP(hipEventRecord(ctx->cur->ws[wa].l1_done, sa));
  ZKP_HIP(hipStreamWaitEvent(sb, ctx->cur->ws[wa].l1_done, 0));
  size_t done = 0;
  for (size_t k = 0; k < nch; k++) {
    const size_t len = std::min((k == 0 && first && first < n) ? first : per_rest, n - done);
    const bool last = k + 1 == nch;
    const int w = ((nch - 1 - k) & 1) ? wb : wa;                        // the last chunk runs on `wa`: result, `done` event, read-back
    ctx->msm_defer_reduce = !last;
    ctx->msm_acc_into = k > 0 ? wa : -1;
    ctx->msm_bucket_ws = wa;
    msm_run_entry(ctx, be.get(), offset + done, scalars_dev + 4 * done, len, montgomery, last ? out_xyz_host : nullptr,
                  last ? out_dev_xyzz : nullptr, nullptr, nullptr, w, -1, nullptr, -1);
    done += len;
  }
  // the partner stream's last chunk was consumed by a later chunk on `sa` through the acc_done event; nothing of this MSM is left on `sb`
}

// Reduction plan of the variable-base mode (per (c, W), cached in the context): block descriptors for the two
// segmented-sum stages.  Level l of the pairwise pyramid holds W windows x (nb_w >> l) entries; O_{l,w} = sum of the odd
// entries of window w at level l; R_t = O_{l,w} for t = c*w + l; result = sum_t 2^t R_t + sum_w 2^(c*w) root_w.
struct VarPlan {
  DevBuf d1, d2;
  uint32_t n1 = 0, n2 = 0;
};
static VarPlan& var_plan(zkp_ctx* ctx, int c, int W, hipStream_t st) {
  // owned by the context (released by msm_free_all / zkp_ctx_destroy, before the HIP runtime goes away): the descriptors
  // are immutable, one plan per (c, W)
  auto key = std::make_pair(c, W);
  auto it = ctx->var_plans.find(key);
  if (it != ctx->var_plans.end()) return *static_cast<VarPlan*>(it->second.get());
  const uint32_t nb_w = 1u << (c - 1);
  std::vector<SegDesc> s1, s2(256);
  uint32_t lvl_off = 0, cnt = (uint32_t)W * nb_w;
  for (auto& d : s2) d = SegDesc{0, 1, 0, 0};
  for (int l = 0; l < c - 1; l++) {
    const uint32_t pw = nb_w >> (l + 1);                         // odd entries per window at this level
    const uint32_t ch = std::min<uint32_t>((uint32_t)SEG_CHUNK, pw);
    for (int w = 0; w < W; w++) {
      const uint32_t first = (uint32_t)s1.size();
      for (uint32_t j = 0; j < pw / ch; j++)
        s1.push_back(SegDesc{lvl_off + (uint32_t)w * (nb_w >> l) + 1 + 2 * j * ch, 2, ch, (uint32_t)s1.size()});
      const int t = c * w + l;
      if (t < 256) s2[t] = SegDesc{first, 1, (uint32_t)s1.size() - first, (uint32_t)t};
    }
    lvl_off += cnt;
    cnt >>= 1;
  }
  for (int t = 0; t < 256; t++) s2[t].out = (uint32_t)t;
  auto vp = std::make_shared<VarPlan>();
  vp->n1 = (uint32_t)s1.size();
  vp->n2 = 256;
  ZKP_HIP(hipMemcpyAsync(vp->d1.as<SegDesc>(s1.size()), s1.data(), s1.size() * sizeof(SegDesc), hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemcpyAsync(vp->d2.as<SegDesc>(256), s2.data(), 256 * sizeof(SegDesc), hipMemcpyHostToDevice, st));
  ZKP_HIP(hipStreamSynchronize(st));                             // the host vectors die with this call
  ctx->var_plans[key] = vp;
  return *vp;
}

static void msm_run_entry(zkp_ctx* ctx, const BasesEntry* be, size_t offset, const uint64_t* scalars_dev, size_t n,
                          bool montgomery, uint64_t* out_xyz_host, void* out_dev_xyzz, float* ms_accumulate,
                          uint64_t* n_entries, int ws_idx, int sort_src, float* ms_scan, int l1_src) {
  const MsmVtbl* vt = be->vt;
  MsmWorkspace& ws = ctx->cur->ws[ws_idx];
  // bucket chaining (ctx.hpp): one-shot requests, consumed by this MSM
  const bool defer = ctx->msm_defer_reduce;
  const int into = ctx->msm_acc_into;
  const int bucket_ws = into >= 0 ? into : ctx->msm_bucket_ws;
  ctx->msm_defer_reduce = false;
  ctx->msm_acc_into = -1;
  ctx->msm_bucket_ws = -1;
  ZKP_REQUIRE(!(defer || into >= 0) || (n > 0 && !be->var && be->group == 1), ZKP_ERR_BAD_ARG);   // G1 only (msm_acc.hip)
  // sort_src: workspace whose sorted entries + task schedule this MSM reuses (same scalars, window configuration and identity
  // pattern); sort_src == ws_idx = the MSM that ran on this workspace just before (A -> L on one stream)
  const bool reuse = sort_src >= 0;
  MsmWorkspace& sw = reuse ? ctx->cur->ws[sort_src] : ws;          // owner of the sorted entries and the task schedule
  hipStream_t st = ws_idx == 0 ? ctx->cur->stream : ws.stream;
  const size_t XB = vt->bucket_bytes;                                // buckets, partial sums, pyramid levels
  const size_t jac_words = 3 * (size_t)vt->fN;
  uint32_t* out_jac = ws.out.as<uint32_t>(64 * 4);
  if (ms_accumulate) *ms_accumulate = 0.f;
  if (ms_scan) *ms_scan = 0.f;
  if (n_entries) *n_entries = 0;
  if (n == 0) {
    vt->write_identity(st, (char*)out_dev_xyzz, out_jac);
  } else {
    const int c = be->c, W = be->W, wide = be->wide;
    const int var = be->var ? 1 : 0;
    const int lgk = be->lgk, K = 1 << lgk;                             // bucket sets (window groups, BasesEntry::lgk)
    ZKP_REQUIRE(var == (lgk > 0), ZKP_ERR_BAD_ARG);
    const uint32_t nb_w = 1u << (c - 1);                               // buckets of one window (digit range)
    const int kbits = (c - 1) + lgk;                                   // bits of a bucket id
    const uint32_t nb = 1u << kbits;                                   // all buckets
    const size_t E = n * (size_t)W;
    ZKP_REQUIRE(E < 2147483000ull, ZKP_ERR_BAD_ARG);
    // l1_src >= 0 (and no full reuse): the level-1 pass over these scalars — digit scan, (bin, tile) counts, scatter into bins —
    // was run by workspace l1_src of this lane for a GROUP of queries (it dropped only the bases that are the identity in all of
    // them); this MSM runs its own level 2 on that list and drops its own identities there (BasesEntry::filter_bit).
    // l1_src == ws_idx: by the MSM that ran on this workspace just before.
    const bool l1_reuse = !reuse && l1_src >= 0;
    MsmWorkspace& lw = l1_reuse ? ctx->cur->ws[l1_src] : sw;           // owner of the level-1 output
    uint32_t* vals = sw.vals.as<uint32_t>(E + 8);                    // + 8: the accumulate kernel reads aligned groups of eight                        // values grouped by bucket (level-2 output)
    uint64_t* kv = lw.keys2.as<uint64_t>(E);                         // level-1 output: (low key, val) pairs
    const uint32_t* sc = reinterpret_cast<const uint32_t*>(scalars_dev);
    const int mont = montgomery ? 1 : 0;
    // K6: group entries by bucket (two-level counting sort); sorted values land back in `vals`
    // level-1 bins: enough of them that a bin (E / bins entries on average) fits the level-2 kernel's LDS stage; tiles:
    // 2048 scalars, more for large MSMs so that the (bin x tile) count matrix stays small
    static const int h1_env = [] { const char* e = getenv("ZKP_SORT_H1"); return e ? atoi(e) : 0; }();
    int H1 = 10;
    while (H1 < SORT_H1_MAX && (E >> H1) > SORT_BIN_TARGET) H1++;
    if (h1_env > 0) H1 = std::min(SORT_H1_MAX, h1_env);
    H1 = std::min(H1, kbits);
    if (kbits - H1 > SORT_L_MAX) H1 = kbits - SORT_L_MAX;
    const int LB = kbits - H1;                                         // low bits per level-1 bin
    const uint32_t nbins1 = 1u << H1;
    uint32_t tile = SORT_SCALARS;
    while ((n + tile - 1) / tile > 1536) tile += 256;
    const uint32_t nblocks = (uint32_t)((n + tile - 1) / tile);
    const size_t hist_n = (size_t)nbins1 * nblocks + 1;                // + total (== number of non-zero digits)
    uint32_t* hist = lw.sort_tmp.as<uint32_t>(2 * hist_n);
    uint32_t* offs = hist + hist_n;
    uint32_t* start = sw.offsets.as<uint32_t>(2 * (size_t)nb);
    uint32_t* end = start + nb;
    // see BasesEntry::sort_inf; a profiled (stand-alone, timed) MSM scans with its own flags so that its entry count is exact
    const uint8_t* scan_inf = be->sort_inf && !ctx->profiling ? be->sort_inf : be->inf;
    if (reuse) {
      if (sort_src != ws_idx) ZKP_HIP(hipStreamWaitEvent(st, sw.sorted, 0));
    } else if (l1_reuse) {
      if (l1_src != ws_idx) ZKP_HIP(hipStreamWaitEvent(st, lw.l1_done, 0));
    } else {
      ZKP_HIP(hipMemsetAsync(hist + hist_n - 1, 0, 4, st));
    }
    const uint8_t* grp = scan_inf == be->sort_inf ? be->group_flags : nullptr;   // group bits only with the group's scan flags
    // staged level-1 scatter (sort_scatter_staged_kernel): sub-rounds of `staged_sub` scalars whose entries fit the LDS stage;
    // not for > 2048 level-1 bins (the two counter arrays would leave one workgroup per CU) or very narrow windows
    static const bool staged_on = !(getenv("ZKP_SORT_STAGED") && atoi(getenv("ZKP_SORT_STAGED")) == 0);
    uint32_t staged_sub = 0;
    size_t staged_lds = 0;
    if (staged_on && nbins1 <= 2048 && (size_t)256 * W * 8 <= SORT_STAGE_BYTES) {
      staged_sub = (uint32_t)(SORT_STAGE_BYTES / ((size_t)W * 8)) / 256 * 256;
      staged_sub = std::min<uint32_t>(staged_sub, tile);
      staged_lds = ((size_t)nbins1 + ((nbins1 + 3) & ~3u)) * 4 + (size_t)staged_sub * W * 8;
    }
    const bool timed_scan = ms_scan && ctx->profiling && !reuse && !l1_reuse;      // K5 "scalar scan": histogram pass + count scan + scatter pass
    if (timed_scan) ZKP_HIP(hipEventRecord(ctx->ev2, st));
    if (reuse || l1_reuse) {
    } else {
      // threads per workgroup of the two level-1 passes (A/B: ZKP_SORT_NT_HIST / ZKP_SORT_NT_SCATTER = 256 restores rounds 3-5)
      static const int nt_hist = [] { const char* e = getenv("ZKP_SORT_NT_HIST"); return e ? atoi(e) : 1024; }();
      static const int nt_scat = [] { const char* e = getenv("ZKP_SORT_NT_SCATTER"); return e ? atoi(e) : 512; }();
      auto level1 = [&](auto tag) {
        using FrP = decltype(tag);
        auto hist_l = [&](auto nt) {
          constexpr int NT = decltype(nt)::value;
          hipLaunchKernelGGL((sort_hist_kernel<FrP, NT>), dim3(nblocks), dim3(NT), 0, st, sc, n, offset, scan_inf, mont, c, W, wide, nb_w,
                             LB, nbins1, hist, nblocks, tile, lgk);
        };
        if (nt_hist >= 1024) hist_l(std::integral_constant<int, 1024>{});
        else if (nt_hist >= 512) hist_l(std::integral_constant<int, 512>{});
        else hist_l(std::integral_constant<int, 256>{});
        exclusive_scan_u32(st, hist, offs, hist_n, ws.scan_tmp);
        if (staged_sub) {
          auto scat_l = [&](auto nt) {
            constexpr int NT = decltype(nt)::value;
            hipLaunchKernelGGL((sort_scatter_staged_kernel<FrP, NT>), dim3(nblocks), dim3(NT), staged_lds, st, sc, n, offset,
                               scan_inf, mont, be->n, c, W, wide, nb_w, LB, nbins1, offs, nblocks, tile, kv, lgk, grp, staged_sub);
          };
          if (nt_scat >= 1024) scat_l(std::integral_constant<int, 1024>{});
          else if (nt_scat >= 512) scat_l(std::integral_constant<int, 512>{});
          else scat_l(std::integral_constant<int, 256>{});
        } else {
          hipLaunchKernelGGL(sort_scatter_kernel<FrP>, dim3(nblocks), dim3(256), 0, st, sc, n, offset, scan_inf, mont, be->n, c, W, wide,
                             nb_w, LB, nbins1, offs, nblocks, tile, kv, lgk, grp);
        }
      };
      if (be->curve == ZKP_BN254) level1(Bn254Fr{});
      else level1(Bls381Fr{});
    }
    if (timed_scan) {
      ZKP_HIP(hipEventRecord(ctx->ev3, st));
      ZKP_HIP(hipEventSynchronize(ctx->ev3));
      ZKP_HIP(hipEventElapsedTime(ms_scan, ctx->ev2, ctx->ev3));
    }
    if (!reuse && !l1_reuse) ZKP_HIP(hipEventRecord(ws.l1_done, st));
    if (!reuse && !l1_reuse) ctx->mark(st, ":l1");
    if (!reuse)
      hipLaunchKernelGGL(sort_bin_kernel, dim3(nbins1), dim3(SORT_BIN_THREADS), ((size_t)4 << LB) + 4 * (size_t)SORT_BIN_STAGE, st,
                         kv, offs, nblocks, LB, vals, start, end, (l1_reuse || grp) ? be->filter_bit : -1);
    uint32_t* const sorted_vals = vals;
    // K7 scheduling: buckets -> tasks (<= CAP entries), ordered by length
    const uint32_t max_tasks = nb + (uint32_t)(E / be->cap) + 1;
    uint32_t* sched = sw.sched.as<uint32_t>((size_t)2 * (nb + 2) + (size_t)8 * max_tasks + TM_WORDS + 8);
    uint32_t* tcount = sched;                       // nb + 1
    uint32_t* toff = tcount + (nb + 2);             // nb + 1  (toff[nb] = number of tasks)
    uint32_t* task_start = toff + (nb + 2);
    uint32_t* task_len = task_start + max_tasks;
    uint32_t* task_dst = task_len + max_tasks;
    uint32_t* long_list = task_dst + max_tasks;
    uint32_t* tmeta = long_list + max_tasks;
    uint4* desc = reinterpret_cast<uint4*>((reinterpret_cast<uintptr_t>(tmeta + TM_WORDS) + 15) & ~(uintptr_t)15);   // max_tasks x 16 B
    if (!reuse) {
      ZKP_HIP(hipMemsetAsync(tmeta, 0, TM_WORDS * 4, st));
      hipLaunchKernelGGL(task_count_kernel, dim3((nb + 256) / 256), dim3(256), 0, st, start, end, nb, tcount, be->cap);
      exclusive_scan_u32(st, tcount, toff, (size_t)nb + 1, ws.scan_tmp2);
      // (1024 threads per workgroup since round 6: a quarter of the per-(block, length) global atomics on the 129 length counters)
      static const uint32_t tnt = [] { const char* e = getenv("ZKP_TASK_NT"); const int v = e ? atoi(e) : 1024; return (uint32_t)(v >= 1024 ? 1024 : v >= 512 ? 512 : 256); }();
      hipLaunchKernelGGL(task_fill_kernel, dim3((nb + tnt - 1) / tnt), dim3(tnt), 0, st, start, end, toff, nb, task_start,
                         task_len, task_dst, long_list, tmeta, be->cap);
      hipLaunchKernelGGL(task_cursor_kernel, dim3(1), dim3(64), 0, st, tmeta);
      hipLaunchKernelGGL(task_order_kernel, dim3((max_tasks + tnt - 1) / tnt), dim3(tnt), 0, st, task_start, task_len, task_dst,
                         toff + nb, tmeta, desc);
      ZKP_HIP(hipEventRecord(ws.sorted, st));
      ctx->mark(st, ":sorted");
    }
    // level l of the reduction pyramid lives at element offset lvl_off[l] of `buckets` (level 0 = buckets);
    // all-zero bytes are a valid identity (zz == 0), so empty buckets need no kernel
    // (into >= 0: the bucket array of workspace `into`, which holds the finished buckets of an MSM over the same bucket range;
    //  its accumulate + combine must be complete before this one's start)
    MsmWorkspace& bws = bucket_ws >= 0 ? ctx->cur->ws[bucket_ws] : ws;
    char* buckets = reinterpret_cast<char*>(bws.buckets.get((size_t)2 * nb * XB + XB));
    char* task_partial = reinterpret_cast<char*>(ws.partial.get((size_t)max_tasks * XB));
    static const bool zero_all = getenv("ZKP_MEMSET_BUCKETS") && atoi(getenv("ZKP_MEMSET_BUCKETS")) != 0;   // A/B: round-2 behaviour
    const uint32_t init = into >= 0 ? 1u : 0u;
    if (init) {
      ZKP_REQUIRE(bws.chain_nb == nb && bws.chain_xb == XB, ZKP_ERR_BAD_ARG);
      ZKP_HIP(hipStreamWaitEvent(st, bws.acc_done, 0));        // (recorded on whichever stream ran the MSM that filled them)
    } else if (zero_all) ZKP_HIP(hipMemsetAsync(buckets, 0, (size_t)nb * XB, st));
    else hipLaunchKernelGGL(zero_empty_buckets_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, start, end, nb, buckets, (uint32_t)XB);
    bws.chain_nb = 0;
    const bool timed = ms_accumulate && ctx->profiling;
    if (timed) ZKP_HIP(hipEventRecord(ctx->ev2, st));
    static const uint32_t force_redo = getenv("ZKP_DEBUG_FORCE_REDO") && atoi(getenv("ZKP_DEBUG_FORCE_REDO")) != 0 ? 2u : 0u;   // tests
    ctx->mark(st, ":acc0");
    vt->accumulate(st, be->table, sorted_vals, desc, toff + nb, max_tasks, buckets, task_partial,
                   ws.redo.as<uint32_t>((size_t)max_tasks + 1), init | force_redo);
    ctx->mark(st, ":acc1");
    if (timed) {
      ZKP_HIP(hipEventRecord(ctx->ev3, st));
      ZKP_HIP(hipEventSynchronize(ctx->ev3));
      ZKP_HIP(hipEventElapsedTime(ms_accumulate, ctx->ev2, ctx->ev3));
    }
    static const int dbg = [] { const char* e = getenv("ZKP_DEBUG_MSM"); return e ? atoi(e) : 0; }();
    if (dbg) {
      uint32_t* rep = ws.redo.as<uint32_t>((size_t)max_tasks + 1);     // accumulate is done with it
      ZKP_HIP(hipStreamSynchronize(st));
      ZKP_HIP(hipMemsetAsync(rep, 0, 32, st));
      hipLaunchKernelGGL(sched_check_kernel, dim3(256), dim3(256), 0, st, toff, long_list, tmeta, nb, max_tasks, rep);
      uint32_t h[8];
      ZKP_HIP(hipMemcpyAsync(h, rep, 32, hipMemcpyDeviceToHost, st));
      ZKP_HIP(hipStreamSynchronize(st));
      printf("[msm] group=%d n=%zu nb=%u max_tasks=%u: toff inversions=%u max tasks/bucket=%u bad long_list=%u n_long=%u n_tasks=%u overflow=%u\n",
             be->group, n, nb, max_tasks, h[0], h[1], h[2], h[3], h[4], h[5]);
      fflush(stdout);
    }
#ifdef ZKP_ABLATION   // ZKP_DEBUG_MSM=2 skips the combine step (wrong buckets): ablation builds only
    if (dbg != 2)
#endif
      vt->combine(st, long_list, tmeta + TM_NLONG, toff, task_partial, buckets, init);
    if (dbg) {
      hipError_t e = hipStreamSynchronize(st);
      printf("[msm] combine: %s\n", hipGetErrorString(e));
      fflush(stdout);
    }
    if (n_entries) {
      *n_entries = E;                                   // nominal: scalars x windows
      if (ctx->profiling) {                             // exact: what the scan emitted (identity bases / zero digits dropped)
        uint32_t emitted = 0;
        ZKP_HIP(hipMemcpyAsync(&emitted, offs + hist_n - 1, 4, hipMemcpyDeviceToHost, st));
        ZKP_HIP(hipStreamSynchronize(st));
        *n_entries = emitted;
      }
    }
    if (var) {
      // K8, variable-base: the pairwise pyramid runs over all W bucket sets at once (pairs never straddle windows), c - 1
      // levels deep; two descriptor-driven segmented sums give R_t; 256 lanes weigh them by 2^t (<= 255 doublings each)
      VarPlan& vp = var_plan(ctx, c, K, st);
      uint32_t lvl_off = 0, cnt = nb;
      for (int l = 0; l < c - 1; l++) {
        vt->pair(st, buckets + (size_t)lvl_off * XB, buckets + (size_t)(lvl_off + cnt) * XB, cnt / 2);
        lvl_off += cnt;
        cnt /= 2;
      }
      const char* roots = buckets + (size_t)lvl_off * XB;            // cnt == K (one root per bucket set)
      char* partial = reinterpret_cast<char*>(ws.tmp.get(((size_t)vp.n1 + 256 + 8) * XB));
      char* R = partial + (size_t)vp.n1 * XB;
      vt->segsum_desc(st, buckets, vp.d1.as<SegDesc>(vp.n1), vp.n1, partial);
      vt->segsum_desc(st, partial, vp.d2.as<SegDesc>(256), 256, R);
      vt->final_var(st, R, roots, c, K, (char*)out_dev_xyzz, out_jac);
      ZKP_HIP(hipGetLastError());
      if (out_xyz_host) {
        ZKP_HIP(hipMemcpyAsync(out_xyz_host, out_jac, jac_words * 4, hipMemcpyDeviceToHost, st));
        ZKP_HIP(hipStreamSynchronize(st));
      }
      return;
    }
    if (defer) {
      // the buckets stay as they are for the MSM that reduces them together with its own (msm_acc_into); this one contributes
      // the identity to whatever sums the results
      bws.chain_nb = nb;
      bws.chain_xb = XB;
      ZKP_HIP(hipEventRecord(bws.acc_done, st));
      vt->write_identity(st, (char*)out_dev_xyzz, out_jac);
      ZKP_HIP(hipGetLastError());
      ZKP_REQUIRE(!out_xyz_host, ZKP_ERR_BAD_ARG);
      return;
    }
#ifdef ZKP_ABLATION
    if (ctx->dbg_skip_k8) return;                  // ablation experiments only (wrong result)
#endif
    // K8: pyramid
    const int L = c - 1;                           // levels with odd entries: 0..L-1 ; root = level L
    SegPlan plan{};
    plan.L = L;
    const uint32_t chunk = ctx->batch_mode ? SEG_CHUNK_BATCH : SEG_CHUNK;
    plan.chunk = chunk;
    uint32_t lvl_off = 0, cnt = nb, blocks = 0;
    // Round 6: the one-workgroup top of the pyramid and the segmented sums of the levels below it run in ONE launch (msm_group.hip
    // pair_top_segsum_kernel); the second stage then sums the partials of the low levels and, directly from the pyramid, the odd
    // entries of the top's levels (<= PAIR_TOP_MAX / 2 = one block each).  ZKP_PAIR_TOP_FUSE_SEG=0: pair_top, then both stages (rounds 3-5).
    static const bool fuse_seg = !(getenv("ZKP_PAIR_TOP_FUSE_SEG") && atoi(getenv("ZKP_PAIR_TOP_FUSE_SEG")) == 0);
    int l_top = -1;                                // first level the fused top produces the successor of
    char* top_base = nullptr;
    uint32_t top_cnt = 0;
    for (int l = 0; l < L; l++) {
      uint32_t next_off = lvl_off + cnt;
      static const bool top_fused = !(getenv("ZKP_PAIR_TOP") && atoi(getenv("ZKP_PAIR_TOP")) == 0);
      // (rounded down to a power of two: the fused top starts at the level whose size EQUALS top_max)
      static const uint32_t top_max = [] {
        const char* e = getenv("ZKP_PAIR_TOP_MAX");
        uint32_t v = e ? (uint32_t)atoi(e) : PAIR_TOP_MAX;
        if (v < 2) v = PAIR_TOP_MAX;
        while (v & (v - 1)) v &= v - 1;
        return v;
      }();
      if (!top_fused || cnt > top_max) vt->pair(st, buckets + (size_t)lvl_off * XB, buckets + (size_t)next_off * XB, cnt / 2);
      else if (cnt == top_max || l == 0) {         // this level and all above it
        if (fuse_seg && cnt / 2 <= chunk) {
          l_top = l;
          top_base = buckets + (size_t)lvl_off * XB;
          top_cnt = cnt;
        } else {
          vt->pair_top(st, buckets + (size_t)lvl_off * XB, cnt);
        }
      }
      plan.first_block[l] = blocks;
      plan.off[l] = lvl_off + 1;
      plan.stride[l] = 2;
      plan.count[l] = cnt / 2;
      blocks += (cnt / 2 + chunk - 1) / chunk;
      lvl_off = next_off;
      cnt /= 2;
    }
    plan.first_block[L] = blocks;
    const char* root = buckets + (size_t)lvl_off * XB;      // cnt == 1
    char* partial = reinterpret_cast<char*>(ws.tmp.get(((size_t)blocks + 64) * XB));
    char* Obuf = partial + (size_t)blocks * XB;
    if (L > 0 && l_top >= 0) {
      SegPlan low = plan;                                    // levels 0 .. l_top - 1: complete before this launch
      low.L = l_top;
      const uint32_t blocks_low = plan.first_block[l_top];
      vt->pair_top_segsum(st, top_base, top_cnt, buckets, &low, partial, blocks_low);
      SegPlan p2{};
      p2.L = L;
      p2.chunk = chunk;
      for (int l = 0; l < L; l++) {
        p2.first_block[l] = l;
        if (l < l_top) {                                     // the level's partials
          p2.off[l] = plan.first_block[l];
          p2.stride[l] = 1;
          p2.count[l] = plan.first_block[l + 1] - plan.first_block[l];
        } else {                                             // the level's odd entries themselves
          p2.off[l] = plan.off[l];
          p2.stride[l] = plan.stride[l];
          p2.count[l] = plan.count[l];
        }
        ZKP_REQUIRE(p2.count[l] <= chunk, ZKP_ERR_BAD_ARG);
      }
      p2.first_block[L] = L;
      vt->segsum(st, partial, &p2, Obuf, L, buckets, l_top);
    } else if (L > 0) {
      vt->segsum(st, buckets, &plan, partial, blocks, nullptr, 1 << 30);
      SegPlan p2{};
      p2.L = L;
      p2.chunk = chunk;
      for (int l = 0; l < L; l++) {
        p2.first_block[l] = l;
        p2.off[l] = plan.first_block[l];
        p2.stride[l] = 1;
        p2.count[l] = plan.first_block[l + 1] - plan.first_block[l];
        ZKP_REQUIRE(p2.count[l] <= chunk, ZKP_ERR_BAD_ARG);
      }
      p2.first_block[L] = L;
      vt->segsum(st, partial, &p2, Obuf, L, nullptr, 1 << 30);
    }
    vt->final(st, Obuf, L, root, (char*)out_dev_xyzz, out_jac);
    ctx->mark(st, ":reduced");
  }
  ZKP_HIP(hipGetLastError());
  if (out_xyz_host) {
    ZKP_HIP(hipMemcpyAsync(out_xyz_host, out_jac, jac_words * 4, hipMemcpyDeviceToHost, st));
    ZKP_HIP(hipStreamSynchronize(st));
  }
}

// `count` MSMs against ONE resident base vector (different offsets / lengths / scalar vectors), three in flight at a
// time: one per MSM workspace (stream + scratch) of the current lane, so that the latency-bound bucket-reduction tail
// of one overlaps the throughput-bound kernels of the next.  Results (Jacobian) land in out_xyz_host[k * 3 * fN u64].
// handles[k]: the resident base vector of job k (G1 and G2 may be mixed); every result occupies a slot of `slot_words` 32-bit
// words in out_xyz_host (>= the Jacobian size of the job's group), its Jacobian limbs at the start of the slot
void msm_run_multi(zkp_ctx* ctx, size_t count, const uint64_t* handles, const size_t* offsets,
                   const uint64_t* const* scalars_dev, const size_t* ns, bool montgomery, uint64_t* out_xyz_host,
                   size_t slot_words) {
  if (count == 0) return;
  const size_t jw = slot_words;
  zkp_lane* L = ctx->cur;
  if (ctx->pinned_cap < count * jw * 4) {
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_cap = 0;
    ZKP_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned), count * jw * 4 + 4096));
    ctx->pinned_cap = count * jw * 4 + 4096;
  }
  constexpr int NW = zkp_lane::N_WS_MSM;                          // MSM workspaces (stream + scratch) per lane
  const int NL = std::max(1, std::min(ctx->cfg.msm_batch_lanes, (int)zkp_ctx::N_LANES));      // zkp_ctx_config.msm_batch_lanes / ZKP_BATCH_LANES
  // job k runs on lane (k / NW) % NL, workspace k % NW: up to NL * NW MSMs in flight, one hardware queue each
  auto stream_of = [&](int li, int w) { return w == 0 ? ctx->lanes[li].stream : ctx->lanes[li].ws[w].stream; };
  struct Restore {                                               // msm_run may throw: always hand the context back on lane L
    zkp_ctx* c;
    zkp_lane* cur;
    int idx;
    ~Restore() {
      c->cur = cur;
      c->cur_idx = idx;
    }
  } restore{ctx, ctx->cur, ctx->cur_idx};
  ZKP_HIP(hipEventRecord(L->ev_fork, L->stream));                  // scalars are complete on the calling lane's stream
  for (int li = 0; li < NL; li++)
    for (int w = 0; w < NW; w++)
      if (stream_of(li, w) != L->stream) ZKP_HIP(hipStreamWaitEvent(stream_of(li, w), L->ev_fork, 0));
  for (size_t k = 0; k < count; k++) {
    const int li = (int)((k / NW) % NL), w = (int)(k % NW);
    auto be = get_bases(ctx, handles[k]);
    const size_t words = 3 * (size_t)be->vt->fN;                  // 32-bit words of this job's Jacobian result
    ZKP_REQUIRE(words <= jw, ZKP_ERR_BAD_ARG);
    ZKP_REQUIRE(offsets[k] <= be->n, ZKP_ERR_BAD_ARG);
    const size_t n = std::min(ns[k], be->n - offsets[k]);        // ark min(len) truncation
    ctx->cur = &ctx->lanes[li];
    ctx->cur_idx = li;
    msm_run(ctx, handles[k], offsets[k], scalars_dev[k], n, montgomery, nullptr, nullptr, nullptr, nullptr, w);
    ZKP_HIP(hipMemcpyAsync(ctx->pinned + k * jw, ctx->lanes[li].ws[w].out.p, words * 4, hipMemcpyDeviceToHost, stream_of(li, w)));
  }
  ctx->cur = restore.cur;
  ctx->cur_idx = restore.idx;
  for (int li = 0; li < NL; li++)
    for (int w = 0; w < NW; w++) {
      if (stream_of(li, w) == L->stream) continue;
      ZKP_HIP(hipEventRecord(ctx->lanes[li].ws[w].done, stream_of(li, w)));
      ZKP_HIP(hipStreamWaitEvent(L->stream, ctx->lanes[li].ws[w].done, 0));
    }
  ZKP_HIP(hipStreamSynchronize(L->stream));
  memcpy(out_xyz_host, ctx->pinned, count * jw * 4);
}

void msm_run_batch(zkp_ctx* ctx, uint64_t handle, size_t count, const size_t* offsets, const uint64_t* const* scalars_dev,
                   const size_t* ns, bool montgomery, uint64_t* out_xyz_host) {
  if (count == 0) return;
  std::vector<uint64_t> handles(count, handle);
  msm_run_multi(ctx, count, handles.data(), offsets, scalars_dev, ns, montgomery, out_xyz_host,
                3 * (size_t)get_bases(ctx, handle)->vt->fN);
}

// True variable-base MSM (ark `VariableBaseMSM::multi_scalar_mul(bases, scalars)` / `Curve::vartime_multiscalar_mul` with
// FRESH bases, curve/src/lib.rs:38-45): nothing is precomputed and nothing stays resident.  Points and scalars are staged
// into context scratch; W windows of c bits with c * W = 256 (c = 16 from 2^12 points on, else 8).
void msm_var_run(zkp_ctx* ctx, int curve, int group, const uint64_t* xy_host, const uint8_t* inf_host,
                 const uint64_t* scalars_host, size_t n, bool montgomery, uint64_t* out_xyz_host) {
  BasesEntry e;
  e.curve = curve;
  e.group = group;
  e.vt = msm_vtbl(curve, group);
  e.n = n;
  e.var = true;
  e.owns = false;
  e.c = n >= 4096 ? 16 : 8;
  if (const char* env = getenv("ZKP_MSM_VAR_C")) {
    int c = atoi(env);
    if (c == 4 || c == 8 || c == 16) e.c = c;
  }
  e.W = 256 / e.c;
  e.wide = e.W;                // equal widths: the variable-base reduction weighs window w by 2^(c*w)
  while ((1 << e.lgk) < e.W) e.lgk++;          // one bucket set per window: the single "table copy" is the points themselves
  ZKP_REQUIRE((double)n * e.W < 2147483000.0, ZKP_ERR_BAD_ARG);
  hipStream_t st = ctx->cur->stream;
  const size_t ab = e.vt->aff_bytes;
  uint64_t* sdev = nullptr;
  if (n) {
    char* buf = reinterpret_cast<char*>(ctx->var_bases.get(n * ab + n + 64));
    e.table = buf;
    ZKP_HIP(hipMemcpyAsync(e.table, xy_host, n * ab, hipMemcpyHostToDevice, st));
    if (inf_host) {
      e.inf = reinterpret_cast<uint8_t*>(buf + n * ab);
      ZKP_HIP(hipMemcpyAsync(e.inf, inf_host, n, hipMemcpyHostToDevice, st));
    }
    sdev = ctx->msm_scalars.as<uint64_t>(n * 4);
    ZKP_HIP(hipMemcpyAsync(sdev, scalars_host, n * 32, hipMemcpyHostToDevice, st));
  }
  msm_run_entry(ctx, &e, 0, sdev, n, montgomery, out_xyz_host, nullptr, nullptr, nullptr, 0, -1, nullptr);
}

void point_fold(zkp_ctx* ctx, int curve, int group, const uint64_t* xyz, size_t k, uint64_t* out) {
  const MsmVtbl* vt = msm_vtbl(curve, group);
  size_t words = 3 * (size_t)vt->fN;
  uint32_t* d = ctx->msm_misc.as<uint32_t>((k + 1) * words);
  if (k) ZKP_HIP(hipMemcpyAsync(d + words, xyz, k * words * 4, hipMemcpyHostToDevice, ctx->cur->stream));
  vt->fold(ctx->cur->stream, d + words, (int)k, d);
  ZKP_HIP(hipGetLastError());
  ZKP_HIP(hipMemcpyAsync(out, d, words * 4, hipMemcpyDeviceToHost, ctx->cur->stream));
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
}

void point_into_affine(zkp_ctx* ctx, int curve, int group, const uint64_t* xyz, uint64_t* xy_out, uint8_t* inf_out) {
  const MsmVtbl* vt = msm_vtbl(curve, group);
  size_t jw = 3 * (size_t)vt->fN, aw = 2 * (size_t)vt->fN;
  uint32_t* d = ctx->msm_misc.as<uint32_t>(jw + aw + 4);
  ZKP_HIP(hipMemcpyAsync(d, xyz, jw * 4, hipMemcpyHostToDevice, ctx->cur->stream));
  vt->into_affine(ctx->cur->stream, d, d + jw, d + jw + aw);
  ZKP_HIP(hipGetLastError());
  uint32_t flag = 0;
  ZKP_HIP(hipMemcpyAsync(xy_out, d + jw, aw * 4, hipMemcpyDeviceToHost, ctx->cur->stream));
  ZKP_HIP(hipMemcpyAsync(&flag, d + jw + aw, 4, hipMemcpyDeviceToHost, ctx->cur->stream));
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  *inf_out = (uint8_t)flag;
}

void points_fold_into_affine(zkp_ctx* ctx, int curve, int group, const uint64_t* a, const uint64_t* b, const uint8_t* has_b,
                             size_t k, uint64_t* xy_out, uint8_t* inf_out) {
  if (k == 0) return;
  const MsmVtbl* vt = msm_vtbl(curve, group);
  const size_t jw = 3 * (size_t)vt->fN, aw = 2 * (size_t)vt->fN;
  uint32_t* d = ctx->msm_misc.as<uint32_t>(k * (2 * jw + aw + 2) + 16);
  uint32_t* da = d;
  uint32_t* db = da + k * jw;
  uint32_t* dh = db + k * jw;
  uint32_t* dxy = dh + k;
  uint32_t* dinf = dxy + k * aw;
  hipStream_t st = ctx->cur->stream;
  std::vector<uint32_t> hb(k);
  for (size_t i = 0; i < k; i++) hb[i] = has_b && has_b[i] ? 1u : 0u;
  ZKP_HIP(hipMemcpyAsync(da, a, k * jw * 4, hipMemcpyHostToDevice, st));
  if (b) ZKP_HIP(hipMemcpyAsync(db, b, k * jw * 4, hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemcpyAsync(dh, hb.data(), k * 4, hipMemcpyHostToDevice, st));
  vt->fold_affine_batch(st, da, db, dh, (int)k, dxy, dinf);
  ZKP_HIP(hipGetLastError());
  std::vector<uint32_t> hinf(k);
  ZKP_HIP(hipMemcpyAsync(xy_out, dxy, k * aw * 4, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipMemcpyAsync(hinf.data(), dinf, k * 4, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
  for (size_t i = 0; i < k; i++) inf_out[i] = (uint8_t)hinf[i];
}

// ark-serialize compressed points (host bytes) -> affine Montgomery points (host), decompressed on the device.  Returns 0, or
// 1 + the index of the first malformed point.
size_t points_decompress(zkp_ctx* ctx, int curve, int group, const uint8_t* bytes, size_t n, uint64_t* xy_out, uint8_t* inf_out) {
  if (n == 0) return 0;
  const MsmVtbl* vt = msm_vtbl(curve, group);
  const size_t pb = (size_t)vt->fN * 4 * (group == 2 ? 1 : 1), ab = vt->aff_bytes;     // compressed bytes per point = one coordinate
  char* buf = reinterpret_cast<char*>(ctx->msm_misc.get(n * pb + n * ab + n + 256 + 64));
  uint32_t* d_bytes = reinterpret_cast<uint32_t*>(buf);
  char* d_xy = buf + ((n * pb + 63) & ~(size_t)63);
  uint8_t* d_inf = reinterpret_cast<uint8_t*>(d_xy + n * ab);
  uint32_t* d_b = reinterpret_cast<uint32_t*>(buf + ((n * pb + 63) & ~(size_t)63) + n * ab + ((n + 63) & ~(size_t)63));
  uint32_t* d_status = d_b + 48;
  hipStream_t st = ctx->cur->stream;
  const uint32_t none = 0xffffffffu;
  ZKP_HIP(hipMemcpyAsync(d_bytes, bytes, n * pb, hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemcpyAsync(d_status, &none, 4, hipMemcpyHostToDevice, st));
  vt->decompress(st, d_bytes, n, d_b, d_xy, d_inf, d_status);
  ZKP_HIP(hipGetLastError());
  uint32_t status = 0;
  ZKP_HIP(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
  if (status != none) return (size_t)status;           // failing lanes wrote nothing: the outputs stay untouched
  ZKP_HIP(hipMemcpyAsync(xy_out, d_xy, n * ab, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipMemcpyAsync(inf_out, d_inf, n, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
  return 0;
}
// 0 = every point is on the curve and in the prime-order subgroup, else 1 + index of the first that is not
size_t points_subgroup_check(zkp_ctx* ctx, int curve, int group, const uint64_t* xy, const uint8_t* inf, size_t n) {
  if (n == 0) return 0;
  const MsmVtbl* vt = msm_vtbl(curve, group);
  const size_t ab = vt->aff_bytes;
  char* buf = reinterpret_cast<char*>(ctx->msm_misc.get(n * ab + n + 512));
  char* d_xy = buf;
  uint8_t* d_inf = reinterpret_cast<uint8_t*>(buf + n * ab);
  uint32_t* d_b = reinterpret_cast<uint32_t*>(buf + n * ab + ((n + 63) & ~(size_t)63));
  uint32_t* d_status = d_b + 48;
  hipStream_t st = ctx->cur->stream;
  const uint32_t none = 0xffffffffu;
  ZKP_HIP(hipMemcpyAsync(d_xy, xy, n * ab, hipMemcpyHostToDevice, st));
  if (inf) ZKP_HIP(hipMemcpyAsync(d_inf, inf, n, hipMemcpyHostToDevice, st));
  ZKP_HIP(hipMemcpyAsync(d_status, &none, 4, hipMemcpyHostToDevice, st));
  vt->subgroup_check(st, d_xy, inf ? d_inf : nullptr, n, d_b, d_status);
  ZKP_HIP(hipGetLastError());
  uint32_t status = 0;
  ZKP_HIP(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
  return status == none ? 0 : (size_t)status;
}
void points_compress(zkp_ctx* ctx, int curve, int group, const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* bytes_out) {
  if (n == 0) return;
  const MsmVtbl* vt = msm_vtbl(curve, group);
  const size_t pb = (size_t)vt->fN * 4, ab = vt->aff_bytes;
  char* buf = reinterpret_cast<char*>(ctx->msm_misc.get(n * pb + n * ab + n + 256));
  char* d_xy = buf;
  uint32_t* d_bytes = reinterpret_cast<uint32_t*>(buf + n * ab);
  uint8_t* d_inf = reinterpret_cast<uint8_t*>(buf + n * ab + n * pb);
  hipStream_t st = ctx->cur->stream;
  ZKP_HIP(hipMemcpyAsync(d_xy, xy, n * ab, hipMemcpyHostToDevice, st));
  if (inf) ZKP_HIP(hipMemcpyAsync(d_inf, inf, n, hipMemcpyHostToDevice, st));
  vt->compress(st, d_xy, inf ? d_inf : nullptr, n, d_bytes);
  ZKP_HIP(hipGetLastError());
  ZKP_HIP(hipMemcpyAsync(bytes_out, d_bytes, n * pb, hipMemcpyDeviceToHost, st));
  ZKP_HIP(hipStreamSynchronize(st));
}

void fixed_base_mul(zkp_ctx* ctx, int curve, int group, const uint64_t* base_xy, const uint64_t* scalars, size_t n,
                    uint64_t* out_xy, uint8_t* out_inf) {
  if (n == 0) return;
  const MsmVtbl* vt = msm_vtbl(curve, group);
  size_t ab = vt->aff_bytes;
  char* buf = reinterpret_cast<char*>(ctx->msm_misc.get(ab + n * 32 + n * ab + n + 64));
  char* d_base = buf;
  char* d_sc = d_base + ab;
  char* d_out = d_sc + n * 32;
  uint8_t* d_inf = reinterpret_cast<uint8_t*>(d_out + n * ab);
  ZKP_HIP(hipMemcpyAsync(d_base, base_xy, ab, hipMemcpyHostToDevice, ctx->cur->stream));
  ZKP_HIP(hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, ctx->cur->stream));
  vt->fixed_base(ctx->cur->stream, reinterpret_cast<const uint32_t*>(d_base), reinterpret_cast<const uint32_t*>(d_sc), n,
                 d_out, d_inf);
  ZKP_HIP(hipGetLastError());
  ZKP_HIP(hipMemcpyAsync(out_xy, d_out, n * ab, hipMemcpyDeviceToHost, ctx->cur->stream));
  ZKP_HIP(hipMemcpyAsync(out_inf, d_inf, n, hipMemcpyDeviceToHost, ctx->cur->stream));
  ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
}

}  // namespace zkp
