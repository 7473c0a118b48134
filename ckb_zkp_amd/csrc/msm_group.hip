// Per-configuration MSM kernels (everything except the hot bucket-accumulate loop, which lives in
// msm_acc.hip): table pre-computation, log-depth bucket reduction, point utilities, Groth16 assembly.
// Compiled once per (curve, group): -DZKP_CFG_CURVE={0,1} -DZKP_CFG_GROUP={1,2}.  See msm.hip for the design.
#include "bucket_dev.hpp"
#include "coop_dev.hpp"
#include "ec_dev.hpp"
#include "msm_vtbl.hpp"

#ifndef ZKP_CFG_CURVE
#error "compile with -DZKP_CFG_CURVE=0|1 -DZKP_CFG_GROUP=1|2"
#endif

namespace zkp {

#if ZKP_CFG_CURVE == 0
using CfgFq = Bn254Fq;
using CfgFr = Bn254Fr;
constexpr int CFG_BITS = 254;
#else
using CfgFq = Bls381Fq;
using CfgFr = Bls381Fr;
constexpr int CFG_BITS = 255;
#endif
#if ZKP_CFG_GROUP == 1
using CfgF = Fp<CfgFq>;
#else
using CfgF = Fp2<CfgFq>;
#endif
#define ZKP_CAT3(a, b, c) a##b##c
#define ZKP_SYM(name, cu, gr) ZKP_CAT3(name, cu, gr)
#define ZKP_CFG_SYM(name) ZKP_SYM(name##_c, ZKP_CFG_CURVE, ZKP_CFG_GROUP)

// non-template kernels get a per-configuration namespace (the same source is compiled four times)
namespace ZKP_CFG_SYM(cfg) {}
using namespace ZKP_CFG_SYM(cfg);

template <class F>
__global__ void ingest_kernel(char* table, const uint8_t* inf, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (inf && inf[i]) Affine<F>::inf().store(table + i * Affine<F>::BYTES);
}

template <class F>
__global__ __launch_bounds__(128) void precompute_kernel(char* table, size_t n, int c, int W, int wide) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = Affine<F>::load(table + i * Affine<F>::BYTES);
  XYZZ<F> acc = XYZZ<F>::from_affine(p);
  for (int w = 1; w < W; w++) {
    const int cw = w - 1 < wide ? c : c - 1;               // width of window w - 1 (balanced windows, msm.hip)
    for (int k = 0; k < cw; k++) acc = acc.dbl();
    Affine<F> a = acc.to_affine();
    a.store(table + ((size_t)w * n + i) * Affine<F>::BYTES);
    acc = XYZZ<F>::from_affine(a);
  }
}

// ------------------------------------------------------------------------------------------- K8 reduce
template <class F>
__global__ __launch_bounds__(256) void pair_kernel(const char* __restrict__ in, char* __restrict__ out,
                                                   uint32_t count) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound tail: win issue arbitration against co-resident accumulate waves
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  BkPoint<F>::add_mem(in + (size_t)(2 * k) * BkPoint<F>::BYTES, in + (size_t)(2 * k + 1) * BkPoint<F>::BYTES,
                      out + (size_t)k * BkPoint<F>::BYTES);
}

// The top of the pyramid in ONE launch: the level with `cnt` <= PAIR_TOP_MAX entries at `base`, every further level
// directly behind its predecessor (the layout msm.hip uses), down to the root.  One workgroup, a barrier per level: the ~10
// levels that hold fewer points than the machine has SIMDs cost one dependent addition each either way, but as one packet in
// the hardware queue instead of ten (streams that share the queue wait behind every packet).
constexpr int PAIR_TOP_THREADS = 512;
template <class F>
__device__ __forceinline__ void pair_top_body(char* __restrict__ base, uint32_t cnt) {
  char* in = base;
  for (; cnt > 1; cnt >>= 1) {
    char* out = in + (size_t)cnt * BkPoint<F>::BYTES;
    bool coop = false;
    if constexpr (QuadCoop<F>::ON) coop = cnt / 2 <= PAIR_TOP_THREADS / 4;       // fewer additions than quads: four lanes per addition
    if constexpr (QuadCoop<F>::ON) {
      if (coop) {
        const uint32_t k = threadIdx.x >> 2;
        if (k < cnt / 2)
          quad_add_any<F>(in + (size_t)(2 * k) * BkPoint<F>::BYTES, in + (size_t)(2 * k + 1) * BkPoint<F>::BYTES,
                          out + (size_t)k * BkPoint<F>::BYTES, threadIdx.x & 3);
      }
    }
    if (!coop)
      for (uint32_t k = threadIdx.x; k < cnt / 2; k += PAIR_TOP_THREADS)
        BkPoint<F>::add_mem(in + (size_t)(2 * k) * BkPoint<F>::BYTES, in + (size_t)(2 * k + 1) * BkPoint<F>::BYTES,
                            out + (size_t)k * BkPoint<F>::BYTES);
    __threadfence_block();
    __syncthreads();
    in = out;
  }
}
template <class F>
__global__ __launch_bounds__(PAIR_TOP_THREADS) void pair_top_kernel(char* __restrict__ base, uint32_t cnt) {
  __builtin_amdgcn_s_setprio(3);
  pair_top_body<F>(base, cnt);
}

// One block of a segmented sum: block -> (segment l, chunk); sums <= plan.chunk entries into partial[block].  Segments l >= l_hi read
// from base_hi instead of base (round 6: the second stage sums the partials of the low levels and, directly, the odd entries of the
// levels the fused top produced).  Written for workgroups of >= 256 threads: threads >= 256 only take part in the barriers.
template <class F>
__device__ __forceinline__ void segsum_body(const char* __restrict__ base, const SegPlan& plan, char* __restrict__ partial, uint32_t block,
                                            const char* __restrict__ base_hi, int l_hi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t t = threadIdx.x;
  const bool act = t < 256;
  int l = 0;
  while (l + 1 < plan.L && block >= plan.first_block[l + 1]) l++;
  if (l >= l_hi) base = base_hi;
  uint32_t chunk = block - plan.first_block[l];
  uint32_t lo = chunk * plan.chunk;
  uint32_t hi = min(plan.count[l], lo + plan.chunk);
  if constexpr (BkPoint<F>::MEM_ADD) {
    // the running sum lives in this thread's LDS slot and every addition streams its operands (BkPoint::add_mem)
    char* my = smem + (act ? t : 0) * BkPoint<F>::BYTES;
    if (act) {
      BkPoint<F>::inf().store(my);
      for (uint32_t i = lo + t; i < hi; i += 256)
        BkPoint<F>::add_mem(my, base + ((size_t)plan.off[l] + (size_t)i * plan.stride[l]) * BkPoint<F>::BYTES, my);
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      bool coop = false;
      if constexpr (QuadCoop<F>::ON) {
        coop = s <= 64;                                                         // 64 quads: four lanes per addition from here on
        if (coop && (int)(t >> 2) < s) {
          char* mine = smem + (t >> 2) * BkPoint<F>::BYTES;
          quad_add_any<F>(mine, smem + ((t >> 2) + s) * BkPoint<F>::BYTES, mine, t & 3);
        }
      }
      if (!coop && (int)t < s) BkPoint<F>::add_mem(my, smem + (t + s) * BkPoint<F>::BYTES, my);
      __syncthreads();
    }
    if (t == 0) BkPoint<F>::copy_point(partial + (size_t)block * BkPoint<F>::BYTES, my);
    return;
  }
  BkPoint<F> acc = BkPoint<F>::inf();
  if (act) {
    for (uint32_t i = lo + t; i < hi; i += 256)
      acc.add(BkPoint<F>::load(base + ((size_t)plan.off[l] + (size_t)i * plan.stride[l]) * BkPoint<F>::BYTES));
    // LDS tree
    acc.store(smem + t * BkPoint<F>::BYTES);
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)t < s) {
      BkPoint<F> o = BkPoint<F>::load(smem + (t + s) * BkPoint<F>::BYTES);
      if (!o.is_inf() || !acc.is_inf()) {
        acc.add(o);
        acc.store(smem + t * BkPoint<F>::BYTES);
      }
    }
    __syncthreads();
  }
  if (t == 0) acc.store(partial + (size_t)block * BkPoint<F>::BYTES);
}
template <class F>
__global__ __launch_bounds__(256) void segsum_kernel(const char* __restrict__ base, SegPlan plan, char* __restrict__ partial,
                                                     const char* __restrict__ base_hi, int l_hi) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound tail: win issue arbitration against co-resident accumulate waves
  segsum_body<F>(base, plan, partial, blockIdx.x, base_hi, l_hi);
}
// Round 6: the top of the pyramid (ONE workgroup, ~85 us of dependent additions) and the segmented sums of every level below it
// (hundreds of workgroups, ~105 us) depend on disjoint data — in ONE launch they run side by side instead of one after the other:
// block 0 = pair_top, blocks 1.. = segment-sum blocks of the levels the plain pair launches produced.
template <class F>
__global__ __launch_bounds__(PAIR_TOP_THREADS) void pair_top_segsum_kernel(char* __restrict__ top_base, uint32_t top_cnt,
                                                                           const char* __restrict__ base, SegPlan plan,
                                                                           char* __restrict__ partial) {
  __builtin_amdgcn_s_setprio(3);
  if (blockIdx.x == 0) {
    pair_top_body<F>(top_base, top_cnt);
    return;
  }
  segsum_body<F>(base, plan, partial, blockIdx.x - 1, nullptr, 1 << 30);
}

// Buckets split into up to COMBINE_SMALL tasks are folded by ONE lane each (64 buckets per wave-addition); only genuinely long
// buckets (skewed scalars, or MSMs far beyond 2^24 points) get a whole wave, whose LDS tree spends ~4 wave-additions on the
// ten additions of an 11-task bucket.  At 2^24 with balanced windows half of the 2^19 buckets hold 11 tasks: with the
// threshold at 8 they all took the wave path and the G1 MSMs lost 7-9 ms each.
constexpr uint32_t COMBINE_SMALL = 32;
template <class F>
__global__ __launch_bounds__(256) void combine_small_kernel(const uint32_t* __restrict__ long_list,
                                                            const uint32_t* __restrict__ n_long_dev,
                                                            const uint32_t* __restrict__ toff,
                                                            const char* __restrict__ partial,
                                                            char* __restrict__ buckets, uint32_t init) {
  __builtin_amdgcn_s_setprio(3);
  const uint32_t n_long = *n_long_dev;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_long; w += gridDim.x * blockDim.x) {
    const uint32_t b = long_list[w];
    const uint32_t p0 = toff[b], p1 = toff[b + 1];
    if (p1 - p0 > COMBINE_SMALL) continue;
    BkPoint<F> acc = BkPoint<F>::load(partial + (size_t)p0 * BkPoint<F>::BYTES);
    for (uint32_t i = p0 + 1; i < p1; i++) acc.add(BkPoint<F>::load(partial + (size_t)i * BkPoint<F>::BYTES));
    if (init) acc.add(BkPoint<F>::load(buckets + (size_t)b * BkPoint<F>::BYTES));      // bucket chaining
    acc.store(buckets + (size_t)b * BkPoint<F>::BYTES);
  }
}

// buckets[b] = sum of the partial sums of a bucket that was split into several tasks; one wave per bucket.
// The bucket walk is kept in SGPRs (readfirstlane): the loop and the "few tasks" skip are then scalar branches, and the only
// exec-masked regions are the two `if (lane ...)` bodies.  With the walk in VGPRs, hipcc 7.2 merged the exec restore after
// `if (lane == 0)` into the one of the enclosing per-lane `if` and then placed a register reload in between — executed by
// lane 0 only, so the other lanes entered the next bucket with a clobbered index register (a memory fault as soon as one wave
// combined two buckets; tests/test_gpu_msm.py::test_msm_many_long_buckets pins it).
template <class F>
__global__ __launch_bounds__(256) void combine_kernel(const uint32_t* __restrict__ long_list,
                                                      const uint32_t* __restrict__ n_long_dev,
                                                      const uint32_t* __restrict__ toff,
                                                      const char* __restrict__ partial, char* __restrict__ buckets, uint32_t init) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  char* my = smem + (size_t)wv * 64 * BkPoint<F>::BYTES;
  const uint32_t n_long = *n_long_dev;
  for (uint32_t w = blockIdx.x * nw + wv; w < n_long; w += gridDim.x * nw) {
    const uint32_t b = __builtin_amdgcn_readfirstlane(long_list[w]);
    const uint32_t p0 = __builtin_amdgcn_readfirstlane(toff[b]), p1 = __builtin_amdgcn_readfirstlane(toff[b + 1]);
    if (p1 - p0 <= COMBINE_SMALL) continue;            // handled by combine_small_kernel
    BkPoint<F> acc = BkPoint<F>::inf();
    for (uint32_t i = p0 + lane; i < p1; i += 64) acc.add(BkPoint<F>::load(partial + (size_t)i * BkPoint<F>::BYTES));
    acc.store(my + lane * BkPoint<F>::BYTES);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int s = 32; s > 0; s >>= 1) {
      __builtin_amdgcn_wave_barrier();
      if (lane < s) {
        BkPoint<F> o = BkPoint<F>::load(my + (lane + s) * BkPoint<F>::BYTES);
        if (!o.is_inf()) {
          acc.add(o);
          acc.store(my + lane * BkPoint<F>::BYTES);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0) {
      if (init) acc.add(BkPoint<F>::load(buckets + (size_t)b * BkPoint<F>::BYTES));    // bucket chaining
      acc.store(buckets + (size_t)b * BkPoint<F>::BYTES);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// out = sum_l 2^l * O[l] + root ; one wave
template <class F>
__global__ __launch_bounds__(128) void final_kernel(const char* __restrict__ O, int L, const char* __restrict__ root,
                                                   char* __restrict__ out_xyzz, uint32_t* __restrict__ out_jac) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound tail: win issue arbitration against co-resident accumulate waves
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int t = threadIdx.x;
  if constexpr (QuadCoop<F>::ON && QuadCoop<F>::GROUP == 1) {
    // G1 (round 4): 128 lanes = 32 quads, quad q owns point q (levels 0..L-1, the root at L): q cooperative doublings, then a tree
    // of cooperative additions — 18 x 3 + 5 x 5.5 product latencies instead of 18 x 9.75 + 6 x 14.5
    const int q = t >> 2, role = t & 3;
    char* my = smem + q * BkPoint<F>::BYTES;
    if (role == 0) {
      if (q < L) BkPoint<F>::copy_point(my, O + (size_t)q * BkPoint<F>::BYTES);
      else if (q == L) BkPoint<F>::copy_point(my, root);
      else BkPoint<F>::inf().store(my);
    }
    __syncthreads();
    if (q < L)
      for (int k = 0; k < q; k++) quad_dbl_mem<typename QuadCoop<F>::P>(my, my, role);
    __syncthreads();
    for (int s = 16; s > 0; s >>= 1) {
      if (q < s) quad_add_mem<typename QuadCoop<F>::P>(my, smem + (q + s) * BkPoint<F>::BYTES, my, role);
      __syncthreads();
    }
    if (t == 0) {
      const XYZZ<F> res = BkPoint<F>::load(my).to_sat();
      if (out_xyzz) res.store(out_xyzz);
      if (out_jac) res.store_jacobian(out_jac);
    }
    return;
  }
  if constexpr (BkPoint<F>::MEM_ADD) {
    // G2: the lane's point lives in its LDS slot, additions / doublings stream their operands (bucket_dev.hpp)
    char* my = smem + t * BkPoint<F>::BYTES;
    if (t < L) BkPoint<F>::copy_point(my, O + (size_t)t * BkPoint<F>::BYTES);
    else if (t == L) BkPoint<F>::copy_point(my, root);
    else BkPoint<F>::inf().store(my);
    if (t < L && !BkPoint<F>::load(my).is_inf())
      for (int k = 0; k < t; k++) BkPoint<F>::dbl_mem(my, my);
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
      bool coop = false;
      if constexpr (QuadCoop<F>::ON) {
        coop = s <= 16;                                              // 16 quads
        if (coop && (t >> 2) < s) {
          char* mine = smem + (t >> 2) * BkPoint<F>::BYTES;
          quad_add_any<F>(mine, smem + ((t >> 2) + s) * BkPoint<F>::BYTES, mine, t & 3);
        }
      }
      if (!coop && t < s) BkPoint<F>::add_mem(my, smem + (t + s) * BkPoint<F>::BYTES, my);
      __syncthreads();
    }
    if (t == 0) {
      const XYZZ<F> res = BkPoint<F>::load(my).to_sat();
      if (out_xyzz) res.store(out_xyzz);
      if (out_jac) res.store_jacobian(out_jac);
    }
    return;
  }
  BkPoint<F> acc = BkPoint<F>::inf();
  if (t < L) {
    acc = BkPoint<F>::load(O + (size_t)t * BkPoint<F>::BYTES);
    for (int k = 0; k < t; k++) acc = acc.dbl();
  } else if (t == L) {
    acc = BkPoint<F>::load(root);
  }
  acc.store(smem + t * BkPoint<F>::BYTES);
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) {
    if (t < s) {
      BkPoint<F> o = BkPoint<F>::load(smem + (t + s) * BkPoint<F>::BYTES);
      if (!o.is_inf() || !acc.is_inf()) {
        acc.add(o);
        acc.store(smem + t * BkPoint<F>::BYTES);
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    const XYZZ<F> res = acc.to_sat();                      // the pyramid lives in the unsaturated layout; results do not
    if (out_xyzz) res.store(out_xyzz);
    if (out_jac) res.store_jacobian(out_jac);
  }
}

// descriptor-driven segmented sum (variable-base reduction): one workgroup per descriptor
template <class F>
__global__ __launch_bounds__(256) void segsum_desc_kernel(const char* __restrict__ base, const SegDesc* __restrict__ descs,
                                                          char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SegDesc d = descs[blockIdx.x];
  BkPoint<F> acc = BkPoint<F>::inf();
  for (uint32_t i = threadIdx.x; i < d.count; i += 256)
    acc.add(BkPoint<F>::load(base + ((size_t)d.off + (size_t)i * d.stride) * BkPoint<F>::BYTES));
  acc.store(smem + threadIdx.x * BkPoint<F>::BYTES);
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      BkPoint<F> o = BkPoint<F>::load(smem + (threadIdx.x + s) * BkPoint<F>::BYTES);
      if (!o.is_inf() || !acc.is_inf()) {
        acc.add(o);
        acc.store(smem + threadIdx.x * BkPoint<F>::BYTES);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) acc.store(out + (size_t)d.out * BkPoint<F>::BYTES);
}

// out = sum_{t < 256} 2^t R[t] + sum_w 2^(c w) roots[w]: lane t adds the window root where t = c*w, doubles t times,
// then an LDS tree over the 256 lanes.  The <= 255-doubling chain of the top lane is the latency of a variable-base MSM
// (fixed-base window tables exist precisely to avoid it).
template <class F>
__global__ __launch_bounds__(256) void final_var_kernel(const char* __restrict__ R, const char* __restrict__ roots, int c,
                                                        int W, char* __restrict__ out_xyzz, uint32_t* __restrict__ out_jac) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  BkPoint<F> acc = BkPoint<F>::load(R + (size_t)t * BkPoint<F>::BYTES);
  if (t % c == 0 && t / c < W) acc.add(BkPoint<F>::load(roots + (size_t)(t / c) * BkPoint<F>::BYTES));
  if (!acc.is_inf())
    for (int k = 0; k < t; k++) acc = acc.dbl();
  acc.store(smem + t * BkPoint<F>::BYTES);
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      BkPoint<F> o = BkPoint<F>::load(smem + (t + s) * BkPoint<F>::BYTES);
      if (!o.is_inf() || !acc.is_inf()) {
        acc.add(o);
        acc.store(smem + t * BkPoint<F>::BYTES);
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    const XYZZ<F> res = acc.to_sat();                      // the pyramid lives in the unsaturated layout; results do not
    if (out_xyzz) res.store(out_xyzz);
    if (out_jac) res.store_jacobian(out_jac);
  }
}

template <class F>
__global__ void write_identity_kernel(char* out_xyzz, uint32_t* out_jac) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    XYZZ<F> z = XYZZ<F>::inf();
    if (out_xyzz) z.store(out_xyzz);
    if (out_jac) z.store_jacobian(out_jac);
  }
}

// ------------------------------------------------------------------------------------------- small point utilities
template <class F>
__device__ XYZZ<F> jac_to_xyzz(const uint32_t* p) {
  F X = F::load(p), Y = F::load(p + F::N), Z = F::load(p + 2 * F::N);
  if (Z.is_zero()) return XYZZ<F>::inf();
  F zz = Z.sqr();
  return {X, Y, zz, zz * Z};
}

template <class F>
__global__ void fold_kernel(const uint32_t* pts, int k, uint32_t* out_jac) {
  if (threadIdx.x || blockIdx.x) return;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 0; i < k; i++) acc.add(jac_to_xyzz<F>(pts + (size_t)i * 3 * F::N));
  acc.store_jacobian(out_jac);
}

// Multi-GPU exchange step: `gathered` holds, per rank, the 5 partial MSM results of a base-sharded
// Groth16 proof as XYZZ slots (A | B1 | B2 | H | L, `slot` bytes each, rank stride `rank_stride`) exactly as the
// all-gather delivered them; lane t sums slot t over the ranks into res.  mask selects the slots of this group.
template <class F>
__global__ void fold_slots_kernel(const char* __restrict__ gathered, size_t rank_stride, int world, size_t slot,
                                  uint32_t mask, char* __restrict__ res) {
  const int t = threadIdx.x;
  if (blockIdx.x || t >= 5 || !((mask >> t) & 1)) return;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int k = 0; k < world; k++) acc.add(XYZZ<F>::load(gathered + (size_t)k * rank_stride + (size_t)t * slot));
  acc.store(res + (size_t)t * slot);
}

template <class F>
__global__ void into_affine_kernel(const uint32_t* jac, uint32_t* xy, uint32_t* inf) {
  if (threadIdx.x || blockIdx.x) return;
  XYZZ<F> p = jac_to_xyzz<F>(jac);
  Affine<F> a = p.to_affine();
  a.store(xy);
  *inf = p.is_inf() ? 1 : 0;
}

template <class F>
__global__ void fold_affine_batch_kernel(const uint32_t* __restrict__ ja, const uint32_t* __restrict__ jb,
                                         const uint32_t* __restrict__ has_b, int k, uint32_t* __restrict__ xy,
                                         uint32_t* __restrict__ inf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  XYZZ<F> p = jac_to_xyzz<F>(ja + (size_t)i * 3 * F::N);
  if (has_b[i]) p.add(jac_to_xyzz<F>(jb + (size_t)i * 3 * F::N));
  Affine<F> a = p.to_affine();
  a.store(xy + (size_t)i * 2 * F::N);
  inf[i] = p.is_inf() ? 1 : 0;
}

template <class F>
__global__ void from_jacobian_kernel(const uint32_t* jac, char* out) {
  if (threadIdx.x || blockIdx.x) return;
  jac_to_xyzz<F>(jac).store(out);
}

// ------------------------------------------------------------------------------------------- ark-serialize point codec
// Compressed short-Weierstrass points of ark-serialize 0.2 (`Parameters::serialize` as cli/src/setup.rs:41-45 writes it,
// `Proof::serialize` of cli/src/zkp_prove.rs:45-49): the canonical little-endian x (Fq2: c0 then c1) with two flags in the top
// bits of the LAST byte — bit 7: y is the larger of {y, -y} (Fq: as integers; Fq2: c1 first, then c0), bit 6: the identity.
// The same layout ckb_zkp_amd/serialize.py restates (parity unpinned: the reference holds no serialized fixture).
// Decompression of a multi-million-point key is a square root per point: one lane per point.
template <class P>
__device__ __forceinline__ bool fp_sqrt(const Fp<P>& a, Fp<P>& out) {  // p = 3 (mod 4) for both base fields: a^((p + 1) / 4)
  constexpr int N = P::N;
  uint32_t e[N];
  uint64_t c = 1;
#pragma unroll
  for (int i = 0; i < N; i++) {                                         // p + 1
    c += (uint64_t)P::MOD[i];
    e[i] = (uint32_t)c;
    c >>= 32;
  }
#pragma unroll
  for (int i = 0; i < N; i++) e[i] = (e[i] >> 2) | (i + 1 < N ? e[i + 1] << 30 : 0);
  out = a.pow_limbs(e);
  return out.sqr() == a;
}
template <class P>
__device__ __forceinline__ bool fp_gt(const Fp<P>& a, const Fp<P>& b) {   // canonical integers, a > b
  for (int i = P::N - 1; i >= 0; i--)
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
  return false;
}
template <class P>
__device__ __forceinline__ bool fp_canonical_lt_mod(const Fp<P>& a) {  // a (plain words) < p
  for (int i = P::N - 1; i >= 0; i--)
    if (a.v[i] != P::MOD[i]) return a.v[i] < P::MOD[i];
  return false;
}
// y "positive" = y > -y in ark's ordering
template <class P>
__device__ __forceinline__ bool y_positive(const Fp<P>& y) { return fp_gt(y.from_mont(), y.neg().from_mont()); }
template <class P>
__device__ __forceinline__ bool y_positive(const Fp2<P>& y) {
  const Fp<P> a1 = y.c1.from_mont(), n1 = y.c1.neg().from_mont();
  if (!(a1 == n1)) return fp_gt(a1, n1);
  return fp_gt(y.c0.from_mont(), y.c0.neg().from_mont());
}
template <class P>
__device__ __forceinline__ bool curve_sqrt(const Fp<P>& a, Fp<P>& out) { return fp_sqrt(a, out); }
template <class P>
__device__ __forceinline__ bool curve_sqrt(const Fp2<P>& a, Fp2<P>& out) {   // norm method, as serialize.py _sqrt_fq2
  using B = Fp<P>;
  if (a.c1.is_zero()) {
    B s;
    if (fp_sqrt(a.c0, s)) {
      out = {s, B::zero()};
      return true;
    }
    if (fp_sqrt(a.c0.neg(), s)) {
      out = {B::zero(), s};
      return true;
    }
    return false;
  }
  B n;
  if (!fp_sqrt(a.c0.sqr() + a.c1.sqr(), n)) return false;
  const B inv2 = (B::one() + B::one()).inv();
  for (int sign = 0; sign < 2; sign++) {
    const B x2 = (sign == 0 ? a.c0 + n : a.c0 - n) * inv2;
    B x;
    if (fp_sqrt(x2, x) && !x.is_zero()) {
      const B y = a.c1 * (x + x).inv();
      const Fp2<P> r{x, y};
      if (r.sqr() == a) {
        out = r;
        return true;
      }
    }
  }
  return false;
}
template <class P>
__device__ __forceinline__ bool coords_canonical(const Fp<P>& x) { return fp_canonical_lt_mod(x); }
template <class P>
__device__ __forceinline__ bool coords_canonical(const Fp2<P>& x) { return fp_canonical_lt_mod(x.c0) && fp_canonical_lt_mod(x.c1); }
template <class P>
__device__ __forceinline__ Fp<P> coords_to_mont(const Fp<P>& x) { return x.to_mont(); }
template <class P>
__device__ __forceinline__ Fp2<P> coords_to_mont(const Fp2<P>& x) { return {x.c0.to_mont(), x.c1.to_mont()}; }
template <class P>
__device__ __forceinline__ Fp<P> coords_from_mont(const Fp<P>& x) { return x.from_mont(); }
template <class P>
__device__ __forceinline__ Fp2<P> coords_from_mont(const Fp2<P>& x) { return {x.c0.from_mont(), x.c1.from_mont()}; }

// the curve's b in Montgomery form: G1 3 / 4 (BN254 / BLS12-381), G2 twist 3 / (9 + u) = (27 - 3u) / 82 and 4 (1 + u)
template <class F>
__global__ void curve_b_kernel(uint32_t* out) {
  if (threadIdx.x || blockIdx.x) return;
  using B = Fp<CfgFq>;
  auto small = [](uint32_t k) {
    B x = B::zero();
    x.v[0] = k;
    return x.to_mont();
  };
#if ZKP_CFG_GROUP == 1
  small(ZKP_CFG_CURVE == 0 ? 3 : 4).store(out);
#else
  if (ZKP_CFG_CURVE == 0) {
    const B i82 = small(82).inv();
    (small(27) * i82).store(out);
    (small(3) * i82).neg().store(out + B::N);
  } else {
    small(4).store(out);
    small(4).store(out + B::N);
  }
#endif
}

// bytes: n points of 4 * F::N bytes each -> xy (affine Montgomery, identity = (0, 0)), inf flags; status[0] = 1 + index of the first
// malformed point (flags, x >= p, no square root), 0 if none
template <class F>
__global__ __launch_bounds__(64) void decompress_kernel(const uint32_t* __restrict__ bytes, size_t n, const uint32_t* __restrict__ bcoef,
                                                        char* __restrict__ xy, uint8_t* __restrict__ inf, uint32_t* __restrict__ status) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F x = F::load(bytes + i * F::N);
  uint32_t* top = reinterpret_cast<uint32_t*>(&x) + (F::N - 1);
  const uint32_t flags = *top >> 30;                                     // bit 1: positive y, bit 0: infinity
  *top &= 0x3fffffffu;
  auto fail = [&] { atomicMin(status, (uint32_t)(i + 1)); };
  if (flags == 3) {
    fail();
    return;
  }
  // ark-serialize reads the field element (deserialize_with_flags -> Fp::read -> from_repr) BEFORE it looks at the infinity flag:
  // x >= p is InvalidData for the identity encoding too
  if (!coords_canonical(x)) {
    fail();
    return;
  }
  if (flags & 1) {
    Affine<F>::inf().store(xy + i * Affine<F>::BYTES);
    inf[i] = 1;
    return;
  }
  const F xm = coords_to_mont(x);
  const F rhs = xm.sqr() * xm + F::load(bcoef);
  F y;
  if (!curve_sqrt(rhs, y)) {
    fail();
    return;
  }
  if (y_positive(y) != (bool)(flags >> 1)) y = y.neg();
  Affine<F>{xm, y}.store(xy + i * Affine<F>::BYTES);
  inf[i] = 0;
}
// ark-ec 0.2 `GroupAffine::is_in_correct_subgroup_assuming_on_curve` ([r]P = O by double-and-add over the bits of the group
// order) plus the curve equation — what the CHECKED `deserialize` of a key or proof element runs per point
// (/root/reference/groth16/src/lib.rs `Parameters::deserialize` -> ark-serialize derive -> GroupAffine::deserialize).  One lane per
// point; status = 1 + index of the first point that is off the curve or outside the prime-order subgroup (atomicMin).
template <class F>
__global__ __launch_bounds__(64) void subgroup_check_kernel(const char* __restrict__ xy, const uint8_t* __restrict__ inf, size_t n,
                                                           const uint32_t* __restrict__ bcoef, uint32_t* __restrict__ status) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (inf && inf[i]) return;
  const Affine<F> p = Affine<F>::load(xy + i * Affine<F>::BYTES);
  if (p.is_inf()) return;
  bool ok = (p.y.sqr() - (p.x.sqr() * p.x + F::load(bcoef))).is_zero();
  if (ok) {
    XYZZ<F> acc = XYZZ<F>::from_affine(p);
    int top = 32 * CfgFr::N - 1;
    while (!((CfgFr::MOD[top >> 5] >> (top & 31)) & 1u)) top--;
    for (int b = top - 1; b >= 0; b--) {
      acc = acc.dbl();
      if ((CfgFr::MOD[b >> 5] >> (b & 31)) & 1u) acc.madd(p);
    }
    ok = acc.is_inf();
  }
  if (!ok) atomicMin(status, (uint32_t)(i + 1));
}
template <class F>
__global__ __launch_bounds__(64) void compress_kernel(const char* __restrict__ xy, const uint8_t* __restrict__ inf, size_t n,
                                                      uint32_t* __restrict__ bytes) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<F> p = Affine<F>::load(xy + i * Affine<F>::BYTES);
  F x = F::zero();
  uint32_t flags = 1;                                                    // identity: x = 0 + the infinity flag
  if (!(inf && inf[i]) && !p.is_inf()) {
    x = coords_from_mont(p.x);
    flags = y_positive(p.y) ? 2u : 0u;
  }
  uint32_t* top = reinterpret_cast<uint32_t*>(&x) + (F::N - 1);
  *top |= flags << 30;
  x.store(bytes + i * F::N);
}

// k_i * P, one lane per scalar (setup-side helper; double-and-add, MSB first)
template <class F, int BITS>
__global__ __launch_bounds__(128) void fixed_base_kernel(const uint32_t* base, const uint32_t* scalars, size_t n,
                                                         char* out_xy, uint8_t* out_inf) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = Affine<F>::load(base);
  uint32_t s[8];
#pragma unroll
  for (int l = 0; l < 8; l++) s[l] = scalars[i * 8 + l];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int bit = BITS; bit >= 0; bit--) {
    acc = acc.dbl();
    if ((s[bit >> 5] >> (bit & 31)) & 1) acc.madd(p);
  }
  Affine<F> a = acc.to_affine();
  a.store(out_xy + i * Affine<F>::BYTES);
  out_inf[i] = acc.is_inf() ? 1 : 0;
}


// ------------------------------------------------------------------------------------------- group-element transform
// The H query in EVALUATION form (groth16.hip, round 4): H'_j = (1/N) sum_i w^(-ij) g^(-i) H_i, so that
// sum_j v_j H'_j = sum_i h_i H_i for h = coset_ifft(v) — the last transform of the witness map (r1cs_to_qap.rs:169) moves into the key.
// A radix-2 decimation-in-time transform over group elements: butterflies (u, q) -> (u + t q, u - t q) with t a scalar; one-time
// work per key (N log N / 2 scalar multiplications), plain double-and-add on the saturated formulas of ec_dev.hpp.
// Points travel through the stages in the unsaturated layout (BkPoint, bucket_dev.hpp: 205 instead of ~330 VALU instructions per
// product); scalar multiplications take two bits at a time against {q, 2q, 3q} (every lane of a wave adds in every step anyway, so
// sparse digit forms do not help; a fixed window does): 128 x (2 doublings + 1 addition) instead of 255 x (1 + 1).
#if ZKP_CFG_GROUP == 1
template <class F>
ZKP_DEV BkPoint<F> bk_select3(uint32_t d, const BkPoint<F>& a1, const BkPoint<F>& a2, const BkPoint<F>& a3) {
  using U = typename BkPoint<F>::U;
  BkPoint<F> r;
#pragma unroll
  for (int i = 0; i < U::L; i++) {
    r.v.x.f.v[i] = d == 1 ? a1.v.x.f.v[i] : d == 2 ? a2.v.x.f.v[i] : a3.v.x.f.v[i];
    r.v.y.f.v[i] = d == 1 ? a1.v.y.f.v[i] : d == 2 ? a2.v.y.f.v[i] : a3.v.y.f.v[i];
    r.v.zz.f.v[i] = d == 1 ? a1.v.zz.f.v[i] : d == 2 ? a2.v.zz.f.v[i] : a3.v.zz.f.v[i];
    r.v.zzz.f.v[i] = d == 1 ? a1.v.zzz.f.v[i] : d == 2 ? a2.v.zzz.f.v[i] : a3.v.zzz.f.v[i];
  }
  r.v.inf = d == 1 ? a1.v.inf : d == 2 ? a2.v.inf : a3.v.inf;
  return r;
}
// k * q, k: canonical scalar of <= BITS + 1 bits
template <class F, int BITS>
ZKP_DEV BkPoint<F> bk_scalar_mul(const BkPoint<F>& q, const uint32_t* k) {
  BkPoint<F> q2 = q.dbl(), q3 = q2;
  q3.add(q);
  BkPoint<F> acc = BkPoint<F>::inf();
  for (int bit = BITS | 1; bit >= 1; bit -= 2) {
    acc = acc.dbl().dbl();
    const uint32_t d = (bit < 256 ? ((k[bit >> 5] >> (bit & 31)) & 1) << 1 : 0) | ((k[(bit - 1) >> 5] >> ((bit - 1) & 31)) & 1);
    if (d) acc.add(bk_select3<F>(d, q, q2, q3));
  }
  return acc;
}
template <class F>
ZKP_DEV BkPoint<F> bk_neg(const BkPoint<F>& p) {
  BkPoint<F> r = p;
  if (!p.v.inf) r.v.y = ub_neg<4>(p.v.y);
  return r;
}
// X[bitrev(i)] = scal[i] * P_i   (scal: canonical 8-word scalars; P_i affine, identity beyond n_in)
template <class F, int BITS>
__global__ __launch_bounds__(128) void gfft_scale_kernel(const char* __restrict__ xy, const uint8_t* __restrict__ inf, size_t n_in,
                                                         const uint32_t* __restrict__ scal, uint32_t log_n, char* __restrict__ X) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  BkPoint<F> acc = BkPoint<F>::inf();
  if (i < n_in && !(inf && inf[i])) {
    uint32_t k[8];
#pragma unroll
    for (int l = 0; l < 8; l++) k[l] = scal[i * 8 + l];
    acc = bk_scalar_mul<F, BITS>(BkPoint<F>::from_sat(XYZZ<F>::from_affine(Affine<F>::load(xy + i * Affine<F>::BYTES))), k);
  }
  const size_t r = log_n ? (size_t)(__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
  acc.store(X + r * BkPoint<F>::BYTES);
}
// stage s (1-based, block size m = 2^s): X[lo], X[hi] <- X[lo] + t X[hi], X[lo] - t X[hi], t = tw[j * N / m]
template <class F, int BITS>
__global__ __launch_bounds__(128) void gfft_stage_kernel(char* __restrict__ X, const uint32_t* __restrict__ tw, uint32_t log_n,
                                                         uint32_t s) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >> (log_n - 1)) return;
  const size_t half = (size_t)1 << (s - 1), j = b & (half - 1), lo = ((b >> (s - 1)) << s) + j, hi = lo + half;
  const size_t e = j << (log_n - s);
  BkPoint<F> t = BkPoint<F>::load(X + hi * BkPoint<F>::BYTES);
  if (e != 0 && !t.is_inf()) {
    uint32_t k[8];
#pragma unroll
    for (int l = 0; l < 8; l++) k[l] = tw[e * 8 + l];
    t = bk_scalar_mul<F, BITS>(t, k);
  }
  BkPoint<F> u = BkPoint<F>::load(X + lo * BkPoint<F>::BYTES), v = u;
  u.add(t);
  v.add(bk_neg<F>(t));
  u.store(X + lo * BkPoint<F>::BYTES);
  v.store(X + hi * BkPoint<F>::BYTES);
}
template <class F>
__global__ __launch_bounds__(128) void gfft_affine_kernel(const char* __restrict__ X, size_t n, char* __restrict__ xy,
                                                          uint8_t* __restrict__ inf) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const XYZZ<F> p = BkPoint<F>::load(X + i * BkPoint<F>::BYTES).to_sat();
  p.to_affine().store(xy + i * Affine<F>::BYTES);
  inf[i] = p.is_inf() ? 1 : 0;
}
#endif

// out_m = L_m - sum_{e in column m of C} coeff_e * G_{row_e}: the C matrix folded into the L query (groth16.hip).  One lane per
// variable; kind[e]: 1 = coefficient +1, 2 = coefficient -1, 0 = coeff[e] (Montgomery, 8 words) by double-and-add.  One-time work.
#if ZKP_CFG_GROUP == 1
template <class F, int BITS>
__global__ __launch_bounds__(128) void lfold_kernel(const char* __restrict__ L_xy, const uint8_t* __restrict__ L_inf, size_t n_vars,
                                                    const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ rows,
                                                    const uint8_t* __restrict__ kind, const uint32_t* __restrict__ coeff,
                                                    const char* __restrict__ G_xy, const uint8_t* __restrict__ G_inf,
                                                    char* __restrict__ out_xy, uint8_t* __restrict__ out_inf) {
  const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_vars) return;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t e = col_ptr[m]; e < col_ptr[m + 1]; e++) {
    const uint32_t k = rows[e];
    if (G_inf[k]) continue;
    Affine<F> g = Affine<F>::load(G_xy + (size_t)k * Affine<F>::BYTES);
    if (kind[e] == 1) {
      acc.madd(g);
    } else if (kind[e] == 2) {
      g.y = g.y.neg();
      acc.madd(g);
    } else {
      const Fp<CfgFr> cm = Fp<CfgFr>::load(coeff + (size_t)e * 8).from_mont();       // coefficients arrive as the matrix holds them
      uint32_t c[8];
#pragma unroll
      for (int l = 0; l < 8; l++) c[l] = cm.v[l];
      XYZZ<F> t = XYZZ<F>::inf();
      for (int bit = BITS; bit >= 0; bit--) {
        t = t.dbl();
        if ((c[bit >> 5] >> (bit & 31)) & 1) t.madd(g);
      }
      acc.add(t);
    }
  }
  XYZZ<F> r = L_inf[m] ? XYZZ<F>::inf() : XYZZ<F>::from_affine(Affine<F>::load(L_xy + m * Affine<F>::BYTES));
  r.add(acc.neg());
  r.to_affine().store(out_xy + m * Affine<F>::BYTES);
  out_inf[m] = r.is_inf() ? 1 : 0;
}
#endif

// ------------------------------------------------------------------------------------------- Groth16 assembly
// (prover.rs:192-210 after the folding described in groth16.hip)
namespace ZKP_CFG_SYM(cfg) {
#if ZKP_CFG_GROUP == 1
// part 1 — T = s*g_a + r*g1_b -> slot 5; proof.a = affine(g_a).  Two waves, one per scalar multiplication.  The 254
// doublings of a dynamic point are an inherently serial chain; everything else is taken off it: lane 0 of each wave runs
// ONLY the doubling chain (Jacobian dbl-2009-l, a = 0: 2M + 5S = 7 products per step instead of the 9 of the XYZZ doubling
// plus ~5 of the conditional mixed addition) and publishes D_k = 2^k P to LDS; afterwards the 64 lanes of the wave sum the
// D_k whose scalar bit is set (4 candidates per lane, then an LDS tree).  3.3 -> ~2 ms of single-lane latency.
// GLV (round 3): both curves have j = 0, so phi(x, y) = (beta x, y) is an endomorphism with phi(P) = lambda P.  k = k1 + k2 lambda
// (mod r) with |k1|, |k2| < 2^129 (glv_constants.inc, generated and checked by tools/gen_glv.py): k P = k1 P + k2 phi(P) needs a
// doubling chain of 130 instead of 255 steps — the chain is the latency of this kernel (1.75 -> ~0.95 ms).
#include "glv_constants.inc"
// c = (k * g) >> 256, k: 8 words, g: 5 words
__device__ __forceinline__ void glv_mulhi(const uint32_t* k, const uint32_t* g, uint32_t* c) {
  uint32_t t[13];
#pragma unroll
  for (int i = 0; i < 13; i++) t[i] = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t x = (uint64_t)k[i] * g[j] + t[i + j] + carry;
      t[i + j] = (uint32_t)x;
      carry = x >> 32;
    }
    t[8 + j] = (uint32_t)carry;
  }
#pragma unroll
  for (int i = 0; i < 5; i++) c[i] = t[8 + i];
}
// acc (10 words, two's complement) += sign * x * y  (x, y: 5-word magnitudes; neg != 0 subtracts)
__device__ __forceinline__ void glv_mac(uint32_t* acc, const uint32_t* x, const uint32_t* y, int neg) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      uint64_t v = (uint64_t)x[i] * y[j] + t[i + j] + carry;
      t[i + j] = (uint32_t)v;
      carry = v >> 32;
    }
    t[5 + j] = (uint32_t)carry;
  }
  uint64_t c = neg ? 1 : 0;                                 // subtract = add the two's complement
#pragma unroll
  for (int i = 0; i < 10; i++) {
    c += (uint64_t)acc[i] + (neg ? ~t[i] : t[i]);
    acc[i] = (uint32_t)c;
    c >>= 32;
  }
}
// 10-word two's complement -> 5-word magnitude + sign
__device__ __forceinline__ int glv_abs(uint32_t* acc, uint32_t* out) {
  const int neg = (int)(acc[9] >> 31);
  uint64_t c = neg ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    c += neg ? (uint32_t)~acc[i] : acc[i];
    acc[i] = (uint32_t)c;
    c >>= 32;
  }
#pragma unroll
  for (int i = 0; i < 5; i++) out[i] = acc[i];
  return neg;
}
// k (canonical scalar, 8 words) -> |k1|, |k2| < 2^BITS and their signs: k = k1 + k2 lambda (mod r).  The identity holds for ANY
// integers c1, c2 (a_i + b_i lambda = 0 mod r), so the truncated quotients only cost a bit of size (gen_glv.py checks the bound).
template <class G>
__device__ __forceinline__ void glv_decompose(const uint32_t* k, uint32_t* k1, int* neg1, uint32_t* k2, int* neg2) {
  uint32_t g1[5], g2[5], a1[5], b1[5], a2[5], b2[5], c1[5], c2[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    g1[i] = G::G1[i]; g2[i] = G::G2[i]; a1[i] = G::A1[i]; b1[i] = G::B1[i]; a2[i] = G::A2[i]; b2[i] = G::B2[i];
  }
  glv_mulhi(k, g1, c1);                                     // |c1|, sign G1_NEG
  glv_mulhi(k, g2, c2);
  uint32_t acc[10];
#pragma unroll
  for (int i = 0; i < 10; i++) acc[i] = i < 8 ? k[i] : 0;
  glv_mac(acc, c1, a1, !(G::G1_NEG ^ G::A1_NEG));           // k1 = k - c1 a1 - c2 a2
  glv_mac(acc, c2, a2, !(G::G2_NEG ^ G::A2_NEG));
  *neg1 = glv_abs(acc, k1);
#pragma unroll
  for (int i = 0; i < 10; i++) acc[i] = 0;
  glv_mac(acc, c1, b1, !(G::G1_NEG ^ G::B1_NEG));           // k2 = -c1 b1 - c2 b2
  glv_mac(acc, c2, b2, !(G::G2_NEG ^ G::B2_NEG));
  *neg2 = glv_abs(acc, k2);
}

template <class F>
__device__ __forceinline__ void jac_dbl(F& X, F& Y, F& Z) {
  F A = X.sqr(), B = Y.sqr();
  F C = B.sqr();
  F D = ((X + B).sqr() - A - C).dbl();
  F E = A.dbl() + A;
  F Fq = E.sqr();
  F Z3 = (Y * Z).dbl();
  X = Fq - D.dbl();
  Y = E * (D - X) - C.dbl().dbl().dbl();
  Z = Z3;
}
__global__ __launch_bounds__(128) void assemble_g1_part1_kernel(char* __restrict__ res, size_t slot,
                                                                const uint32_t* __restrict__ rs,
                                                                uint32_t* __restrict__ out,
                                                                uint32_t* __restrict__ flags) {
  using F = CfgF;
  using G = GlvConst<CfgFq>;
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NB = G::BITS;                                  // doubling-chain length: bits of |k1|, |k2|
  constexpr size_t JB = 3 * 4 * F::N;                          // bytes of a Jacobian point
  const int t = threadIdx.x, chain = t >> 6, lane = t & 63;
  char* D = smem + (size_t)chain * NB * JB;                    // D[k] = 2^k * P (Jacobian)
  char* red = smem + 2 * NB * JB + (size_t)chain * 64 * XYZZ<F>::BYTES;
  // chain 0: g_a (slot 0) times s ; chain 1: g1_b (slot 1) times r
  Fp<CfgFr> k = Fp<CfgFr>::load(rs + (chain == 0 ? 8 : 0)).from_mont();
  uint32_t k1[5], k2[5];
  int neg1, neg2;
  glv_decompose<G>(k.v, k1, &neg1, k2, &neg2);                 // k P = k1 P + k2 phi(P)
  if (lane == 0) {
    XYZZ<F> p = XYZZ<F>::load(res + (size_t)chain * slot);
    F X, Y, Z;
    if (out) {
      Affine<F> a = p.to_affine();
      if (chain == 0) {
        a.store(out);                                          // proof.a
        flags[0] = p.is_inf() ? 1 : 0;
      }
      X = a.x, Y = a.y, Z = p.is_inf() ? F::zero() : F::one();
    } else {
      // host tail (round 4): proof.a becomes affine on the host from the XYZZ slot; the doubling chain starts from the Jacobian
      // form (X ZZ^2 : Y ZZZ^2 : ZZZ) of the same point — four products instead of a 254-step Fermat inversion in front of it
      X = p.x * p.zz.sqr();
      Y = p.y * p.zzz.sqr();
      Z = p.is_inf() ? F::zero() : p.zzz;
    }
    for (int b = 0; b < NB; b++) {
      X.store(D + (size_t)b * JB);
      Y.store(D + (size_t)b * JB + 4 * F::N);
      Z.store(D + (size_t)b * JB + 8 * F::N);
      if (b + 1 < NB) jac_dbl(X, Y, Z);
    }
  }
  __syncthreads();
  F beta;
#pragma unroll
  for (int i = 0; i < F::N; i++) beta.v[i] = G::BETA_MONT[i];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int b = lane; b < NB; b += 64) {
    const bool t1 = (k1[b >> 5] >> (b & 31)) & 1, t2 = (k2[b >> 5] >> (b & 31)) & 1;
    if (!(t1 || t2)) continue;
    const F X = F::load(D + (size_t)b * JB), Y = F::load(D + (size_t)b * JB + 4 * F::N), Z = F::load(D + (size_t)b * JB + 8 * F::N);
    if (Z.is_zero()) continue;
    const F zz = Z.sqr();
    const F zzz = zz * Z;
    if (t1) acc.add(XYZZ<F>{X, neg1 ? Y.neg() : Y, zz, zzz});
    if (t2) acc.add(XYZZ<F>{beta * X, neg2 ? Y.neg() : Y, zz, zzz});      // phi(X : Y : Z) = (beta X : Y : Z)
  }
  acc.store(red + (size_t)lane * XYZZ<F>::BYTES);
  __syncthreads();
  for (int s2 = 32; s2 > 0; s2 >>= 1) {
    if (lane < s2) {
      XYZZ<F> o = XYZZ<F>::load(red + (size_t)(lane + s2) * XYZZ<F>::BYTES);
      if (!o.is_inf() || !acc.is_inf()) {
        acc.add(o);
        acc.store(red + (size_t)lane * XYZZ<F>::BYTES);
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    acc.add(XYZZ<F>::load(smem + 2 * NB * JB + 64 * XYZZ<F>::BYTES));     // chain 1's sum
    acc.store(res + 5 * slot);
  }
}
// part 2 — C = T + h_acc + l'  (l' includes -rs*delta)
__global__ __launch_bounds__(64) void assemble_g1_part2_kernel(char* __restrict__ res, size_t slot,
                                                               uint32_t* __restrict__ out,
                                                               uint32_t* __restrict__ flags, int c_off_words) {
  using F = CfgF;
  if (threadIdx.x || blockIdx.x) return;
  XYZZ<F> acc = XYZZ<F>::load(res + 5 * slot);
  acc.add(XYZZ<F>::load(res + 3 * slot));
  acc.add(XYZZ<F>::load(res + 4 * slot));
  if (!out) {                                                  // host tail: C leaves as XYZZ (slot 5), the host inverts
    acc.store(res + 5 * slot);
    return;
  }
  Affine<F> c = acc.to_affine();
  c.store(out + c_off_words);
  flags[2] = acc.is_inf() ? 1 : 0;
}
#else
__global__ __launch_bounds__(64) void assemble_g2_kernel(const char* __restrict__ res, size_t slot,
                                                         uint32_t* __restrict__ out, uint32_t* __restrict__ flags,
                                                         int out_off_words) {
  if (threadIdx.x || blockIdx.x) return;
  XYZZ<CfgF> p = XYZZ<CfgF>::load(res + 2 * slot);
  p.to_affine().store(out + out_off_words);
  flags[1] = p.is_inf() ? 1 : 0;
}
#endif
}  // namespace ZKP_CFG_SYM(cfg)

// ------------------------------------------------------------------------------------------- launch table
void ZKP_CFG_SYM(msm_accumulate_launch)(hipStream_t, const char*, const uint32_t*, const uint4*, const uint32_t*, uint32_t,
                                        char*, char*, uint32_t*, uint32_t);   // msm_acc.hip

namespace {
using F = CfgF;
constexpr size_t XB = XYZZ<F>::BYTES;        // canonical (saturated) XYZZ: results, assembly slots
constexpr size_t BB = BkPoint<F>::BYTES;     // bucket / pyramid points (unsaturated layout, bucket_dev.hpp)
void l_ingest(hipStream_t s, char* table, const uint8_t* inf, size_t n) {
  hipLaunchKernelGGL(ingest_kernel<F>, dim3((n + 255) / 256), dim3(256), 0, s, table, inf, n);
}
void l_precompute(hipStream_t s, char* table, size_t n, int c, int W, int wide) {
  hipLaunchKernelGGL(precompute_kernel<F>, dim3((n + 127) / 128), dim3(128), 0, s, table, n, c, W, wide);
}
void l_combine(hipStream_t s, const uint32_t* long_list, const uint32_t* n_long_dev, const uint32_t* toff,
               const char* partial, char* buckets, uint32_t init) {
  hipLaunchKernelGGL(combine_small_kernel<F>, dim3(256), dim3(256), 0, s, long_list, n_long_dev, toff, partial, buckets, init);
  hipLaunchKernelGGL(combine_kernel<F>, dim3(512), dim3(256), 256 * BB, s, long_list, n_long_dev, toff, partial, buckets, init);
}
void l_pair(hipStream_t s, const char* in, char* out, uint32_t count) {
  hipLaunchKernelGGL(pair_kernel<F>, dim3((count + 255) / 256), dim3(256), 0, s, in, out, count);
}
void l_pair_top(hipStream_t s, char* base, uint32_t count) {
  hipLaunchKernelGGL(pair_top_kernel<F>, dim3(1), dim3(PAIR_TOP_THREADS), 0, s, base, count);
}
void l_segsum(hipStream_t s, const char* base, const SegPlan* plan, char* partial, uint32_t blocks, const char* base_hi, int l_hi) {
  hipLaunchKernelGGL(segsum_kernel<F>, dim3(blocks), dim3(256), 256 * BB, s, base, *plan, partial, base_hi, l_hi);
}
void l_pair_top_segsum(hipStream_t s, char* top_base, uint32_t top_cnt, const char* base, const SegPlan* plan, char* partial,
                       uint32_t blocks) {
  hipLaunchKernelGGL(pair_top_segsum_kernel<F>, dim3(1 + blocks), dim3(PAIR_TOP_THREADS), 256 * BB, s, top_base, top_cnt, base, *plan,
                     partial);
}
void l_final(hipStream_t s, const char* O, int L, const char* root, char* out_xyzz, uint32_t* out_jac) {
  constexpr int threads = (QuadCoop<F>::ON && QuadCoop<F>::GROUP == 1) ? 128 : 64;            // G1: 32 quads (coop_dev.hpp)
  hipLaunchKernelGGL(final_kernel<F>, dim3(1), dim3(threads), 64 * BB, s, O, L, root, out_xyzz, out_jac);
}
void l_identity(hipStream_t s, char* out_xyzz, uint32_t* out_jac) {
  hipLaunchKernelGGL(write_identity_kernel<F>, dim3(1), dim3(64), 0, s, out_xyzz, out_jac);
}
void l_fold(hipStream_t s, const uint32_t* pts, int k, uint32_t* out_jac) {
  hipLaunchKernelGGL(fold_kernel<F>, dim3(1), dim3(64), 0, s, pts, k, out_jac);
}
void l_segsum_desc(hipStream_t s, const char* base, const SegDesc* descs, uint32_t n_desc, char* out) {
  if (n_desc) hipLaunchKernelGGL(segsum_desc_kernel<F>, dim3(n_desc), dim3(256), 256 * BB, s, base, descs, out);
}
void l_final_var(hipStream_t s, const char* R, const char* roots, int c, int W, char* out_xyzz, uint32_t* out_jac) {
  hipLaunchKernelGGL(final_var_kernel<F>, dim3(1), dim3(256), 256 * BB, s, R, roots, c, W, out_xyzz, out_jac);
}
#if ZKP_CFG_GROUP == 1
void l_gfft(hipStream_t s, const char* xy, const uint8_t* inf, size_t n_in, const uint32_t* scal, const uint32_t* tw, uint32_t log_n,
            char* X, char* out_xy, uint8_t* out_inf) {
  const size_t N = (size_t)1 << log_n;
  hipLaunchKernelGGL((gfft_scale_kernel<F, CFG_BITS>), dim3((N + 127) / 128), dim3(128), 0, s, xy, inf, n_in, scal, log_n, X);
  for (uint32_t st = 1; st <= log_n; st++)
    hipLaunchKernelGGL((gfft_stage_kernel<F, CFG_BITS>), dim3((N / 2 + 127) / 128), dim3(128), 0, s, X, tw, log_n, st);
  hipLaunchKernelGGL(gfft_affine_kernel<F>, dim3((N + 127) / 128), dim3(128), 0, s, X, N, out_xy, out_inf);
}
void l_lfold(hipStream_t s, const char* L_xy, const uint8_t* L_inf, size_t n_vars, const uint32_t* col_ptr, const uint32_t* rows,
             const uint8_t* kind, const uint32_t* coeff, const char* G_xy, const uint8_t* G_inf, char* out_xy, uint8_t* out_inf) {
  if (n_vars)
    hipLaunchKernelGGL((lfold_kernel<F, CFG_BITS>), dim3((n_vars + 127) / 128), dim3(128), 0, s, L_xy, L_inf, n_vars, col_ptr, rows,
                       kind, coeff, G_xy, G_inf, out_xy, out_inf);
}
#endif
void l_fold_slots(hipStream_t s, const char* gathered, size_t rank_stride, int world, size_t slot, uint32_t mask,
                  char* res) {
  hipLaunchKernelGGL(fold_slots_kernel<F>, dim3(1), dim3(64), 0, s, gathered, rank_stride, world, slot, mask, res);
}
void l_into_affine(hipStream_t s, const uint32_t* jac, uint32_t* xy, uint32_t* inf) {
  hipLaunchKernelGGL(into_affine_kernel<F>, dim3(1), dim3(64), 0, s, jac, xy, inf);
}
void l_fold_affine_batch(hipStream_t s, const uint32_t* ja, const uint32_t* jb, const uint32_t* has_b, int k, uint32_t* xy,
                         uint32_t* inf) {
  if (k > 0) hipLaunchKernelGGL(fold_affine_batch_kernel<F>, dim3((k + 63) / 64), dim3(64), 0, s, ja, jb, has_b, k, xy, inf);
}
void l_decompress(hipStream_t s, const uint32_t* bytes, size_t n, const uint32_t* bcoef, char* xy, uint8_t* inf, uint32_t* status) {
  hipLaunchKernelGGL(curve_b_kernel<F>, dim3(1), dim3(64), 0, s, const_cast<uint32_t*>(bcoef));   // bcoef: F::N words of scratch
  if (n) hipLaunchKernelGGL(decompress_kernel<F>, dim3((n + 63) / 64), dim3(64), 0, s, bytes, n, bcoef, xy, inf, status);
}
void l_compress(hipStream_t s, const char* xy, const uint8_t* inf, size_t n, uint32_t* bytes) {
  if (n) hipLaunchKernelGGL(compress_kernel<F>, dim3((n + 63) / 64), dim3(64), 0, s, xy, inf, n, bytes);
}
void l_subgroup_check(hipStream_t s, const char* xy, const uint8_t* inf, size_t n, const uint32_t* bcoef, uint32_t* status) {
  hipLaunchKernelGGL(curve_b_kernel<F>, dim3(1), dim3(64), 0, s, const_cast<uint32_t*>(bcoef));
  if (n) hipLaunchKernelGGL(subgroup_check_kernel<F>, dim3((n + 63) / 64), dim3(64), 0, s, xy, inf, n, bcoef, status);
}
void l_from_jacobian(hipStream_t s, const uint32_t* jac, char* out) {
  hipLaunchKernelGGL(from_jacobian_kernel<F>, dim3(1), dim3(64), 0, s, jac, out);
}
void l_fixed_base(hipStream_t s, const uint32_t* base, const uint32_t* scalars, size_t n, char* out_xy,
                  uint8_t* out_inf) {
  hipLaunchKernelGGL((fixed_base_kernel<F, CFG_BITS>), dim3((n + 127) / 128), dim3(128), 0, s, base, scalars, n,
                     out_xy, out_inf);
}
#if ZKP_CFG_GROUP == 1
void l_assemble_g1_p1(hipStream_t s, char* res, size_t slot, const uint32_t* rs, uint32_t* out, uint32_t* flags) {
  const size_t lds = 2 * (size_t)GlvConst<CfgFq>::BITS * 3 * 4 * F::N + 2 * 64 * XB;
  hipLaunchKernelGGL(assemble_g1_part1_kernel, dim3(1), dim3(128), lds, s, res, slot, rs, out, flags);
}
void l_assemble_g1_p2(hipStream_t s, char* res, size_t slot, uint32_t* out, uint32_t* flags, int c_off_words) {
  hipLaunchKernelGGL(assemble_g1_part2_kernel, dim3(1), dim3(64), 0, s, res, slot, out, flags, c_off_words);
}
#else
void l_assemble_g2(hipStream_t s, const char* res, size_t slot, uint32_t* out, uint32_t* flags, int off) {
  hipLaunchKernelGGL(assemble_g2_kernel, dim3(1), dim3(64), 0, s, res, slot, out, flags, off);
}
#endif
}  // namespace

const MsmVtbl* ZKP_CFG_SYM(msm_vtbl)() {
  static const MsmVtbl v = {
      F::N, Affine<F>::BYTES, XYZZ<F>::BYTES, BkPoint<F>::BYTES, CFG_BITS,
      l_ingest, l_precompute, ZKP_CFG_SYM(msm_accumulate_launch), l_combine, l_pair, l_segsum, l_final, l_identity, l_fold,
      l_into_affine, l_fold_affine_batch, l_decompress, l_compress, l_subgroup_check, l_from_jacobian, l_fixed_base, l_segsum_desc, l_final_var, l_fold_slots,
#if ZKP_CFG_GROUP == 1
      l_assemble_g1_p1, l_assemble_g1_p2, nullptr,
#else
      nullptr, nullptr, l_assemble_g2,
#endif
      l_pair_top,
#if ZKP_CFG_GROUP == 1
      l_gfft, l_lfold,
#else
      nullptr, nullptr,
#endif
      l_pair_top_segsum,
  };
  return &v;
}

}  // namespace zkp
