// K7 — bucket accumulation: the dominant kernel of the whole prover (>= 85 % of the field multiplications of
// a Groth16 proof).  One lane owns one bucket: it walks the bucket's run of the sorted (bucket, point) list,
// gathers the pre-multiplied affine base T[w][i] (64 B for BN254 G1; random access, served mostly from the
// 256 MiB Infinity Cache / HBM), negates y for negative signed digits, and mixed-adds (madd-2008-s, 8M+2S)
// into an XYZZ accumulator that stays in VGPRs for the whole run.  Buckets are written once, coalesced by
// bucket index.  Compiled per configuration; the BN254 configurations inline the Montgomery multiplier.
//
// Replaces the bucket loop of ark-ec 0.2 `VariableBaseMSM::multi_scalar_mul`
// (reference call sites: /root/reference/groth16/src/prover.rs:187,190,220).
#include <cstdlib>

#include "bucket_dev.hpp"
#include "ec_dev.hpp"
#include "msm_vtbl.hpp"
#include "unsat_dev.hpp"

namespace zkp {

#if ZKP_CFG_CURVE == 0
using CfgFq = Bn254Fq;
#else
using CfgFq = Bls381Fq;
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

// -DZKP_ACC_NT_GATHER: window-table gathers carry the non-temporal hint (A/B switch)
#if defined(ZKP_ACC_NT_GATHER)
#define ZKP_GATHER load_nt
#else
#define ZKP_GATHER load
#endif

namespace ZKP_CFG_SYM(cfg) {
// One 16-B load brings four consecutive values of the lane's run (entries are 4 B; a lane walks its own run, so the 64 lanes of
// a wave touch 64 different lines per step and a 4-B load per entry cost up to one 128-B fabric read each once the window-table
// gathers had swept the line out of L2: PMC, 1.55 -> 1.30 read requests per entry with four values per load, eight values now).
struct ValQuads {
  const uint32_t* vals;
  uint4 q0, q1;                                     // eight consecutive values: one 32-B-aligned pair of 16-B loads per 8 entries
  __device__ __forceinline__ void fill(uint32_t e) {
    const uint4* p = reinterpret_cast<const uint4*>(vals + (e & ~7u));
    q0 = p[0];
    q1 = p[1];
  }
  __device__ __forceinline__ ValQuads(const uint32_t* v, uint32_t e0) : vals(v) { fill(e0); }
  __device__ __forceinline__ uint32_t get(uint32_t e, uint32_t e0) {
    const uint32_t k = e & 7u;
    if (k == 0 && e != e0) fill(e);
    const uint4 q = (k & 4u) ? q1 : q0;
    const uint32_t j = k & 3u;
    return j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w;
  }
};
template <int MINW>
__global__ __launch_bounds__(256, MINW) void accumulate_kernel(const char* __restrict__ table,
                                                         const uint32_t* __restrict__ vals,
                                                         const uint4* __restrict__ desc,
                                                         const uint32_t* __restrict__ n_tasks_dev,
                                                         char* __restrict__ buckets, char* __restrict__ partial, uint32_t idx_mask,
                                                         uint32_t* __restrict__ redo, uint32_t init) {
  using F = CfgF;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= *n_tasks_dev) return;
  const uint4 td = desc[t];
  const uint32_t id = t;                       // what the redo list records: the task's position in the schedule
  const uint32_t e0 = td.x, e1 = e0 + td.y;
  const uint32_t d = td.z;
  // buckets and partial sums are BkPoint<F> (unsaturated layout): the accumulator is stored as it stands
  char* out = (d >> 31) ? partial + (size_t)(d & 0x7fffffffu) * BkPoint<F>::BYTES : buckets + (size_t)d * BkPoint<F>::BYTES;
  ValQuads vq(vals, e0);
#if ZKP_CFG_GROUP == 1 && defined(ZKP_ACC_UNSAT)
  // G1: accumulate on unsaturated limbs (unsat_dev.hpp); the window table and the buckets keep the saturated layout
  XYZZu<CfgFq> acc;
  acc.inf = true;
  if ((init & 1u) && !(d >> 31)) {                      // bucket chaining: continue from the bucket another MSM left here
    const BkPoint<F> b0 = BkPoint<F>::load(out);
    if (!b0.is_inf()) acc = b0.v;
  }
#if defined(ZKP_ACC_PREFETCH)
  // software pipeline: the gather of entry e + 1 is in flight while entry e is added (A/B switch; + 16 VGPRs -> 3 waves/SIMD)
  uint32_t v_next = vq.get(e0, e0);
  Affine<F> p_next = Affine<F>::ZKP_GATHER(table + (size_t)(v_next & idx_mask) * Affine<F>::BYTES);
#endif
  for (uint32_t e = e0; e < e1; e++) {
#if defined(ZKP_ACC_PREFETCH)
    const uint32_t v = v_next;
    Affine<F> p = p_next;
    if (e + 1 < e1) {
      v_next = vq.get(e + 1, e0);
      p_next = Affine<F>::ZKP_GATHER(table + (size_t)(v_next & idx_mask) * Affine<F>::BYTES);
    }
#else
    uint32_t v = vq.get(e, e0);
    Affine<F> p = Affine<F>::ZKP_GATHER(table + (size_t)(v & idx_mask) * Affine<F>::BYTES);
#endif
    if (p.is_inf()) continue;
    const Fu<CfgFq> ux = Fu<CfgFq>::from_sat(p.x), uy = Fu<CfgFq>::from_sat(p.y);
    xyzz_madd_u<CfgFq>(acc, ux, uy, 0u - (v >> 31));   // the digit's sign goes into the formulas (R = +-S2 - Y1)
  }
  // (init bit 1, ZKP_DEBUG_FORCE_REDO: every eighth task takes the exact path as well — the tests exercise the redo kernel, with
  //  and without chaining, on ordinary inputs)
  if (xyzz_u_degenerate<CfgFq>(acc) || ((init & 2u) && (t & 7u) == 0)) {
    // some operand equalled +-accumulator (doubling / cancellation; zz == 0 from then on): the whole task goes to the exact
    // kernel below instead of carrying the exceptional formulas — or a test per addition — through the hot loop
    redo[1 + atomicAdd(redo, 1u)] = id;
    return;
  }
  BkPoint<F> r;
  r.v = acc;
  r.store(out);
#elif ZKP_CFG_GROUP == 2 && defined(ZKP_ACC_UNSAT_G2)
  // G2 (BN254): Fq2 accumulator on unsaturated limbs, schoolbook products with lazily reduced sums (unsat_dev.hpp)
  // (no bucket chaining for G2 — msm.hip refuses it: loading a stored bucket in this prologue took the kernel from 213 to 256 VGPRs,
  //  2 -> 1 waves per SIMD, 1.24 -> 1.39 ms per B-query, for a path no caller uses)
  XYZZu2<CfgFq> acc;
  acc.inf = true;
  (void)init;
  for (uint32_t e = e0; e < e1; e++) {
    uint32_t v = vq.get(e, e0);
    Affine<F> p = Affine<F>::ZKP_GATHER(table + (size_t)(v & idx_mask) * Affine<F>::BYTES);
    if (p.is_inf()) continue;
    using U = Fu<CfgFq>;
    xyzz_madd_u2<CfgFq>(acc, U::from_sat(p.x.c0), U::from_sat(p.x.c1), U::from_sat(p.y.c0), U::from_sat(p.y.c1), 0u - (v >> 31));
  }
  if (xyzz_u2_degenerate<CfgFq>(acc) || ((init & 2u) && (t & 7u) == 0)) {
    redo[1 + atomicAdd(redo, 1u)] = id;
    return;
  }
  BkPoint<F> r;
  r.v = acc;
  r.store(out);
#else
  XYZZ<F> acc = (ZKP_CFG_GROUP == 1 && (init & 1u) && !(d >> 31)) ? BkPoint<F>::load(out).to_sat() : XYZZ<F>::inf();
  // (a software-pipelined gather of entry e+1 was tried twice — G1: +20 VGPRs -> spills; G2 after the redo split:
  //  276 VGPRs -> 1 wave/SIMD, or 256 with launch bounds — no gain either time: the gather latency is covered)
  for (uint32_t e = e0; e < e1; e++) {
    uint32_t v = vq.get(e, e0);
    Affine<F> p = Affine<F>::ZKP_GATHER(table + (size_t)(v & idx_mask) * Affine<F>::BYTES);
    if (v >> 31) p.y = p.y.neg();
    if (!acc.madd_fast(p)) {                     // p = +-acc: the exact redo kernel takes the whole task
      redo[1 + atomicAdd(redo, 1u)] = id;
      return;
    }
  }
  BkPoint<F>::from_sat(acc).store(out);
#endif
}
// exact (saturated, all exceptional cases) accumulation of the tasks listed in redo[1 .. redo[0]]
__global__ __launch_bounds__(64) void accumulate_redo_kernel(const char* __restrict__ table, const uint32_t* __restrict__ vals,
                                                            const uint4* __restrict__ desc, char* __restrict__ buckets,
                                                            char* __restrict__ partial, uint32_t idx_mask,
                                                            const uint32_t* __restrict__ redo, uint32_t init) {
  using F = CfgF;
  const uint32_t count = redo[0];
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
    const uint4 td = desc[redo[1 + k]];
    const uint32_t e0 = td.x, e1 = e0 + td.y;
    const uint32_t d = td.z;
    // (the fast kernel did not store anything for this task: with bucket chaining the bucket still holds the other MSM's value)
    XYZZ<F> acc = ((init & 1u) && !(d >> 31)) ? BkPoint<F>::load(buckets + (size_t)d * BkPoint<F>::BYTES).to_sat() : XYZZ<F>::inf();
    for (uint32_t e = e0; e < e1; e++) {
      uint32_t v = vals[e];
      Affine<F> p = Affine<F>::load(table + (size_t)(v & idx_mask) * Affine<F>::BYTES);
      if (v >> 31) p.y = p.y.neg();
      acc.madd(p);
    }
    BkPoint<F>::from_sat(acc).store((d >> 31) ? partial + (size_t)(d & 0x7fffffffu) * BkPoint<F>::BYTES
                                              : buckets + (size_t)d * BkPoint<F>::BYTES);
  }
}
}  // namespace ZKP_CFG_SYM(cfg)

// Profiling builds only (-DZKP_DEBUG_GATHER, never the shipped library): ZKP_DEBUG_GATHER_MASK=0xffff makes every gather
// hit cache — WRONG RESULTS, used once to show the kernel is VALU-bound.  The shipped build has no such switch.
static uint32_t dbg_mask() {
#ifdef ZKP_DEBUG_GATHER
  static uint32_t m = [] { const char* e = getenv("ZKP_DEBUG_GATHER_MASK"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0x7fffffffu; }();
  return m;
#else
  return 0x7fffffffu;
#endif
}
void ZKP_CFG_SYM(msm_accumulate_launch)(hipStream_t s, const char* table, const uint32_t* vals, const uint4* desc,
                                        const uint32_t* n_tasks_dev, uint32_t max_tasks, char* buckets, char* partial,
                                        uint32_t* redo, uint32_t init) {
  // BLS12-381 G2: 1 wave/SIMD (VGPRs + AGPRs as spill space) vs 2 waves/SIMD (256 VGPRs + 168 B scratch since the round-3 streamed
  // reductions freed registers; it was 704 B in round 2, when two waves made the proof 8 % slower).  Round 4, 2^22 proofs on one box:
  // kernel 13.16 -> 12.29 ms, 0.717 -> 0.770 of the multiplier ceiling, 19.97 -> 20.15 proofs/s: two waves are the default
  // (ZKP_G2_ACC_OCC=1 restores).  BN254 G2 compiles to 213 VGPRs / two waves either way.
  static const unsigned lds = [] { const char* e = getenv("ZKP_ACC_LDS_BYTES"); return e ? (unsigned)atoi(e) : 0u; }();
  static const int occ = [] { const char* e = getenv("ZKP_G2_ACC_OCC"); return e ? atoi(e) : 2; }();
  static const int occ1 = [] { const char* e = getenv("ZKP_G1_ACC_OCC"); return e ? atoi(e) : 3; }();
  (void)hipMemsetAsync(redo, 0, sizeof(uint32_t), s);
  struct Redo {                                  // every launch path below is followed by the exact redo kernel
    hipStream_t s;
    const char* table;
    const uint32_t* vals;
    const uint4* desc;
    char *buckets, *partial;
    uint32_t* redo;
    uint32_t init;
    ~Redo() {
      hipLaunchKernelGGL(accumulate_redo_kernel, dim3(64), dim3(64), 0, s, table, vals, desc, buckets, partial,
                         dbg_mask(), redo, init);
    }
  } redo_after{s, table, vals, desc, buckets, partial, redo, init};
#if ZKP_CFG_GROUP == 1 && defined(ZKP_ACC_UNSAT)
  // BLS12-381 G1: 169 VGPRs = one register over three waves per SIMD; ZKP_G1_ACC_WAVES=3 compiles for three (A/B switch)
  static const int g1w = [] { const char* e = getenv("ZKP_G1_ACC_WAVES"); return e ? atoi(e) : 0; }();
  if (g1w == 3)
    hipLaunchKernelGGL(accumulate_kernel<3>, dim3((max_tasks + 255) / 256), dim3(256), lds, s, table, vals, desc,
                       n_tasks_dev, buckets, partial, dbg_mask(), redo, init);
  else
    hipLaunchKernelGGL(accumulate_kernel<1>, dim3((max_tasks + 255) / 256), dim3(256), lds, s, table, vals, desc,
                       n_tasks_dev, buckets, partial, dbg_mask(), redo, init);
  return;
#endif
  if (ZKP_CFG_GROUP == 1 && occ1 == 4)
    hipLaunchKernelGGL(accumulate_kernel<4>, dim3((max_tasks + 255) / 256), dim3(256), lds, s, table, vals, desc,
                       n_tasks_dev, buckets, partial, dbg_mask(), redo, init);
  else if (ZKP_CFG_GROUP == 2 && occ == 3)
    hipLaunchKernelGGL(accumulate_kernel<3>, dim3((max_tasks + 255) / 256), dim3(256), lds, s, table, vals, desc,
                       n_tasks_dev, buckets, partial, dbg_mask(), redo, init);
  else if (ZKP_CFG_GROUP == 2 && occ == 2)
    hipLaunchKernelGGL(accumulate_kernel<2>, dim3((max_tasks + 255) / 256), dim3(256), lds, s, table, vals, desc,
                       n_tasks_dev, buckets, partial, dbg_mask(), redo, init);
  else
    hipLaunchKernelGGL(accumulate_kernel<1>, dim3((max_tasks + 255) / 256), dim3(256), lds, s, table, vals, desc,
                       n_tasks_dev, buckets, partial, dbg_mask(), redo, init);
}

}  // namespace zkp
