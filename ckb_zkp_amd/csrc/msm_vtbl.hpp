// Per-(curve, group) kernel launch table.  Each configuration is compiled in its own translation unit
// (msm_group.hip / msm_acc.hip with -DZKP_CFG_CURVE=.. -DZKP_CFG_GROUP=..) so the four configurations build in
// parallel and the hot accumulate kernel can be rebuilt alone.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zkp {

// Entries per block of the segmented sums (chunk / 256 - 1 serial additions per thread, then an 8-level LDS tree: the tree runs
// at < 50 % lane utilisation, the serial part at 100 %).  Pipelined batches take the large chunk (1024: 122.3, 2048: 122.7,
// 4096: 124.0, 8192: 123.5 proofs/s), single proofs / single MSMs the small one (latency 10.3 vs 10.5 ms per proof).
constexpr int SEG_CHUNK = 1024;
constexpr int SEG_CHUNK_BATCH = 4096;
constexpr uint32_t MSM_TASK_CAP = 128;     // max entries one lane accumulates before the bucket is split (64: 128.7 / 8.44, 128: 129.8 / 8.51 proofs/s at 2^20 / 2^24)
struct SegPlan {
  int L;                       // number of segments
  uint32_t chunk;              // entries per block
  uint32_t first_block[26];    // blocks of segment l: first_block[l] .. first_block[l+1]
  uint32_t off[25];            // element offset of segment l's first entry
  uint32_t stride[25];         // element stride
  uint32_t count[25];          // entries in segment l
};

// one block of a descriptor-driven segmented sum: out[d.out] = sum_{i < count} base[off + i * stride]
struct SegDesc {
  uint32_t off, stride, count, out;
};

struct MsmVtbl {
  int fN;                      // 32-bit words per coordinate (8 / 16 / 12 / 24)
  size_t aff_bytes, xyzz_bytes;
  size_t bucket_bytes;         // one bucket / partial / pyramid point (unsaturated XYZZ layout, bucket_dev.hpp)
  int scalar_bits;
  void (*ingest)(hipStream_t, char* table, const uint8_t* inf, size_t n);
  void (*precompute)(hipStream_t, char* table, size_t n, int c, int W, int wide);
  // one lane per TASK (<= MSM_TASK_CAP consecutive entries of one bucket, tasks ordered by length);
  // dst < 0x80000000: bucket index (single-task bucket), else partial slot (dst & 0x7fffffff)
  // desc[t] = {first entry, length, dst, -} of the t-th task IN SCHEDULE ORDER (task_order_kernel): one coalesced 16-B load per
  // lane instead of four dependent random ones
  // init != 0: the bucket array already holds the buckets of another MSM (bucket chaining, ctx.hpp): a single-task bucket starts
  // from its stored value instead of the identity
  void (*accumulate)(hipStream_t, const char* table, const uint32_t* vals, const uint4* desc,
                     const uint32_t* n_tasks_dev, uint32_t max_tasks, char* buckets, char* partial, uint32_t* redo, uint32_t init);
  // redo: max_tasks + 1 words of scratch ([0] = count, zeroed by the launcher): tasks the fast path abandoned because an
  // operand might equal +-accumulator are listed there and redone by an exact second kernel
  // one wave per multi-task bucket: buckets[b] = sum of its partials
  void (*combine)(hipStream_t, const uint32_t* long_list, const uint32_t* n_long_dev, const uint32_t* toff,
                  const char* partial, char* buckets, uint32_t init);   // init: add the bucket's stored value too
  void (*pair)(hipStream_t, const char* in, char* out, uint32_t count);
  // segments l >= l_hi read from base_hi instead of base (pass nullptr, 1 << 30 for a plain segmented sum)
  void (*segsum)(hipStream_t, const char* base, const SegPlan* plan, char* partial, uint32_t blocks, const char* base_hi, int l_hi);
  void (*final)(hipStream_t, const char* O, int L, const char* root, char* out_xyzz, uint32_t* out_jac);
  void (*write_identity)(hipStream_t, char* out_xyzz, uint32_t* out_jac);
  void (*fold)(hipStream_t, const uint32_t* pts, int k, uint32_t* out_jac);
  void (*into_affine)(hipStream_t, const uint32_t* jac, uint32_t* xy, uint32_t* inf);
  // k points at once, one lane each: out_i = affine(a_i + b_i) (b_i Jacobian too; has_b[i] == 0: a_i alone) — the k
  // commitments of a Marlin round share ONE launch instead of k single-lane inversions (0.3 ms each) one after the other
  void (*fold_affine_batch)(hipStream_t, const uint32_t* jac_a, const uint32_t* jac_b, const uint32_t* has_b, int k,
                            uint32_t* xy, uint32_t* inf);
  // ark-serialize compressed points <-> affine Montgomery, one lane per point (msm_group.hip "point codec"); bcoef = the curve's b
  void (*decompress)(hipStream_t, const uint32_t* bytes, size_t n, const uint32_t* bcoef, char* xy, uint8_t* inf, uint32_t* status);
  void (*compress)(hipStream_t, const char* xy, const uint8_t* inf, size_t n, uint32_t* bytes);
  // on the curve and [r]P = O, one lane per point; status: 1 + index of the first failing point (atomicMin)
  void (*subgroup_check)(hipStream_t, const char* xy, const uint8_t* inf, size_t n, const uint32_t* bcoef, uint32_t* status);
  void (*from_jacobian)(hipStream_t, const uint32_t* jac, char* out_xyzz);
  void (*fixed_base)(hipStream_t, const uint32_t* base, const uint32_t* scalars, size_t n, char* out_xy,
                     uint8_t* out_inf);
  // variable-base reduction (msm.hip msm_var_run): descriptor-driven segmented sums, then
  // out = sum_{t < 256} 2^t R[t] + sum_{w < W} 2^(c*w) roots[w]
  void (*segsum_desc)(hipStream_t, const char* base, const SegDesc* descs, uint32_t n_desc, char* out);
  void (*final_var)(hipStream_t, const char* R, const char* roots, int c, int W, char* out_xyzz, uint32_t* out_jac);
  // sum slot t (t < 5, mask bit t set) of `world` gathered rank buffers into res + t * slot (XYZZ slots)
  void (*fold_slots)(hipStream_t, const char* gathered, size_t rank_stride, int world, size_t slot, uint32_t mask,
                     char* res);
  // Groth16 assembly (groth16.hip): G1 tables implement assemble_g1, G2 tables assemble_g2
  // part 1 needs only g_a (slot 0) and g1_b (slot 1): proof.a = affine(g_a); T = s*g_a + r*g1_b -> slot 5.
  // part 2: C = T + h_acc (slot 3) + l' (slot 4) -> affine.  Split so part 1 overlaps the remaining MSMs.
  void (*assemble_g1_part1)(hipStream_t, char* res, size_t slot, const uint32_t* rs, uint32_t* out, uint32_t* flags);
  void (*assemble_g1_part2)(hipStream_t, char* res, size_t slot, uint32_t* out, uint32_t* flags, int c_off_words);
  void (*assemble_g2)(hipStream_t, const char* res, size_t slot, uint32_t* out, uint32_t* flags, int out_off_words);
  // the top of the pairwise pyramid in one launch: levels from the one with `count` (<= PAIR_TOP_MAX; 1024: 129.4, 2048: 130.1, 4096: 129.5 proofs/s) entries at `base` down to
  // the root, each level stored directly behind its predecessor
  void (*pair_top)(hipStream_t, char* base, uint32_t count);
  // group-element transform of a base vector (msm_group.hip "group-element transform"; groth16.hip: the H query in evaluation form):
  // out_j = sum_i tw^(ij) (scal_i * P_i), i, j < 2^log_n; scal / tw: canonical 8-word scalars (tw: 2^(log_n-1) powers);
  // X: 2^log_n XYZZ points of scratch
  void (*gfft)(hipStream_t, const char* xy, const uint8_t* inf, size_t n_in, const uint32_t* scal, const uint32_t* tw, uint32_t log_n,
               char* X, char* out_xy, uint8_t* out_inf);
  // out_m = L_m - sum over column m of a sparse matrix (CSC: col_ptr / rows / kind / coeff) of coeff * G_row, m < n_vars
  void (*lfold)(hipStream_t, const char* L_xy, const uint8_t* L_inf, size_t n_vars, const uint32_t* col_ptr, const uint32_t* rows,
                const uint8_t* kind, const uint32_t* coeff, const char* G_xy, const uint8_t* G_inf, char* out_xy, uint8_t* out_inf);
  // pair_top on (top_base, top_cnt) and the segmented sums of `plan` over `base` in ONE launch (block 0 | blocks 1..blocks)
  void (*pair_top_segsum)(hipStream_t, char* top_base, uint32_t top_cnt, const char* base, const SegPlan* plan, char* partial,
                          uint32_t blocks);
};
constexpr uint32_t PAIR_TOP_MAX = 2048;

const MsmVtbl* msm_vtbl(int curve, int group);     // msm.hip; throws StatusError on unknown config

}  // namespace zkp
