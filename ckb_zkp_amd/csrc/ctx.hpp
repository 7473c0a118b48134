// Library context: device, stream, cached twiddle tables, grow-only scratch arena, resident bases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/zkp_accel.h"

namespace zkp {

struct HipError {
  hipError_t e;
  const char* what;
  int line;
};

#define ZKP_HIP(call)                                                   \
  do {                                                                  \
    hipError_t _e = (call);                                             \
    if (_e != hipSuccess) throw ::zkp::HipError{_e, #call, __LINE__};   \
  } while (0)

struct StatusError {
  int32_t status;
};
#define ZKP_REQUIRE(cond, status) \
  do {                            \
    if (!(cond)) throw ::zkp::StatusError{(status)}; \
  } while (0)

// Device buffer that grows on demand (never shrinks): scratch is sized once for the largest job.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void* get(size_t bytes) {
    if (bytes > cap) {
      if (p) (void)hipFree(p);
      p = nullptr;
      cap = 0;
      size_t want = bytes + (bytes >> 3) + 256;
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) throw StatusError{ZKP_ERR_OOM};
      cap = want;
    }
    return p;
  }
  template <class T>
  T* as(size_t count) { return reinterpret_cast<T*>(get(count * sizeof(T))); }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

// Twiddle tables for one (curve, log_n): two-level powers of w_N, w_N^-1, and coset-shift powers.
struct NttTables {
  int log_n = 0, h = 0;           // e = hi << h | lo
  uint32_t* w_lo = nullptr;       // w^i, i < 2^h            (forward)
  uint32_t* w_hi = nullptr;       // w^(i << h), i < 2^(log_n-h)
  uint32_t* wi_lo = nullptr;      // inverse root
  uint32_t* wi_hi = nullptr;
  uint32_t* g_lo = nullptr;       // g^i  (coset shift)
  uint32_t* g_hi = nullptr;
  uint32_t* gi_lo = nullptr;      // g^-i
  uint32_t* gi_hi = nullptr;      // g^-(i<<h) / N   (1/N folded in)
  uint32_t* sub_fwd = nullptr;    // w_R^k, k < R/2, R = 2^SMAX_TABLE (sub-FFT butterflies)
  uint32_t* sub_inv = nullptr;
  uint32_t* n_inv = nullptr;      // 1/N (one element)
  void* block = nullptr;
  // full-size tables (N elements each, built lazily for log_n <= 24): the inter-pass twiddle of pass p indexed by
  // output position, and the coset factors g^j, g^-k/N — one load + one product per element instead of two + two
  uint32_t* full_fwd[8] = {};
  uint32_t* full_inv[8] = {};
  uint32_t* full_g = nullptr;
  uint32_t* full_gi = nullptr;
  uint32_t* g_hi_n = nullptr;     // g^(i<<h) / N: coset pre-scale with the inverse transform's 1/N folded in (fused ifft -> coset_fft)
  uint32_t* full_g_n = nullptr;
  std::vector<void*> extra;
};

struct BasesEntry;   // msm.hip

// Scratch + stream of ONE in-flight MSM.  A context owns several so that independent MSMs of one proof run
// concurrently: the log-depth bucket-reduction tail of one MSM is latency-bound (a few wavefronts), and
// overlapping it with the throughput-bound accumulate kernel of another keeps the 256 CUs busy.
struct MsmWorkspace {
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t done = nullptr;
  hipEvent_t sorted = nullptr;     // recorded when the bucket sort + task schedule of the current MSM are complete
  hipEvent_t l1_done = nullptr;    // recorded when the level-1 pass (entries scattered into bins) of the current MSM is complete
  uint32_t chain_nb = 0;           // != 0: `buckets` holds the complete, unreduced buckets (chain_nb of them, chain_xb bytes each) of a deferred MSM
  size_t chain_xb = 0;
  hipEvent_t acc_done = nullptr;   // recorded when the buckets of an MSM whose reduction was deferred are complete (msm_defer_reduce)
  DevBuf keys, vals, keys2, vals2, sort_tmp, offsets, buckets, tmp, out, sched, scan_tmp, scan_tmp2, partial, redo;
};
struct Groth16Timing {
  zkp_groth16_timing t{};
};

}  // namespace zkp

// Everything one in-flight proof needs: a main stream, MSM workspaces (each with its own stream), NTT scratch.
// A context owns two lanes so that consecutive proofs overlap (zkp_groth16_prove_batch_dev): the latency-bound
// tails of proof i run while the throughput-bound kernels of proof i+1 keep the CUs busy.
struct zkp_lane {
  hipStream_t stream = nullptr;
  bool own_stream = true;
  static constexpr int N_WS = 4;                                // main stream + three MSM workspace streams
  static constexpr int N_WS_MSM = N_WS;                         // workspaces the batched MSM entry points / Marlin rotate over
  zkp::MsmWorkspace ws[N_WS];                                   // ws[0].stream aliases `stream`
  zkp::DevBuf ntt_scratch;
  hipEvent_t ev_fork = nullptr, ev_a = nullptr, ev_b1 = nullptr;
  // pinned host landing zone of the async proof read-back
  uint32_t* host_proof = nullptr;                               // 256 words proof + 4 words flags; host tail: A | B | C as XYZZ
  bool host_tail = false;                                       // how the pending proof's points came back (prove_finish)
  bool busy = false;
};

// Resolved per-context configuration (include/zkp_accel.h zkp_ctx_config): filled from the environment when the context is created
// (the A/B variables stay the defaults), then overridden field by field by zkp_ctx_create_ex.  Read through ctx->cfg everywhere:
// no prover switch is process-global.
struct zkp_cfg {
  int lanes = 0;                      // 0: 8, 4 above 2^22 (groth16.hip prove_batch)
  int msm_batch_lanes = 1;
  int msm_c = 0, msm_c_g2 = 0;        // 0: round(log2 n) <= 20 (msm.hip pick_window_bits)
  long long msm_chunk = (long long)3 << 19;   // 0: never chunk
  double table_budget_gb = 0;         // 0: free device memory minus a quarter of the device
  bool h_lagrange = true, c_fold = true, host_affine = true;
  long long lfold_heavy_cost = 0;     // 0: chosen per key (groth16.hip fold_c_into_l)
  int multi_exchange = 0;             // ZKP_EXCHANGE_*
  int multi_exchange_timeout_ms = 30000;
  int multi_wm_split = -1;            // -1: measured per key; 0 / 1 forced
};

struct zkp_ctx {
  int device = 0;
  zkp_cfg cfg;
  // Every C-ABI entry point that takes this context holds this lock for the whole call (capi.hip guarded()): two host threads entering
  // ONE context are serialised instead of interleaving its lanes / scratch (since ABI 0.5; the documented contract stays "one ctx per
  // prover thread" — that is what runs concurrently).  Recursive: internal helpers may re-enter through a public entry point.
  std::recursive_mutex mu;
  static constexpr int N_LANES = 8;        // allocated; zkp_groth16_prove_batch_dev uses 8 (domains <= 2^22) or 4 of them unless ZKP_LANES says otherwise
  static constexpr int N_WS = zkp_lane::N_WS;
  zkp_lane lanes[N_LANES];
  zkp_lane* cur = &lanes[0];
  int cur_idx = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // zkp_timer_*
  hipEvent_t ev2 = nullptr, ev3 = nullptr;   // internal per-kernel timing
  bool profiling = false;
  size_t table_bytes = 0;      // resident window tables of this context (ZKP_TABLE_BUDGET_GB accounting)
  // Bucket chaining (round 3): two MSMs whose results are only ever ADDED (Groth16: L and H, C = ... + l' + h_acc) share ONE bucket array
  // and ONE reduction.  msm_defer_reduce: the next MSM stops after its buckets are complete (output = the identity); msm_acc_into >= 0:
  // the next MSM accumulates on top of the buckets workspace `msm_acc_into` holds and reduces the sum.  Both reset by msm_run.
  bool msm_defer_reduce = false;
  int msm_acc_into = -1;
  int msm_bucket_ws = -1;      // >= 0: the next MSM keeps its buckets in the bucket array of that workspace (chunked MSMs, msm.hip msm_run)
  bool dbg_skip_k8 = false;    // ABLATION ONLY (ZKP_DEBUG_SKIP_K8_MASK): the next MSM skips its bucket reduction — wrong results, timing experiments
  bool batch_mode = false;     // inside zkp_groth16_prove_batch*: kernels are tuned for throughput of many proofs in flight, not latency
  std::map<std::pair<int, int>, zkp::NttTables> ntt_tables;   // (curve, log_n)
  zkp::DevBuf ntt_io, poly_tmp, poly_consts, spmv_list;
  // MSM scratch
  zkp::DevBuf msm_scalars, msm_misc, var_bases;
  uint32_t* pinned = nullptr;      // pinned host landing zone for batched MSM results
  size_t pinned_cap = 0;
  std::unordered_map<uint64_t, std::shared_ptr<zkp::BasesEntry>> bases;
  uint64_t next_handle = 1;
  zkp_groth16_timing last_timing{};
  zkp_marlin_timing last_marlin_timing{};
  std::string last_error;
  // Single-process multi-GPU (zkp_ctx_create_multi): the ROOT context lists one context per requested device in rank order
  // (devs[0] == this; device ids may repeat: several ranks on one GPU).  Empty on an ordinary context and on the members.
  std::vector<zkp_ctx*> devs;
  // variable-base reduction plans (msm.hip var_plan), owned by the context and released with it
  std::map<std::pair<int, int>, std::shared_ptr<void>> var_plans;
  // ZKP_TIMELINE=1 (diagnostics): events recorded on the streams of ONE blocking proof at phase boundaries, printed to stderr by
  // prove_finish as offsets from the first — where the chains of a proof end when no tracer slows the host's launches
  bool tl_on = false;
  std::string tl_tag;
  std::vector<std::pair<std::string, hipEvent_t>> tl;
  void mark(hipStream_t st, const char* what) {
    if (!tl_on) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    tl.emplace_back(tl_tag + what, e);
  }
};
