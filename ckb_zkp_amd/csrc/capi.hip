// extern "C" boundary (include/zkp_accel.h): argument checking, exception -> status mapping, host<->device
// staging for the host-pointer variants.  No arithmetic lives here.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ctx.hpp"
#include "host_field.hpp"
#include "internal.hpp"

using namespace zkp;

namespace {

template <class Fn>
int32_t guarded(zkp_ctx* ctx, Fn&& fn) {
  if (!ctx) return ZKP_ERR_BAD_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);          // one call at a time per context (ctx.hpp)
  try {
    ZKP_HIP(hipSetDevice(ctx->device));
    fn();
    return ZKP_OK;
  } catch (const StatusError& e) {
    return e.status;
  } catch (const HipError& e) {
    ctx->last_error = std::string(e.what) + ": " + hipGetErrorString(e.e);
    fprintf(stderr, "[zkp_accel] HIP error %s at line %d: %s\n", hipGetErrorString(e.e), e.line, e.what);
    return e.e == hipErrorOutOfMemory ? ZKP_ERR_OOM : ZKP_ERR_DEVICE;
  } catch (const std::bad_alloc&) {
    return ZKP_ERR_OOM;
  } catch (...) {
    return ZKP_ERR_DEVICE;
  }
}

// The multi-GPU entry points drive every member context from worker threads through internal functions: they hold the lock of the
// root AND of every member (rank order: no two callers can take them in opposite orders) for the whole call, so a second thread that
// enters a member obtained from zkp_ctx_device waits instead of interleaving with a running multi-prove.
template <class Fn>
int32_t guarded_multi(zkp_ctx* root, Fn&& fn) {
  if (!root) return ZKP_ERR_BAD_ARG;
  std::vector<std::unique_lock<std::recursive_mutex>> member_locks;
  std::unique_lock<std::recursive_mutex> root_lock(root->mu);
  for (size_t k = 1; k < root->devs.size(); k++) member_locks.emplace_back(root->devs[k]->mu);
  return guarded(root, fn);
}

zkp_cfg cfg_from_env() {
  zkp_cfg c;
  auto num = [](const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; };
  auto off = [](const char* name) { const char* e = getenv(name); return e && atoi(e) == 0; };
  c.lanes = (int)num("ZKP_LANES", 0);
  c.msm_batch_lanes = (int)std::max<long long>(1, std::min<long long>(num("ZKP_BATCH_LANES", 1), zkp_ctx::N_LANES));
  c.msm_c = (int)num("ZKP_MSM_C", 0);
  c.msm_c_g2 = (int)num("ZKP_MSM_C_G2", 0);
  if (const char* e = getenv("ZKP_MSM_CHUNK")) c.msm_chunk = (long long)strtoull(e, nullptr, 0);
  if (const char* e = getenv("ZKP_TABLE_BUDGET_GB")) c.table_budget_gb = atof(e);
  c.h_lagrange = !off("ZKP_H_LAGRANGE");
  c.c_fold = !off("ZKP_C_FOLD");
  c.host_affine = !off("ZKP_HOST_AFFINE");
  c.lfold_heavy_cost = num("ZKP_LFOLD_HEAVY_COST", c.lfold_heavy_cost);
  if (const char* e = getenv("ZKP_MULTI_EXCHANGE")) c.multi_exchange = !strcmp(e, "rccl") ? ZKP_EXCHANGE_RCCL : !strcmp(e, "peer") ? ZKP_EXCHANGE_PEER : ZKP_EXCHANGE_AUTO;
  c.multi_exchange_timeout_ms = (int)num("ZKP_MULTI_EXCHANGE_TIMEOUT_MS", c.multi_exchange_timeout_ms);
  c.multi_wm_split = (int)num("ZKP_MULTI_WM_SPLIT", -1);
  return c;
}

// zkp_ctx_config (caller's view: 0 = default) -> zkp_cfg (resolved), on top of the environment defaults
int32_t apply_config(zkp_cfg* c, const zkp_ctx_config* u) {
  if (!u) return ZKP_OK;
  if (u->struct_size < sizeof(uint32_t) * 2) return ZKP_ERR_BAD_ARG;
  zkp_ctx_config f{};                                            // fields beyond the caller's struct_size stay 0 = default
  memcpy(&f, u, std::min<size_t>(u->struct_size, sizeof f));
  auto tri_ok = [](int32_t v) { return v == ZKP_DEFAULT || v == ZKP_ON || v == ZKP_OFF; };
  if (f.lanes < 0 || f.lanes > zkp_ctx::N_LANES || f.msm_batch_lanes < 0 || f.msm_batch_lanes > zkp_ctx::N_LANES) return ZKP_ERR_BAD_ARG;
  for (int32_t b : {f.msm_window_bits, f.msm_window_bits_g2})
    if (b != 0 && (b < 2 || b > 22)) return ZKP_ERR_BAD_ARG;
  if (f.msm_chunk_points < -1 || (f.msm_chunk_points > 0 && f.msm_chunk_points < 1024)) return ZKP_ERR_BAD_ARG;
  if (!(f.table_budget_gb >= 0.0) || f.c_fold_heavy_cost < 0 || f.multi_exchange_timeout_ms < 0) return ZKP_ERR_BAD_ARG;
  if (!tri_ok(f.h_evaluation_form) || !tri_ok(f.c_fold) || !tri_ok(f.host_affine) || !tri_ok(f.multi_witness_split)) return ZKP_ERR_BAD_ARG;
  if (f.multi_exchange < ZKP_EXCHANGE_AUTO || f.multi_exchange > ZKP_EXCHANGE_PEER) return ZKP_ERR_BAD_ARG;
  if (f.lanes) c->lanes = f.lanes;
  if (f.msm_batch_lanes) c->msm_batch_lanes = f.msm_batch_lanes;
  if (f.msm_window_bits) c->msm_c = f.msm_window_bits;
  if (f.msm_window_bits_g2) c->msm_c_g2 = f.msm_window_bits_g2;
  if (f.msm_chunk_points) c->msm_chunk = f.msm_chunk_points < 0 ? 0 : f.msm_chunk_points;
  if (f.table_budget_gb > 0.0) c->table_budget_gb = f.table_budget_gb;
  if (f.h_evaluation_form) c->h_lagrange = f.h_evaluation_form == ZKP_ON;
  if (f.c_fold) c->c_fold = f.c_fold == ZKP_ON;
  if (f.host_affine) c->host_affine = f.host_affine == ZKP_ON;
  if (f.c_fold_heavy_cost) c->lfold_heavy_cost = f.c_fold_heavy_cost;
  if (f.multi_exchange) c->multi_exchange = f.multi_exchange;
  if (f.multi_exchange_timeout_ms) c->multi_exchange_timeout_ms = f.multi_exchange_timeout_ms;
  if (f.multi_witness_split) c->multi_wm_split = f.multi_witness_split == ZKP_ON ? 1 : 0;
  return ZKP_OK;
}

}  // namespace

extern "C" {

const char* zkp_status_string(int32_t s) {
  switch (s) {
    case ZKP_OK: return "ok";
    case ZKP_ERR_BAD_ARG: return "bad argument";
    case ZKP_ERR_UNSUPPORTED_CURVE: return "unsupported curve";
    case ZKP_ERR_DOMAIN_TOO_LARGE: return "domain too large (PolynomialDegreeTooLarge)";
    case ZKP_ERR_OOM: return "out of device memory";
    case ZKP_ERR_DEVICE: return "HIP device error / no gfx950 device (there is no CPU fallback)";
    case ZKP_ERR_BAD_HANDLE: return "bad handle";
    case ZKP_ERR_INVALID_POINT: return "malformed point / not on the curve / not in the prime-order subgroup (InvalidData)";
    default: return "unknown status";
  }
}

// 0.2: ZKP_ERR_INVALID_POINT; partials slot 4 (L) = identity, slot 3 (H) = h + l (bucket chaining)
const char* zkp_version(void) { return "zkp_accel 0.6 (gfx950)"; }

int32_t zkp_ctx_get_config(zkp_ctx* ctx, zkp_ctx_config* out) {
  if (!ctx || !out) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    const zkp_cfg& c = ctx->cfg;
    zkp_ctx_config f{};
    f.struct_size = sizeof f;
    f.lanes = c.lanes;
    f.msm_batch_lanes = c.msm_batch_lanes;
    f.msm_window_bits = c.msm_c;
    f.msm_window_bits_g2 = c.msm_c_g2;
    f.msm_chunk_points = c.msm_chunk > 0 ? c.msm_chunk : -1;
    f.table_budget_gb = c.table_budget_gb;
    f.h_evaluation_form = c.h_lagrange ? ZKP_ON : ZKP_OFF;
    f.c_fold = c.c_fold ? ZKP_ON : ZKP_OFF;
    f.host_affine = c.host_affine ? ZKP_ON : ZKP_OFF;
    f.c_fold_heavy_cost = c.lfold_heavy_cost;
    f.multi_exchange = c.multi_exchange;
    f.multi_exchange_timeout_ms = c.multi_exchange_timeout_ms;
    f.multi_witness_split = c.multi_wm_split < 0 ? ZKP_DEFAULT : c.multi_wm_split ? ZKP_ON : ZKP_OFF;
    *out = f;
  });
}

int32_t zkp_ctx_create(zkp_ctx** out, int device_id) { return zkp_ctx_create_ex(out, device_id, nullptr); }

int32_t zkp_ctx_create_ex(zkp_ctx** out, int device_id, const zkp_ctx_config* user_cfg) {
  if (!out) return ZKP_ERR_BAD_ARG;
  *out = nullptr;
  zkp_cfg cfg = cfg_from_env();
  if (int32_t st = apply_config(&cfg, user_cfg); st != ZKP_OK) return st;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) {
    fprintf(stderr, "[zkp_accel] no usable HIP device (count=%d, requested %d); there is no CPU fallback\n", count,
            device_id);
    return ZKP_ERR_DEVICE;
  }
  // The proof pipeline keeps ~32 streams busy (8 lanes x 4).  ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES
  // hardware queues (default 4) and streams that share a queue serialise; 16 queues measured best on MI355X
  // (84 -> 95 proofs/s at 2^20).  The library does NOT touch the process environment: the host exports
  // GPU_MAX_HW_QUEUES=16 before its first HIP call (the Python package and bench.py do so).
  {
    static bool warned = false;
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    if (!warned && (!q || atoi(q) < 16)) {
      warned = true;
      fprintf(stderr, "[zkp_accel] GPU_MAX_HW_QUEUES is %s: the pipelined provers keep ~32 HIP streams busy and lose about 10 %% of "
                      "their throughput on the default 4 hardware queues — export GPU_MAX_HW_QUEUES=16 before the process makes its "
                      "first HIP call (INTEGRATION.md)\n", q ? q : "unset");
    }
  }
  zkp_ctx* ctx = new (std::nothrow) zkp_ctx();
  if (!ctx) return ZKP_ERR_OOM;
  ctx->device = device_id;
  ctx->cfg = cfg;
  int32_t st = guarded(ctx, [&] {
    ZKP_HIP(hipEventCreate(&ctx->ev0));
    ZKP_HIP(hipEventCreate(&ctx->ev1));
    ZKP_HIP(hipEventCreate(&ctx->ev2));
    ZKP_HIP(hipEventCreate(&ctx->ev3));
    // (HIP stream priorities per role — high for the G2 chain and the B1 -> s*g_a + r*g1_b chain, whose reduction tails end a single
    //  proof — were measured in round 4 and lose everywhere: 8.2 -> 8.7-14 ms per single proof, 143 -> 90-133 pipelined proofs/s.)
    for (int l = 0; l < zkp_ctx::N_LANES; l++) {
      zkp_lane& L = ctx->lanes[l];
      ZKP_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
      ZKP_HIP(hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming));
      ZKP_HIP(hipEventCreateWithFlags(&L.ev_a, hipEventDisableTiming));
      ZKP_HIP(hipEventCreateWithFlags(&L.ev_b1, hipEventDisableTiming));
      ZKP_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.host_proof), 1100));
      for (int i = 0; i < zkp_lane::N_WS; i++) {
        ZKP_HIP(hipEventCreateWithFlags(&L.ws[i].done, hipEventDisableTiming));
        ZKP_HIP(hipEventCreateWithFlags(&L.ws[i].sorted, hipEventDisableTiming));
        ZKP_HIP(hipEventCreateWithFlags(&L.ws[i].l1_done, hipEventDisableTiming));
        ZKP_HIP(hipEventCreateWithFlags(&L.ws[i].acc_done, hipEventDisableTiming));
        if (i > 0) {
          ZKP_HIP(hipStreamCreateWithFlags(&L.ws[i].stream, hipStreamNonBlocking));
          L.ws[i].own_stream = true;
        }
      }
    }
  });
  if (st != ZKP_OK) {
    (void)zkp_ctx_destroy(ctx);                 // releases whatever streams / events / pinned buffers already exist
    return st;
  }
  *out = ctx;
  return ZKP_OK;
}

int32_t zkp_ctx_create_multi(zkp_ctx** out, const int* device_ids, int n_devices) {
  return zkp_ctx_create_multi_ex(out, device_ids, n_devices, nullptr);
}

int32_t zkp_ctx_create_multi_ex(zkp_ctx** out, const int* device_ids, int n_devices, const zkp_ctx_config* cfg) {
  if (!out || !device_ids || n_devices < 1 || n_devices > 64) return ZKP_ERR_BAD_ARG;
  *out = nullptr;
  zkp_ctx* root = nullptr;
  int32_t st = zkp_ctx_create_ex(&root, device_ids[0], cfg);
  if (st != ZKP_OK) return st;
  root->devs.push_back(root);
  for (int k = 1; k < n_devices; k++) {
    zkp_ctx* m = nullptr;
    st = zkp_ctx_create_ex(&m, device_ids[k], cfg);
    if (st != ZKP_OK) {
      (void)zkp_ctx_destroy(root);
      return st;
    }
    root->devs.push_back(m);
  }
  // xGMI peer access between distinct devices (hipMemcpyPeerAsync then goes device to device instead of through the host)
  for (int a = 0; a < n_devices; a++)
    for (int b = 0; b < n_devices; b++) {
      if (device_ids[a] == device_ids[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) == hipSuccess && can) {
        (void)hipSetDevice(device_ids[a]);
        hipError_t e = hipDeviceEnablePeerAccess(device_ids[b], 0);
        if (e != hipSuccess) (void)hipGetLastError();        // already enabled
      }
    }
  (void)hipSetDevice(device_ids[0]);
  *out = root;
  return ZKP_OK;
}
int32_t zkp_ctx_num_devices(zkp_ctx* ctx, int32_t* n) {
  if (!ctx || !n) return ZKP_ERR_BAD_ARG;
  *n = ctx->devs.empty() ? 1 : (int32_t)ctx->devs.size();
  return ZKP_OK;
}
int32_t zkp_ctx_device(zkp_ctx* ctx, int32_t rank, zkp_ctx** member) {
  if (!ctx || !member) return ZKP_ERR_BAD_ARG;
  if (ctx->devs.empty()) {
    if (rank != 0) return ZKP_ERR_BAD_ARG;
    *member = ctx;
    return ZKP_OK;
  }
  if (rank < 0 || rank >= (int32_t)ctx->devs.size()) return ZKP_ERR_BAD_ARG;
  *member = ctx->devs[rank];
  return ZKP_OK;
}

int32_t zkp_ctx_destroy(zkp_ctx* ctx) {
  if (!ctx) return ZKP_ERR_BAD_ARG;
  { std::lock_guard<std::recursive_mutex> wait_for_calls_in_flight(ctx->mu); }     // (destroying a context another thread still uses stays the caller's bug)
  for (size_t k = 1; k < ctx->devs.size(); k++) (void)zkp_ctx_destroy(ctx->devs[k]);      // members of a multi-device root
  ctx->devs.clear();
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  ntt_free_tables(ctx);
  msm_free_all(ctx);
  for (hipEvent_t e : {ctx->ev0, ctx->ev1, ctx->ev2, ctx->ev3})
    if (e) (void)hipEventDestroy(e);
  for (int l = 0; l < zkp_ctx::N_LANES; l++) {
    zkp_lane& L = ctx->lanes[l];
    for (hipEvent_t e : {L.ev_fork, L.ev_a, L.ev_b1})
      if (e) (void)hipEventDestroy(e);
    if (L.host_proof) (void)hipHostFree(L.host_proof);
    for (int i = 0; i < zkp_lane::N_WS; i++) {
      if (L.ws[i].done) (void)hipEventDestroy(L.ws[i].done);
      if (L.ws[i].sorted) (void)hipEventDestroy(L.ws[i].sorted);
      if (L.ws[i].l1_done) (void)hipEventDestroy(L.ws[i].l1_done);
      if (L.ws[i].acc_done) (void)hipEventDestroy(L.ws[i].acc_done);
      if (L.ws[i].own_stream && L.ws[i].stream) (void)hipStreamDestroy(L.ws[i].stream);
    }
    if (L.own_stream && L.stream) (void)hipStreamDestroy(L.stream);
  }
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  delete ctx;
  return ZKP_OK;
}

int32_t zkp_ctx_set_stream(zkp_ctx* ctx, void* s) {
  return guarded(ctx, [&] {
    zkp_lane& L = ctx->lanes[0];
    ZKP_HIP(hipStreamSynchronize(L.stream));
    if (s) {
      if (L.own_stream && L.stream) (void)hipStreamDestroy(L.stream);
      L.stream = reinterpret_cast<hipStream_t>(s);
      L.own_stream = false;
    } else if (!L.own_stream) {
      ZKP_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
      L.own_stream = true;
    }
  });
}

int32_t zkp_ctx_sync(zkp_ctx* ctx) {
  return guarded(ctx, [&] { ZKP_HIP(hipStreamSynchronize(ctx->cur->stream)); });
}

int32_t zkp_dev_alloc(zkp_ctx* ctx, size_t bytes, void** dptr) {
  if (!dptr) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    *dptr = nullptr;
    if (hipMalloc(dptr, bytes ? bytes : 16) != hipSuccess) throw StatusError{ZKP_ERR_OOM};
  });
}
int32_t zkp_dev_free(zkp_ctx* ctx, void* dptr) {
  return guarded(ctx, [&] {
    ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
    if (dptr) ZKP_HIP(hipFree(dptr));
  });
}
int32_t zkp_h2d(zkp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    if (bytes) ZKP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->cur->stream));
    ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  });
}
int32_t zkp_d2h(zkp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    if (bytes) ZKP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->cur->stream));
    ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  });
}
int32_t zkp_timer_start(zkp_ctx* ctx) {
  return guarded(ctx, [&] { ZKP_HIP(hipEventRecord(ctx->ev0, ctx->cur->stream)); });
}
int32_t zkp_timer_stop_ms(zkp_ctx* ctx, float* ms) {
  if (!ms) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    ZKP_HIP(hipEventRecord(ctx->ev1, ctx->cur->stream));
    ZKP_HIP(hipEventSynchronize(ctx->ev1));
    ZKP_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  });
}
int32_t zkp_set_profiling(zkp_ctx* ctx, int32_t enable) {
  return guarded(ctx, [&] { ctx->profiling = enable != 0; });
}

// ------------------------------------------------------------------------------------------- NTT
int32_t zkp_ntt_dev(zkp_ctx* ctx, zkp_curve_t curve, uint64_t* data, uint32_t log_n, int32_t op) {
  if (!data) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    ZKP_REQUIRE(log_n <= 30, ZKP_ERR_DOMAIN_TOO_LARGE);
    ntt_run(ctx, curve, reinterpret_cast<uint32_t*>(data), (int)log_n, op);
  });
}
int32_t zkp_ntt(zkp_ctx* ctx, zkp_curve_t curve, uint64_t* data, uint32_t log_n, int32_t op) {
  if (!data) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    ZKP_REQUIRE(log_n <= 30, ZKP_ERR_DOMAIN_TOO_LARGE);
    ZKP_REQUIRE(curve == ZKP_BN254 || curve == ZKP_BLS12_381, ZKP_ERR_UNSUPPORTED_CURVE);
    ZKP_REQUIRE((int)log_n <= (curve == ZKP_BN254 ? 28 : 32), ZKP_ERR_DOMAIN_TOO_LARGE);
    size_t bytes = ((size_t)1 << log_n) * 32;
    uint32_t* d = ctx->ntt_io.as<uint32_t>(bytes / 4);
    ZKP_HIP(hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->cur->stream));
    ntt_run(ctx, curve, d, (int)log_n, op);
    ZKP_HIP(hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->cur->stream));
    ZKP_HIP(hipStreamSynchronize(ctx->cur->stream));
  });
}

// ------------------------------------------------------------------------------------------- bases / MSM
int32_t zkp_bases_upload_g1(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n,
                            uint64_t* handle) {
  if (!handle || (n && !xy)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *handle = bases_upload(ctx, curve, 1, xy, inf, n, /*c_hint=*/-1); });
}
int32_t zkp_bases_upload_g2(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n,
                            uint64_t* handle) {
  if (!handle || (n && !xy)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *handle = bases_upload(ctx, curve, 2, xy, inf, n, /*c_hint=*/-1); });
}
int32_t zkp_bases_share(zkp_ctx* dst, zkp_ctx* src, uint64_t src_handle, uint64_t* dst_handle) {
  if (!src || !dst_handle) return ZKP_ERR_BAD_ARG;
  return guarded(dst, [&] { *dst_handle = bases_share(dst, src, src_handle); });
}
int32_t zkp_bases_free(zkp_ctx* ctx, uint64_t handle) {
  return guarded(ctx, [&] { bases_free(ctx, handle); });
}
int32_t zkp_bases_len(zkp_ctx* ctx, uint64_t handle, size_t* n) {
  if (!n) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *n = bases_len(ctx, handle); });
}

static int32_t msm_common(zkp_ctx* ctx, int group, uint64_t handle, size_t offset, const uint64_t* scalars, size_t n,
                          uint64_t* out, bool scalars_on_device, bool montgomery) {
  if (!out || (n && !scalars)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    size_t avail = bases_len(ctx, handle);
    ZKP_REQUIRE(bases_group(ctx, handle) == group, ZKP_ERR_BAD_HANDLE);
    ZKP_REQUIRE(offset <= avail, ZKP_ERR_BAD_ARG);
    if (n > avail - offset) n = avail - offset;      // ark min(len) truncation
    const uint64_t* sdev = scalars;
    if (!scalars_on_device && n) {
      uint64_t* d = ctx->msm_scalars.as<uint64_t>(n * 4);
      ZKP_HIP(hipMemcpyAsync(d, scalars, n * 32, hipMemcpyHostToDevice, ctx->cur->stream));
      sdev = d;
    }
    msm_run(ctx, handle, offset, sdev, n, montgomery, out);
  });
}
int32_t zkp_msm_g1(zkp_ctx* ctx, uint64_t h, size_t off, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 1, h, off, s, n, out, false, false);
}
int32_t zkp_msm_g2(zkp_ctx* ctx, uint64_t h, size_t off, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 2, h, off, s, n, out, false, false);
}
int32_t zkp_msm_g1_dev(zkp_ctx* ctx, uint64_t h, size_t off, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 1, h, off, s, n, out, true, false);
}
int32_t zkp_msm_g2_dev(zkp_ctx* ctx, uint64_t h, size_t off, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 2, h, off, s, n, out, true, false);
}
int32_t zkp_vartime_multiscalar_mul_g1(zkp_ctx* ctx, uint64_t h, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 1, h, 0, s, n, out, false, true);
}

int32_t zkp_vartime_multiscalar_mul_g2(zkp_ctx* ctx, uint64_t h, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 2, h, 0, s, n, out, false, true);
}

int32_t zkp_msm_g1_mont_dev(zkp_ctx* ctx, uint64_t h, size_t off, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 1, h, off, s, n, out, true, true);
}

int32_t zkp_msm_g2_mont_dev(zkp_ctx* ctx, uint64_t h, size_t off, const uint64_t* s, size_t n, uint64_t* out) {
  return msm_common(ctx, 2, h, off, s, n, out, true, true);
}
int32_t zkp_msm_g1_var(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, const uint64_t* scalars,
                       size_t n, int32_t montgomery, uint64_t* out_xyz) {
  if (!out_xyz || (n && (!xy || !scalars))) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { msm_var_run(ctx, curve, 1, xy, inf, scalars, n, montgomery != 0, out_xyz); });
}
int32_t zkp_msm_g2_var(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, const uint64_t* scalars,
                       size_t n, int32_t montgomery, uint64_t* out_xyz) {
  if (!out_xyz || (n && (!xy || !scalars))) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { msm_var_run(ctx, curve, 2, xy, inf, scalars, n, montgomery != 0, out_xyz); });
}
int32_t zkp_msm_g1_mont_batch_dev(zkp_ctx* ctx, uint64_t h, size_t count, const size_t* offsets,
                                  const uint64_t* const* scalars_dev, const size_t* ns, uint64_t* out_xyz) {
  if (count && (!offsets || !scalars_dev || !ns || !out_xyz)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    ZKP_REQUIRE(bases_group(ctx, h) == 1, ZKP_ERR_BAD_HANDLE);
    msm_run_batch(ctx, h, count, offsets, scalars_dev, ns, true, out_xyz);
  });
}

int32_t zkp_msm_mont_multi_dev(zkp_ctx* ctx, size_t count, const uint64_t* handles, const size_t* offsets,
                               const uint64_t* const* scalars_dev, const size_t* ns, uint64_t* out_xyz,
                               size_t slot_u64) {
  if (count && (!handles || !offsets || !scalars_dev || !ns || !out_xyz)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { msm_run_multi(ctx, count, handles, offsets, scalars_dev, ns, true, out_xyz, 2 * slot_u64); });
}

// ------------------------------------------------------------------------------------------- Fr vectors / polynomials
int32_t zkp_fr_vec_op_dev(zkp_ctx* ctx, zkp_curve_t curve, int32_t op, const uint64_t* a, const uint64_t* b,
                          const uint64_t* k, uint64_t* out, size_t n) {
  if (n && (!a || !out)) return ZKP_ERR_BAD_ARG;
  if (n && op != ZKP_VEC_SCALE && op != ZKP_VEC_ADDC && !b) return ZKP_ERR_BAD_ARG;
  if ((op == ZKP_VEC_SCALE || op == ZKP_VEC_AXPY || op == ZKP_VEC_ADDC) && !k) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { fr_vec_op(ctx, curve, op, a, b, k, out, n); });
}
int32_t zkp_fr_spmv_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint32_t* row_ptr, const uint32_t* col,
                        const uint64_t* coeff, size_t nrows, const uint64_t* x, uint64_t* out) {
  if (nrows && (!row_ptr || !x || !out)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { fr_spmv(ctx, curve, row_ptr, col, coeff, nrows, x, out); });
}
int32_t zkp_fr_gather_dev(zkp_ctx* ctx, const uint64_t* in, const int32_t* idx, size_t n, uint64_t* out) {
  if (n && (!in || !idx || !out)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { fr_gather(ctx, in, idx, n, out); });
}
int32_t zkp_poly_divide_by_vanishing_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* p, size_t len, size_t n,
                                         uint64_t* q, uint64_t* rem) {
  if ((len && !p) || n == 0) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { poly_vanishing_fold(ctx, curve, p, len, n, q, rem); });
}
int32_t zkp_d2d(zkp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { if (bytes) ZKP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->cur->stream)); });
}
int32_t zkp_dev_zero(zkp_ctx* ctx, void* dst, size_t bytes) {
  if (bytes && !dst) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { if (bytes) ZKP_HIP(hipMemsetAsync(dst, 0, bytes, ctx->cur->stream)); });
}
int32_t zkp_fr_batch_inverse_dev(zkp_ctx* ctx, zkp_curve_t curve, uint64_t* v, size_t n) {
  if (n && !v) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { fr_batch_inverse(ctx, curve, v, n); });
}
int32_t zkp_poly_evaluate_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* p, size_t n, const uint64_t* z,
                              uint64_t* eval_out) {
  if ((n && !p) || !z || !eval_out) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { poly_div_linear(ctx, curve, p, n, z, nullptr, eval_out); });
}
int32_t zkp_poly_div_linear_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* p, size_t n, const uint64_t* z,
                                uint64_t* q, uint64_t* eval_out) {
  if ((n && !p) || !z || (n > 1 && !q) || q == p) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { poly_div_linear(ctx, curve, p, n, z, n > 1 ? q : nullptr, eval_out); });
}

int32_t zkp_g1_fold(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz, size_t k, uint64_t* out) {
  if (!out || (k && !xyz)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { point_fold(ctx, curve, 1, xyz, k, out); });
}
int32_t zkp_g2_fold(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz, size_t k, uint64_t* out) {
  if (!out || (k && !xyz)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { point_fold(ctx, curve, 2, xyz, k, out); });
}
int32_t zkp_g1_into_affine(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz, uint64_t* xy, uint8_t* inf) {
  if (!xyz || !xy || !inf) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { point_into_affine(ctx, curve, 1, xyz, xy, inf); });
}
int32_t zkp_g2_into_affine(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz, uint64_t* xy, uint8_t* inf) {
  if (!xyz || !xy || !inf) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { point_into_affine(ctx, curve, 2, xyz, xy, inf); });
}
int32_t zkp_groth16_points_into_affine(zkp_curve_t curve, const uint64_t* a_xyzz, const uint64_t* b_xyzz, const uint64_t* c_xyzz,
                                       uint64_t* proof_out, uint8_t* inf_out) {
  if (!a_xyzz || !b_xyzz || !c_xyzz || !proof_out || !inf_out) return ZKP_ERR_BAD_ARG;
  if (curve != ZKP_BN254 && curve != ZKP_BLS12_381) return ZKP_ERR_UNSUPPORTED_CURVE;
  const zkp::hostf::HostField Fq = zkp::hostf::fq_field(curve);
  zkp::hostf::groth16_points_into_affine(Fq, reinterpret_cast<const uint32_t*>(a_xyzz), reinterpret_cast<const uint32_t*>(b_xyzz),
                                         reinterpret_cast<const uint32_t*>(c_xyzz), reinterpret_cast<uint32_t*>(proof_out), inf_out);
  return ZKP_OK;
}
static int32_t decompress_any(zkp_ctx* ctx, zkp_curve_t curve, int group, const uint8_t* bytes, size_t n, uint64_t* xy,
                              uint8_t* inf, size_t* bad) {
  if (bad) *bad = SIZE_MAX;                       // only ZKP_ERR_INVALID_POINT carries an index
  if (n && (!bytes || !xy || !inf)) return ZKP_ERR_BAD_ARG;
  size_t st = 0;
  const int32_t rc = guarded(ctx, [&] { st = points_decompress(ctx, curve, group, bytes, n, xy, inf); });
  if (rc != ZKP_OK) return rc;
  if (st) {
    if (bad) *bad = st - 1;
    return ZKP_ERR_INVALID_POINT;
  }
  return ZKP_OK;
}
int32_t zkp_g1_decompress(zkp_ctx* ctx, zkp_curve_t curve, const uint8_t* bytes, size_t n, uint64_t* xy, uint8_t* inf, size_t* bad) {
  return decompress_any(ctx, curve, 1, bytes, n, xy, inf, bad);
}
int32_t zkp_g2_decompress(zkp_ctx* ctx, zkp_curve_t curve, const uint8_t* bytes, size_t n, uint64_t* xy, uint8_t* inf, size_t* bad) {
  return decompress_any(ctx, curve, 2, bytes, n, xy, inf, bad);
}
int32_t zkp_g1_compress(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* bytes) {
  if (n && (!xy || !bytes)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { points_compress(ctx, curve, 1, xy, inf, n, bytes); });
}
int32_t zkp_g2_compress(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* bytes) {
  if (n && (!xy || !bytes)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { points_compress(ctx, curve, 2, xy, inf, n, bytes); });
}
static int32_t subgroup_any(zkp_ctx* ctx, zkp_curve_t curve, int group, const uint64_t* xy, const uint8_t* inf, size_t n, size_t* bad) {
  if (bad) *bad = SIZE_MAX;
  if (n && !xy) return ZKP_ERR_BAD_ARG;
  size_t st = 0;
  const int32_t rc = guarded(ctx, [&] { st = points_subgroup_check(ctx, curve, group, xy, inf, n); });
  if (rc != ZKP_OK) return rc;
  if (st) {
    if (bad) *bad = st - 1;
    return ZKP_ERR_INVALID_POINT;
  }
  return ZKP_OK;
}
int32_t zkp_g1_subgroup_check(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, size_t* bad) {
  return subgroup_any(ctx, curve, 1, xy, inf, n, bad);
}
int32_t zkp_g2_subgroup_check(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, size_t* bad) {
  return subgroup_any(ctx, curve, 2, xy, inf, n, bad);
}
int32_t zkp_fixed_base_mul_g1(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* base, const uint64_t* scalars, size_t n,
                              uint64_t* out_xy, uint8_t* out_inf) {
  if (!base || (n && (!scalars || !out_xy || !out_inf))) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { fixed_base_mul(ctx, curve, 1, base, scalars, n, out_xy, out_inf); });
}
int32_t zkp_fixed_base_mul_g2(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* base, const uint64_t* scalars, size_t n,
                              uint64_t* out_xy, uint8_t* out_inf) {
  if (!base || (n && (!scalars || !out_xy || !out_inf))) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { fixed_base_mul(ctx, curve, 2, base, scalars, n, out_xy, out_inf); });
}

// ------------------------------------------------------------------------------------------- Groth16
int32_t zkp_groth16_pk_upload(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, zkp_groth16_pk** out) {
  if (!desc || !out) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = groth16_pk_upload(ctx, desc); });
}
int32_t zkp_groth16_pk_upload_ex(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, uint32_t flags, zkp_groth16_pk** out) {
  if (!desc || !out || (flags & ~(uint32_t)ZKP_PK_KEEP_FORM)) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = groth16_pk_upload(ctx, desc, 0, 0, (int)flags); });
}
int32_t zkp_groth16_pk_upload_shard(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, int32_t rank, int32_t world,
                                    zkp_groth16_pk** out) {
  if (!desc || !out || world < 1 || rank < 0 || rank >= world) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = groth16_pk_upload(ctx, desc, rank, world); });
}
int32_t zkp_groth16_partials_bytes(zkp_curve_t curve, size_t* bytes) {
  if (!bytes || (curve != ZKP_BN254 && curve != ZKP_BLS12_381)) return ZKP_ERR_BAD_ARG;
  *bytes = groth16_partials_bytes(curve);
  return ZKP_OK;
}
int32_t zkp_groth16_prove_partials_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_dev, const uint64_t* r,
                                       const uint64_t* s, void* partials_dev) {
  if (!pk || !z_dev || !r || !s || !partials_dev) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_prove_partials(ctx, pk, z_dev, r, s, partials_dev); });
}
int32_t zkp_groth16_fold_assemble_dev(zkp_ctx* ctx, zkp_curve_t curve, const void* gathered_dev, int32_t world,
                                      const uint64_t* r, const uint64_t* s, uint64_t* proof, uint8_t* inf) {
  if (!gathered_dev || world < 1 || !r || !s || !proof || !inf) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_fold_assemble(ctx, curve, gathered_dev, world, r, s, proof, inf); });
}
int32_t zkp_groth16_pk_upload_multi(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, int32_t mode, zkp_groth16_pk_multi** out) {
  if (!ctx || !desc || !out || ctx->devs.empty() || (mode != ZKP_MULTI_SHARD && mode != ZKP_MULTI_REPLICATE)) return ZKP_ERR_BAD_ARG;
  return guarded_multi(ctx, [&] { *out = groth16_pk_upload_multi(ctx, desc, mode); });
}
int32_t zkp_groth16_pk_multi_free(zkp_ctx* ctx, zkp_groth16_pk_multi* pk) {
  if (!ctx || !pk || ctx->devs.empty()) return ZKP_ERR_BAD_ARG;
  return guarded_multi(ctx, [&] { groth16_pk_multi_free(ctx, pk); });
}
int32_t zkp_groth16_multi_info(zkp_ctx* ctx, zkp_groth16_pk_multi* pk, uint64_t info[6]) {
  if (!ctx || !pk || !info || ctx->devs.empty()) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_multi_info(ctx, pk, info); });
}
int32_t zkp_groth16_prove_multi(zkp_ctx* ctx, zkp_groth16_pk_multi* pk, const uint64_t* const* z, int32_t z_on_device,
                                const uint64_t* r, const uint64_t* s, uint64_t* proof, uint8_t* inf) {
  if (!ctx || !pk || !z || !z[0] || !r || !s || !proof || !inf || ctx->devs.empty()) return ZKP_ERR_BAD_ARG;
  if (z_on_device)
    for (size_t k = 0; k < ctx->devs.size(); k++)
      if (!z[k]) return ZKP_ERR_BAD_ARG;
  return guarded_multi(ctx, [&] { groth16_prove_multi(ctx, pk, z, z_on_device != 0, r, s, proof, inf); });
}
int32_t zkp_groth16_prove_batch_multi(zkp_ctx* ctx, zkp_groth16_pk_multi* pk, size_t count, const uint64_t* const* z,
                                      int32_t z_on_device, const uint64_t* r, const uint64_t* s, uint64_t* proofs,
                                      uint8_t* inf) {
  if (!ctx || !pk || ctx->devs.empty() || (count && (!z || !r || !s || !proofs || !inf))) return ZKP_ERR_BAD_ARG;
  return guarded_multi(ctx, [&] { groth16_prove_batch_multi(ctx, pk, count, z, z_on_device != 0, r, s, proofs, inf); });
}
int32_t zkp_groth16_pk_free(zkp_ctx* ctx, zkp_groth16_pk* pk) {
  if (!pk) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_pk_free(ctx, pk); });
}
int32_t zkp_groth16_pk_info(zkp_ctx* ctx, zkp_groth16_pk* pk, uint64_t info[8]) {
  if (!ctx || !pk || !info) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_pk_info(ctx, pk, info); });
}
int32_t zkp_groth16_domain_size(zkp_groth16_pk* pk, uint64_t* n) {
  if (!pk || !n) return ZKP_ERR_BAD_ARG;
  *n = groth16_domain_size(pk);
  return ZKP_OK;
}
int32_t zkp_groth16_witness_map(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, uint64_t* h) {
  if (!pk || !z || !h) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_witness_map(ctx, pk, z, h, false); });
}
int32_t zkp_groth16_witness_map_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, uint64_t* h) {
  if (!pk || !z || !h) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_witness_map(ctx, pk, z, h, true); });
}
int32_t zkp_groth16_prove(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, const uint64_t* r, const uint64_t* s,
                          uint64_t* proof, uint8_t* inf) {
  if (!pk || !z || !r || !s || !proof || !inf) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_prove(ctx, pk, z, false, r, s, proof, inf); });
}
int32_t zkp_groth16_prove_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, const uint64_t* r,
                              const uint64_t* s, uint64_t* proof, uint8_t* inf) {
  if (!pk || !z || !r || !s || !proof || !inf) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_prove(ctx, pk, z, true, r, s, proof, inf); });
}
int32_t zkp_groth16_prove_batch_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, size_t n, const uint64_t* const* z_dev,
                                    const uint64_t* r, const uint64_t* s, uint64_t* proofs, uint8_t* inf) {
  if (!pk || (n && (!z_dev || !r || !s || !proofs || !inf))) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_prove_batch(ctx, pk, n, z_dev, r, s, proofs, inf); });
}
int32_t zkp_groth16_prove_batch(zkp_ctx* ctx, zkp_groth16_pk* pk, size_t n, const uint64_t* const* z_host,
                                const uint64_t* r, const uint64_t* s, uint64_t* proofs, uint8_t* inf) {
  if (!pk || (n && (!z_host || !r || !s || !proofs || !inf))) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_prove_batch(ctx, pk, n, z_host, r, s, proofs, inf, false); });
}
int32_t zkp_groth16_assemble(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* sums, const uint64_t* r,
                             const uint64_t* s, uint64_t* proof, uint8_t* inf) {
  if (!sums || !r || !s || !proof || !inf) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { groth16_assemble(ctx, curve, sums, r, s, proof, inf); });
}
int32_t zkp_marlin_index_upload(zkp_ctx* ctx, const zkp_marlin_index_desc* desc, zkp_marlin_index** out) {
  if (!desc || !out) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = marlin_index_upload(ctx, desc); });
}
int32_t zkp_marlin_index_free(zkp_ctx* ctx, zkp_marlin_index* index) {
  if (!index) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { marlin_index_free(ctx, index); });
}
int32_t zkp_marlin_index_info(const zkp_marlin_index* index, uint64_t info[6]) {
  if (!index || !info) return ZKP_ERR_BAD_ARG;
  marlin_index_info(index, info);
  return ZKP_OK;
}
int32_t zkp_marlin_index_commit(zkp_ctx* ctx, zkp_marlin_index* index, uint64_t powers_of_g, uint64_t* comms_xy,
                                uint8_t* inf) {
  if (!index || !comms_xy || !inf) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { marlin_index_commit(ctx, index, powers_of_g, comms_xy, inf); });
}
int32_t zkp_marlin_prove(zkp_ctx* ctx, zkp_marlin_index* index, uint64_t powers_of_g, uint64_t powers_of_gamma_g,
                         const uint8_t* ivk_bytes, size_t ivk_len, const uint64_t* x, const uint64_t* w, size_t n_w,
                         const zkp_marlin_rand* rnd, const uint64_t* fixed_challenges, zkp_marlin_proof* out) {
  if (!index || !x || (n_w && !w) || !rnd || !out || (!fixed_challenges && !ivk_bytes)) return ZKP_ERR_BAD_ARG;
  if (!rnd->w || !rnd->z_a || !rnd->z_b || !rnd->mask || !rnd->blind_w || !rnd->blind_z_a || !rnd->blind_z_b ||
      !rnd->blind_g_1 || !rnd->blind_shifted_g_1)
    return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    marlin_prove(ctx, index, powers_of_g, powers_of_gamma_g, ivk_bytes, ivk_len, x, w, n_w, rnd, fixed_challenges, out);
  });
}
int32_t zkp_bench_mulmod(zkp_ctx* ctx, zkp_curve_t curve, int32_t field, int32_t unsaturated, double* out) {
  if (!out || (curve != ZKP_BN254 && curve != ZKP_BLS12_381) || field < 0 || field > 1) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = bench_mulmod(ctx, curve, field, unsaturated != 0); });
}
int32_t zkp_bench_hbm_copy(zkp_ctx* ctx, size_t bytes, double* out) {
  if (!out) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = bench_hbm_copy(ctx, bytes); });
}
int32_t zkp_marlin_last_timing(zkp_ctx* ctx, zkp_marlin_timing* out) {
  if (!ctx || !out) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = ctx->last_marlin_timing; });
}
int32_t zkp_groth16_last_timing(zkp_ctx* ctx, zkp_groth16_timing* out) {
  if (!ctx || !out) return ZKP_ERR_BAD_ARG;
  return guarded(ctx, [&] { *out = ctx->last_timing; });
}

}  // extern "C"
