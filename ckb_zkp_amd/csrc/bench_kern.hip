// Measurement helpers behind bench.py's `valu_roof` entry ("mulmod/s vs a measured v_mad_u64_u32
// microbenchmark peak"): the sustained rate of the library's own Montgomery multipliers — the saturated product-scanning
// multiplier of field_dev.hpp and the unsaturated-limb multiplier of unsat_dev.hpp — with four independent dependency chains per
// lane and every CU saturated.  This is the roof the bucket-accumulation and NTT kernels are bound by (integer VALU),
// measured in the same process as the benchmark.  Not on any product path.
#include <algorithm>

#include "field_dev.hpp"
#include "internal.hpp"
#include "unsat_dev.hpp"

namespace zkp {

template <class P, int UNSAT, int CH>               // CH independent dependency chains per lane (the EC formulas offer 2-4)
__global__ __launch_bounds__(256) void mulmod_rate_kernel(uint32_t* __restrict__ out, int iters) {
  using F = Fp<P>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a[CH];
#pragma unroll
  for (int k = 0; k < CH; k++) {
    a[k] = F::one();
    a[k].v[k & 1] ^= t * (2654435761u + 40503u * k) + k;   // distinct, lane-dependent operands
    a[k] = F::reduce_once(a[k]);
  }
  if constexpr (UNSAT != 0) {
    using U = Fu<P>;
    U ua[CH], c[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) c[k] = ua[k] = U::from_sat_reduced(a[k]);
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < CH; k++) c[k] = U::mul(c[k], ua[k]);
    }
    U s = c[0];
#pragma unroll
    for (int k = 1; k < CH; k++) s = U::add(s, c[k]);
#pragma unroll
    for (int i = 0; i < P::N; i++) out[(size_t)t * P::N + i] = s.v[i];
  } else {
    F c[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) c[k] = a[k];
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < CH; k++) c[k] = c[k] * a[k];
    }
    F s = c[0];
#pragma unroll
    for (int k = 1; k < CH; k++) s = s + c[k];
    s.store(out + (size_t)t * P::N);
  }
}

// -> 1e9 Montgomery products per second
double bench_mulmod(zkp_ctx* ctx, int curve, int field, bool unsaturated) {
  // field: 0 = Fr, 1 = Fq
  hipStream_t st = ctx->cur->stream;
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const int blocks = cus * 8, threads = 256, iters = 800;
  uint32_t* out = ctx->msm_misc.as<uint32_t>((size_t)blocks * threads * 12);
  // two variants (2 and 4 chains per lane): the 14-limb BLS12-381 operands spill with 4 chains, the 8/9-limb BN254 ones gain
  // from them; the ceiling is the better of the two
  double best = 0.0;
  for (int ch : {2, 4}) {
    auto launch = [&] {
#define ZKP_RATE(PP, UU)                                                                                             \
  do {                                                                                                               \
    if (ch == 2) hipLaunchKernelGGL((mulmod_rate_kernel<PP, UU, 2>), dim3(blocks), dim3(threads), 0, st, out, iters); \
    else hipLaunchKernelGGL((mulmod_rate_kernel<PP, UU, 4>), dim3(blocks), dim3(threads), 0, st, out, iters);         \
  } while (0)
      if (curve == ZKP_BN254 && field == 1) {
        if (unsaturated) ZKP_RATE(Bn254Fq, 1);
        else ZKP_RATE(Bn254Fq, 0);
      } else if (curve == ZKP_BN254) {
        if (unsaturated) ZKP_RATE(Bn254Fr, 1);                       // the NTT products since round 3
        else ZKP_RATE(Bn254Fr, 0);
      } else if (field == 1) {
        if (unsaturated) ZKP_RATE(Bls381Fq, 1);
        else ZKP_RATE(Bls381Fq, 0);
      } else {
        if (unsaturated) ZKP_RATE(Bls381Fr, 1);
        else ZKP_RATE(Bls381Fr, 0);
      }
#undef ZKP_RATE
    };
    launch();                                                        // warm-up (code load, clocks)
    ZKP_HIP(hipEventRecord(ctx->ev2, st));
    launch();
    ZKP_HIP(hipEventRecord(ctx->ev3, st));
    ZKP_HIP(hipEventSynchronize(ctx->ev3));
    ZKP_HIP(hipGetLastError());
    float ms = 0.f;
    ZKP_HIP(hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3));
    best = std::max(best, (double)blocks * threads * iters * (double)ch / (ms * 1e-3) / 1e9);
  }
  return best;
}

// "also measure an on-box copy kernel and quote both" — the HBM bandwidth a plain streaming copy reaches on THIS
// device, next to the nominal 8 TB/s: every lane moves 16 B per iteration (global_load_dwordx4 / global_store_dwordx4), a
// grid-stride loop over a buffer much larger than the 256 MB of L2 + Infinity Cache, read once and written once.
typedef uint32_t copy_v4 __attribute__((ext_vector_type(4)));
// U independent 16-B loads in flight per lane before the first store; NT: non-temporal loads / stores (no L2 allocation)
template <int U, bool NT>
__global__ __launch_bounds__(256) void hbm_copy_kernel(const copy_v4* __restrict__ src, copy_v4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    copy_v4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride);
      else dst[i + k * stride] = v[k];
    }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

// -> GB/s (read + written bytes over the HIP-event time of `reps` back-to-back copies of `bytes` bytes)
double bench_hbm_copy(zkp_ctx* ctx, size_t bytes) {
  hipStream_t st = ctx->cur->stream;
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  bytes &= ~(size_t)0xFFF;
  ZKP_REQUIRE(bytes >= (1u << 20), ZKP_ERR_BAD_ARG);
  copy_v4 *src = nullptr, *dst = nullptr;
  ZKP_HIP(hipMalloc(&src, bytes));
  if (hipMalloc(&dst, bytes) != hipSuccess) {
    (void)hipFree(src);
    throw StatusError{ZKP_ERR_OOM};
  }
  const size_t n16 = bytes / 16;
  double best = 0.0;
  try {
    ZKP_HIP(hipMemsetAsync(src, 0x5a, bytes, st));
    ZKP_HIP(hipMemsetAsync(dst, 0, bytes, st));
    // the best of (grid, loads in flight, cache policy) is the device's figure
    for (int variant = 0; variant < 4; variant++) {
      for (int wg_per_cu : {4, 8, 16, 32}) {
        const int blocks = cus * wg_per_cu, reps = 5;
        auto launch = [&] {
          switch (variant) {
            case 0: hipLaunchKernelGGL((hbm_copy_kernel<1, false>), dim3(blocks), dim3(256), 0, st, src, dst, n16); break;
            case 1: hipLaunchKernelGGL((hbm_copy_kernel<4, false>), dim3(blocks), dim3(256), 0, st, src, dst, n16); break;
            case 2: hipLaunchKernelGGL((hbm_copy_kernel<1, true>), dim3(blocks), dim3(256), 0, st, src, dst, n16); break;
            default: hipLaunchKernelGGL((hbm_copy_kernel<4, true>), dim3(blocks), dim3(256), 0, st, src, dst, n16); break;
          }
        };
        launch();                                                       // warm-up
        ZKP_HIP(hipEventRecord(ctx->ev2, st));
        for (int r = 0; r < reps; r++) launch();
        ZKP_HIP(hipEventRecord(ctx->ev3, st));
        ZKP_HIP(hipEventSynchronize(ctx->ev3));
        ZKP_HIP(hipGetLastError());
        float ms = 0.f;
        ZKP_HIP(hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3));
        best = std::max(best, 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9);
      }
    }
  } catch (...) {
    (void)hipFree(src);
    (void)hipFree(dst);
    throw;
  }
  (void)hipFree(src);
  (void)hipFree(dst);
  return best;
}

}  // namespace zkp
