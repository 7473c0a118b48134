// Measurement helpers behind bench.py's `valu_roof` entry ("mulmod/s vs a measured v_mad_u64_u32
// microbenchmark peak"): the sustained rate of the library's own Montgomery multipliers — the saturated product-scanning
// multiplier of field.cuh and the unsaturated-limb multiplier of unsat.cuh — with two independent dependency chains per
// lane and every CU saturated.  This is the roof the bucket-accumulation and NTT kernels are bound by (integer VALU),
// measured in the same process as the benchmark.  Not on any product path.
#include "field.cuh"
#include "internal.hpp"
#include "unsat.cuh"

namespace zkp {

template <class P, int UNSAT>
__global__ __launch_bounds__(256) void mulmod_rate_kernel(uint32_t* __restrict__ out, int iters) {
  using F = Fp<P>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::one(), b = F::one();
  a.v[0] ^= t * 2654435761u;                    // distinct, lane-dependent operands (values need not be reduced for a rate test)
  b.v[1] ^= t * 40503u + 1;
  a = F::reduce_once(a);
  b = F::reduce_once(b);
  if constexpr (UNSAT != 0) {
    using U = Fu<P>;
    const U ua = U::from_sat_reduced(a), ub = U::from_sat_reduced(b);
    U c = ua, d = ub;
    for (int i = 0; i < iters; i++) {
      c = U::mul(c, ua);
      d = U::mul(d, ub);
    }
    U s = U::add(c, d);
#pragma unroll
    for (int i = 0; i < P::N; i++) out[(size_t)t * P::N + i] = s.v[i];
  } else {
    F c = a, d = b;
    for (int i = 0; i < iters; i++) {
      c = c * a;
      d = d * b;
    }
    (c + d).store(out + (size_t)t * P::N);
  }
}

// -> 1e9 Montgomery products per second
double bench_mulmod(zkp_ctx* ctx, int curve, int field, bool unsaturated) {
  // field: 0 = Fr, 1 = Fq
  hipStream_t st = ctx->cur->stream;
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const int blocks = cus * 8, threads = 256, iters = 1500;
  uint32_t* out = ctx->msm_misc.as<uint32_t>((size_t)blocks * threads * 12);
  auto launch = [&] {
    if (curve == ZKP_BN254 && field == 1) {
      if (unsaturated) hipLaunchKernelGGL((mulmod_rate_kernel<Bn254Fq, 1>), dim3(blocks), dim3(threads), 0, st, out, iters);
      else hipLaunchKernelGGL((mulmod_rate_kernel<Bn254Fq, 0>), dim3(blocks), dim3(threads), 0, st, out, iters);
    } else if (curve == ZKP_BN254) {
      ZKP_REQUIRE(!unsaturated, ZKP_ERR_BAD_ARG);
      hipLaunchKernelGGL((mulmod_rate_kernel<Bn254Fr, 0>), dim3(blocks), dim3(threads), 0, st, out, iters);
    } else if (field == 1) {
      if (unsaturated) hipLaunchKernelGGL((mulmod_rate_kernel<Bls381Fq, 1>), dim3(blocks), dim3(threads), 0, st, out, iters);
      else hipLaunchKernelGGL((mulmod_rate_kernel<Bls381Fq, 0>), dim3(blocks), dim3(threads), 0, st, out, iters);
    } else {
      ZKP_REQUIRE(!unsaturated, ZKP_ERR_BAD_ARG);
      hipLaunchKernelGGL((mulmod_rate_kernel<Bls381Fr, 0>), dim3(blocks), dim3(threads), 0, st, out, iters);
    }
  };
  launch();                                                        // warm-up (code load, clocks)
  ZKP_HIP(hipEventRecord(ctx->ev2, st));
  launch();
  ZKP_HIP(hipEventRecord(ctx->ev3, st));
  ZKP_HIP(hipEventSynchronize(ctx->ev3));
  ZKP_HIP(hipGetLastError());
  float ms = 0.f;
  ZKP_HIP(hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3));
  return (double)blocks * threads * iters * 2.0 / (ms * 1e-3) / 1e9;
}

}  // namespace zkp
