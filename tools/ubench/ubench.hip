// Micro-benchmarks that set the *binding* roof for this path on MI355X: integer-multiply issue rate
// (v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u32_u24), f64 FMA rate (candidate 52-bit-limb
// multiplier), 256-bit Montgomery multiplications per second, and a float4 copy for the achievable HBM rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../ckb_zkp_amd/csrc ubench.hip -o ubench && ./ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ZKP_INLINE_MUL
#include "field_dev.hpp"
using namespace zkp;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, ILP = 8;

__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[ILP];
  uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (uint64_t)(uint32_t)acc[i] * y + (acc[i] ^ x);   // v_mad_u64_u32 (+xor)
  }
  uint64_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad64_pure(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[ILP];
  uint32_t y = b + blockIdx.x;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (uint64_t)(uint32_t)acc[i] * y + acc[i];          // pure v_mad_u64_u32 chain
  }
  uint64_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  uint32_t y = (b + blockIdx.x) | 1;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = acc[i] * y + 1;                                    // v_mul_lo_u32 (+add)
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  uint32_t y = (b + blockIdx.x) | 0x80000001u;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __umulhi(acc[i] | 0x80000000u, y);
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  uint32_t y = (b + blockIdx.x) & 0xffffff;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __umul24(acc[i], y) + acc[i];                      // v_mad_u32_u24
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  uint32_t y = (b + blockIdx.x);
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (acc[i] + y) ^ (acc[i] >> 3);                       // 3 full-rate ops
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dfma(double* out, double a, double b) {
  double acc[ILP];
  double y = b + blockIdx.x * 1e-9;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __fma_rn(acc[i], y, 0.5);
  }
  double s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class P>
__global__ __launch_bounds__(256) void k_montmul(uint32_t* out, const uint32_t* in, int iters) {
  using F = Fp<P>;
  size_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::load(in + (t % 1024) * 8), b = F::load(in + ((t + 7) % 1024) * 8);
  F c = a, d = b;
  for (int i = 0; i < iters; i++) {   // two independent chains
    c = c * a;
    d = d * b;
  }
  (c + d).store(out + t * 8);
}
template <class P>
__global__ __launch_bounds__(256) void k_montmul_chain(uint32_t* out, const uint32_t* in, int iters) {
  using F = Fp<P>;
  size_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::load(in + (t % 1024) * 8);
  F c = a;
  for (int i = 0; i < iters; i++) c = c * c;      // dependent chain: latency
  c.store(out + t * 8);
}
__global__ void k_copy(float4* __restrict__ dst, const float4* __restrict__ src, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

template <class Fn>
static float time_ms(Fn&& fn, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  fn();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0);
    fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, CUs %d, clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  const int blocks = p.multiProcessorCount * 8, threads = 256;
  void* buf;
  CK(hipMalloc(&buf, (size_t)blocks * threads * 64));
  const double lanes_ops = (double)blocks * threads * ITERS * ILP;
  float ms;
  ms = time_ms([&] { hipLaunchKernelGGL(k_mad64_pure, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 3u, 5u); });
  printf("{\"bench\":\"v_mad_u64_u32\",\"Gops\":%.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 3u, 5u); });
  printf("{\"bench\":\"v_mad_u64_u32+2xor\",\"Gops\":%.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 3u, 5u); });
  printf("{\"bench\":\"v_mul_lo_u32+add\",\"Gops\":%.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_mulhi, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 3u, 5u); });
  printf("{\"bench\":\"v_mul_hi_u32+or\",\"Gops\":%.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_mad24, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 3u, 5u); });
  printf("{\"bench\":\"v_mad_u32_u24\",\"Gops\":%.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_add32, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 3u, 5u); });
  printf("{\"bench\":\"add+shift+xor (3 ops)\",\"Gops\":%.1f}\n", 3 * lanes_ops / ms * 1e-6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_dfma, dim3(blocks), dim3(threads), 0, 0, (double*)buf, 3.0, 1.0000001); });
  printf("{\"bench\":\"v_fma_f64\",\"Gops\":%.1f}\n", lanes_ops / ms * 1e-6);
  // Montgomery multiplication
  uint32_t* in;
  CK(hipMalloc(&in, 1024 * 32));
  CK(hipMemset(in, 0x5a, 1024 * 32));
  const int miters = 512;
  for (int occ = 1; occ <= 8; occ *= 2) {
    int mb = p.multiProcessorCount * occ;
    ms = time_ms([&] { hipLaunchKernelGGL(k_montmul<Bn254Fq>, dim3(mb), dim3(256), 0, 0, (uint32_t*)buf, in, miters); });
    printf("{\"bench\":\"montmul256 throughput\",\"blocks_per_cu\":%d,\"Gmulmod_s\":%.2f}\n", occ,
           (double)mb * 256 * miters * 2 / ms * 1e-6);
  }
  ms = time_ms([&] { hipLaunchKernelGGL(k_montmul_chain<Bn254Fq>, dim3(1), dim3(64), 0, 0, (uint32_t*)buf, in, 4096); });
  printf("{\"bench\":\"montmul256 dependent latency (1 wave)\",\"ns_per_mulmod\":%.1f}\n", ms * 1e6 / 4096);
  // HBM copy
  size_t n = (size_t)1 << 28;   // 4 GiB each way
  float4 *s, *d;
  CK(hipMalloc(&s, n * 16));
  CK(hipMalloc(&d, n * 16));
  CK(hipMemset(s, 1, n * 16));
  ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(p.multiProcessorCount * 16), dim3(256), 0, 0, d, s, n); });
  printf("{\"bench\":\"float4 copy\",\"GBps_read_plus_write\":%.1f}\n", 2.0 * n * 16 / ms * 1e-6);
  return 0;
}
