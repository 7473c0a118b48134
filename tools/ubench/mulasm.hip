// Experiment: product-scanning Montgomery multiplication with an explicit (64-bit acc, overflow counter) pair and
// v_mad_u64_u32 + v_addc_co_u32 MACs in inline asm, vs the C (CIOS) version in field_dev.hpp.  Checks equality and speed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ZKP_INLINE_MUL
#include "field_dev.hpp"
using namespace zkp;

#define MACV(acc, ovf, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ovf) : "v"(x), "v"(y) : "vcc")
#define MACS(acc, ovf, x, ys) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ovf) : "v"(x), "s"(ys) : "vcc")

template <class P>
__device__ __forceinline__ Fp<P> mul_ps(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  uint32_t m[N];
  Fp<P> r;
  uint64_t acc = 0;
  uint32_t ovf = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      int j = k - i;
      if (j >= 0 && j < N) MACV(acc, ovf, a.v[i], b.v[j]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      int j = k - i;
      if (j >= 0 && j < N && i < k) { if (i < N) MACS(acc, ovf, m[i], P::MOD[j]); }
    }
    if (k < N) {
      m[k] = (uint32_t)acc * P::INV;
      MACS(acc, ovf, m[k], P::MOD[0]);
    } else {
      r.v[k - N] = (uint32_t)acc;
    }
    acc = (acc >> 32) | ((uint64_t)ovf << 32);
    ovf = 0;
  }
  r.v[N - 1] = (uint32_t)acc;
  return Fp<P>::reduce_once(r);
}

template <class P>
__device__ __forceinline__ Fp<P> mul_ps2(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  uint32_t m[N];
  Fp<P> r;
  uint64_t accA = 0, accB = 0;
  uint32_t ovfA = 0, ovfB = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
    int par = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      int j = k - i;
      if (j >= 0 && j < N) { if (par & 1) MACV(accB, ovfB, a.v[i], b.v[j]); else MACV(accA, ovfA, a.v[i], b.v[j]); par++; }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      int j = k - i;
      if (j >= 0 && j < N && i < k) { if (par & 1) MACS(accB, ovfB, m[i], P::MOD[j]); else MACS(accA, ovfA, m[i], P::MOD[j]); par++; }
    }
    // merge B into A
    asm("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %3, vcc\n\tv_addc_co_u32_e32 %4, vcc, %4, %5, vcc"
        : "+v"(*(uint32_t*)&accA), "+v"(*((uint32_t*)&accA + 1)), "+v"(*(uint32_t*)&accB), "+v"(*((uint32_t*)&accB + 1)), "+v"(ovfA), "+v"(ovfB) :: "vcc");
    accB = 0; ovfB = 0;
    if (k < N) {
      m[k] = (uint32_t)accA * P::INV;
      MACS(accA, ovfA, m[k], P::MOD[0]);
    } else {
      r.v[k - N] = (uint32_t)accA;
    }
    accA = (accA >> 32) | ((uint64_t)ovfA << 32);
    ovfA = 0;
  }
  r.v[N - 1] = (uint32_t)accA;
  return Fp<P>::reduce_once(r);
}

template <class P, int V>
__global__ void k_lat(uint32_t* out, const uint32_t* in, int iters) {
  using F = Fp<P>;
  size_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::load(in + (t % 1024) * P::N);
  F c = a;
  for (int i = 0; i < iters; i++) { if (V == 0) c = c * c; else if (V == 1) c = mul_ps<P>(c, c); else c = mul_ps2<P>(c, c); }
  c.store(out + t * P::N);
}

template <class P, int V>
__global__ __launch_bounds__(256) void k_tp(uint32_t* out, const uint32_t* in, int iters) {
  using F = Fp<P>;
  size_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::load(in + (t % 1024) * P::N), b = F::load(in + ((t + 7) % 1024) * P::N);
  F c = a, d = b;
  for (int i = 0; i < iters; i++) {
    if (V == 0) { c = c * a; d = d * b; } else if (V == 1) { c = mul_ps<P>(c, a); d = mul_ps<P>(d, b); } else { c = mul_ps2<P>(c, a); d = mul_ps2<P>(d, b); }
  }
  (c + d).store(out + t * P::N);
}
template <class P>
__global__ void k_check(const uint32_t* in, int n, int* bad) {
  using F = Fp<P>;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  F a = F::load(in + (size_t)t * 2 * P::N), b = F::load(in + (size_t)(t * 2 + 1) * P::N);
  a = F::reduce_once(a); b = F::reduce_once(b);
  F x = a * b, y = mul_ps<P>(a, b);
  F x2 = a * a, y2 = mul_ps2<P>(a, a), y3 = mul_ps2<P>(a, b);
  if (x != y || x2 != y2 || x != y3) atomicAdd(bad, 1);
}
template <class P>
int run(const char* name) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n = 1 << 16;
  std::vector<uint32_t> h((size_t)n * 2 * P::N);
  uint64_t s = 88172645463325252ull;
  for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
  for (int i = 0; i < 2 * n; i++) h[(size_t)i * P::N + P::N - 1] &= (P::MOD[P::N - 1] >> 1) | 0;   // < p
  // edge values
  for (int j = 0; j < P::N; j++) { h[j] = 0; h[P::N + j] = P::MOD[j]; h[2 * P::N + j] = 0xffffffffu & (j == P::N - 1 ? (P::MOD[j] >> 1) : 0xffffffffu); }
  h[P::N] -= 1;   // p - 1
  uint32_t* d; int* bad; void* out;
  hipMalloc(&d, h.size() * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check<P>, dim3(n / 256), dim3(256), 0, 0, d, n, bad);
  int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("%s: mismatches %d of %d\n", name, hb, 2 * n);
  int blocks = p.multiProcessorCount * 8; hipMalloc(&out, (size_t)blocks * 256 * P::N * 4);
  for (int v = 0; v < 3; v++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (v == 0) hipLaunchKernelGGL((k_lat<P, 0>), dim3(1), dim3(64), 0, 0, (uint32_t*)out, d, 4096);
    else if (v == 1) hipLaunchKernelGGL((k_lat<P, 1>), dim3(1), dim3(64), 0, 0, (uint32_t*)out, d, 4096);
    else hipLaunchKernelGGL((k_lat<P, 2>), dim3(1), dim3(64), 0, 0, (uint32_t*)out, d, 4096);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s variant %d: single-wave dependent latency %.0f ns/mulmod\n", name, v, ms * 1e6 / 4096);
  }
  for (int occ = 1; occ <= 8; occ *= 2) for (int v = 1; v < 3; v++) {
    int bl = p.multiProcessorCount * occ;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (v == 1) hipLaunchKernelGGL((k_tp<P, 1>), dim3(bl), dim3(256), 0, 0, (uint32_t*)out, d, 256);
      else hipLaunchKernelGGL((k_tp<P, 2>), dim3(bl), dim3(256), 0, 0, (uint32_t*)out, d, 256);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s variant %d blocks/CU %d: %.2f Gmulmod/s\n", name, v, occ, (double)bl * 256 * 256 * 2 / best * 1e-6);
  }
  for (int v = 0; v < 2; v++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL((k_tp<P, 0>), dim3(blocks), dim3(256), 0, 0, (uint32_t*)out, d, 256);
      else hipLaunchKernelGGL((k_tp<P, 1>), dim3(blocks), dim3(256), 0, 0, (uint32_t*)out, d, 256);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s %s: %.2f Gmulmod/s\n", name, v ? "asm product-scanning" : "C CIOS", (double)blocks * 256 * 256 * 2 / best * 1e-6);
  }
  return hb;
}
#include <vector>
int main() { int a = run<Bn254Fq>("Bn254Fq"); int c = run<Bls381Fq>("Bls381Fq"); return (a | c) ? 1 : 0; }
