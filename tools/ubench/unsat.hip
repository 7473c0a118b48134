// Experiment: Montgomery multiplication on UNSATURATED limbs (9 x 29 bit for the 254-bit BN254 fields, 14 x 28 bit for
// the 381-bit BLS12-381 base field) versus the saturated product-scanning multiplier of field_dev.hpp.
// With limbs < 2^29 a column of the product scan (<= 18 partial products < 2^58) fits a 64-bit accumulator, so each
// 32x32 product is ONE v_mad_u64_u32 — no v_addc_co_u32 carry bank, no per-column register shuffling — at the price of
// (L/N)^2 more products.  R' = 2^(L*B) leaves p/R' <= 2^-7, so results stay < 2p WITHOUT a final subtraction and
// additions can be left unreduced.  Saturated Montgomery values X = x*2^256 map to this form by a 5-bit (11-bit) shift:
// 32*X = x*2^261 (mod p).  Checks equality against field_dev.hpp and measures throughput.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>
#define ZKP_INLINE_MUL
#include "field_dev.hpp"
using namespace zkp;

template <class P, int L_, int B_>
struct Unsat {
  static constexpr int L = L_, B = B_, N = P::N;
  static constexpr uint32_t MASK = (1u << B) - 1;
  static constexpr int SHIFT = L * B - 32 * N;            // R'/R = 2^SHIFT
  // limb i of the modulus in B-bit limbs
  static constexpr uint32_t modl(int i) {
    int bit = i * B, w = bit >> 5, o = bit & 31;
    uint64_t lo = w < N ? P::MOD[w] : 0, hi = (w + 1) < N ? P::MOD[w + 1] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> o) & MASK);
  }
  static constexpr uint32_t ninv() {                      // -p^-1 mod 2^B
    uint32_t p0 = P::MOD[0], x = 1;
    for (int i = 0; i < 6; i++) x *= 2u - p0 * x;         // Newton: p0 * x == 1 mod 2^32
    return (0u - x) & MASK;
  }
};

template <class U>
struct UF {
  uint32_t v[U::L];
};

// saturated (8 x 32 / 12 x 32, value < p, Montgomery R = 2^(32N)) -> unsaturated, shifted left by `sh` bits
template <class U, class P>
__device__ __forceinline__ UF<U> to_unsat(const Fp<P>& a, int sh) {
  UF<U> r;
#pragma unroll
  for (int i = 0; i < U::L; i++) {
    int bit = i * U::B - sh;                              // bit position in a of limb i's LSB
    uint64_t w = 0;
    int wi = bit >> 5, o = bit & 31;                      // arithmetic shift: negative bit -> wi = -1
    uint32_t lo = (wi >= 0 && wi < P::N) ? a.v[wi] : 0, hi = (wi + 1 >= 0 && wi + 1 < P::N) ? a.v[wi + 1] : 0;
    w = ((uint64_t)hi << 32) | lo;
    r.v[i] = (uint32_t)(w >> o) & U::MASK;
  }
  return r;
}
// unsaturated with normalised limbs (value < 2^(32N)) -> saturated words
template <class U, class P>
__device__ __forceinline__ Fp<P> to_sat(const UF<U>& a) {
  Fp<P> r;
#pragma unroll
  for (int w = 0; w < P::N; w++) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < U::L; i++) {
      int sh = i * U::B - 32 * w;                         // limb i contributes bits [sh, sh + B) of word w
      if (sh > -U::B && sh < 32) acc |= sh >= 0 ? ((uint64_t)a.v[i] << sh) : ((uint64_t)a.v[i] >> (-sh));
    }
    r.v[w] = (uint32_t)acc;
  }
  return r;
}

// product scanning, one 64-bit accumulator, no carry bank; result limbs < 2^B except the top one; value < 2p for
// inputs whose product is < R' * p
template <class U, class P>
__device__ __forceinline__ UF<U> umul(const UF<U>& a, const UF<U>& b) {
  constexpr int L = U::L;
  uint32_t m[L];
  UF<U> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L) acc += (uint64_t)a.v[i] * b.v[j];
    }
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L && i < k) acc += (uint64_t)m[i] * U::modl(j);
    }
    if (k < L) {
      m[k] = ((uint32_t)acc * U::ninv()) & U::MASK;
      acc += (uint64_t)m[k] * U::modl(0);
    } else {
      r.v[k - L] = (uint32_t)acc & U::MASK;
    }
    acc >>= U::B;
  }
  r.v[L - 1] = (uint32_t)acc;
  return r;
}

template <class P, class U>
__global__ void k_check(const uint32_t* in, int n, int* bad) {
  using F = Fp<P>;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  F a = F::load(in + (size_t)t * 2 * P::N), b = F::load(in + (size_t)(t * 2 + 1) * P::N);
  a = F::reduce_once(a);
  b = F::reduce_once(b);
  F want = (a * b) * a;                                   // two chained products (the second sees a lazy input)
  UF<U> ua = to_unsat<U, P>(a, U::SHIFT), ub = to_unsat<U, P>(b, U::SHIFT);
  UF<U> w = umul<U, P>(umul<U, P>(ua, ub), ua);
  UF<U> back = umul<U, P>(w, to_unsat<U, P>(F::one(), 0));  // * 2^(32N) / R'  -> saturated Montgomery form, < 2p
  F got = F::reduce_once(to_sat<U, P>(back));
  if (got != want) atomicAdd(bad, 1);
}

// two accumulators per column (even / odd partial products): halves the dependent v_mad_u64_u32 chain
template <class U, class P>
__device__ __forceinline__ UF<U> umul2(const UF<U>& a, const UF<U>& b) {
  constexpr int L = U::L;
  uint32_t m[L];
  UF<U> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    uint64_t acc1 = 0;
    int par = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L) {
        if (par & 1) acc1 += (uint64_t)a.v[i] * b.v[j];
        else acc += (uint64_t)a.v[i] * b.v[j];
        par++;
      }
    }
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L && i < k) {
        if (par & 1) acc1 += (uint64_t)m[i] * U::modl(j);
        else acc += (uint64_t)m[i] * U::modl(j);
        par++;
      }
    }
    asm volatile("" : "+v"(acc1));                         // keep the two chains separate
    acc += acc1;
    if (k < L) {
      m[k] = ((uint32_t)acc * U::ninv()) & U::MASK;
      acc += (uint64_t)m[k] * U::modl(0);
    } else {
      r.v[k - L] = (uint32_t)acc & U::MASK;
    }
    acc >>= U::B;
  }
  r.v[L - 1] = (uint32_t)acc;
  return r;
}

template <class P, class U, int V>
__global__ __launch_bounds__(256) void k_chain(uint32_t* out, const uint32_t* in, int iters) {
  using F = Fp<P>;
  size_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::load(in + (t % 1024) * P::N);
  if (V == 0) {
    F c = a;
    for (int i = 0; i < iters; i++) c = c * a;
    c.store(out + t * P::N);
  } else {
    UF<U> ua = to_unsat<U, P>(a, U::SHIFT), c = ua;
    for (int i = 0; i < iters; i++) c = V == 1 ? umul<U, P>(c, ua) : umul2<U, P>(c, ua);
    (to_sat<U, P>(c)).store(out + t * P::N);
  }
}

template <class P, class U, int V>
__global__ __launch_bounds__(256) void k_tp(uint32_t* out, const uint32_t* in, int iters) {
  using F = Fp<P>;
  size_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::load(in + (t % 1024) * P::N), b = F::load(in + ((t + 7) % 1024) * P::N);
  if (V == 0) {
    F c = a, d = b;
    for (int i = 0; i < iters; i++) {
      c = c * a;
      d = d * b;
    }
    (c + d).store(out + t * P::N);
  } else {
    UF<U> ua = to_unsat<U, P>(a, U::SHIFT), ub = to_unsat<U, P>(b, U::SHIFT), c = ua, d = ub;
    for (int i = 0; i < iters; i++) {
      c = umul<U, P>(c, ua);
      d = umul<U, P>(d, ub);
    }
#pragma unroll
    for (int i = 0; i < U::L; i++) c.v[i] += d.v[i];
    (to_sat<U, P>(c)).store(out + t * P::N);
  }
}

template <class P, class U>
int run(const char* name) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n = 1 << 16;
  std::vector<uint32_t> h((size_t)n * 2 * P::N);
  uint64_t s = 88172645463325252ull;
  for (auto& w : h) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    w = (uint32_t)(s >> 16);
  }
  for (int i = 0; i < 2 * n; i++) h[(size_t)i * P::N + P::N - 1] &= (P::MOD[P::N - 1] >> 1);   // < p
  for (int j = 0; j < P::N; j++) {
    h[j] = 0;
    h[P::N + j] = P::MOD[j];
    h[2 * P::N + j] = j == P::N - 1 ? (P::MOD[j] >> 1) : 0xffffffffu;
  }
  h[P::N] -= 1;   // p - 1
  uint32_t* d;
  int* bad;
  void* out;
  hipMalloc(&d, h.size() * 4);
  hipMalloc(&bad, 4);
  hipMemset(bad, 0, 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_check<P, U>), dim3(n / 256), dim3(256), 0, 0, d, n, bad);
  int hb = -1;
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("%s (%d x %d bit): mismatches %d of %d\n", name, U::L, U::B, hb, n);
  int blocks = p.multiProcessorCount * 8;
  hipMalloc(&out, (size_t)blocks * 256 * P::N * 4);
  for (int occ = 1; occ <= 8; occ *= 2)
    for (int v = 0; v < 2; v++) {
      int bl = p.multiProcessorCount * occ;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL((k_tp<P, U, 0>), dim3(bl), dim3(256), 0, 0, (uint32_t*)out, d, 256);
        else hipLaunchKernelGGL((k_tp<P, U, 1>), dim3(bl), dim3(256), 0, 0, (uint32_t*)out, d, 256);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%s %s blocks/CU %d: %.2f Gmulmod/s\n", name, v ? "unsaturated" : "saturated asm", occ,
             (double)bl * 256 * 256 * 2 / best * 1e-6);
    }
  // single dependent chain per lane at low occupancy: what the bucket loop looks like to the scheduler
  for (int occ = 1; occ <= 4; occ *= 2)
    for (int v = 0; v < 3; v++) {
      int bl = p.multiProcessorCount * occ;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL((k_chain<P, U, 0>), dim3(bl), dim3(256), 0, 0, (uint32_t*)out, d, 512);
        else if (v == 1) hipLaunchKernelGGL((k_chain<P, U, 1>), dim3(bl), dim3(256), 0, 0, (uint32_t*)out, d, 512);
        else hipLaunchKernelGGL((k_chain<P, U, 2>), dim3(bl), dim3(256), 0, 0, (uint32_t*)out, d, 512);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%s single chain %s blocks/CU %d: %.2f Gmulmod/s\n", name,
             v == 0 ? "saturated asm" : v == 1 ? "unsaturated" : "unsaturated, 2 accumulators", occ,
             (double)bl * 256 * 512 / best * 1e-6);
    }
  return hb;
}

int main() {
  int a = run<Bn254Fq, Unsat<Bn254Fq, 9, 29>>("Bn254Fq");
  int c = run<Bls381Fq, Unsat<Bls381Fq, 14, 28>>("Bls381Fq");
  return (a | c) ? 1 : 0;
}
