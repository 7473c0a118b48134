// Experiment: the constant-operand half of a Montgomery reduction on the int8 matrix cores.
//
// In the product scan of unsat_dev.hpp half of the 162 v_mad_u64_u32 of a 254-bit Montgomery product multiply by the CONSTANT
// modulus: U = m * p with m the 261-bit Montgomery factor.  A product by a constant is a matrix product with the Toeplitz matrix of
// the constant's digits, so 64 field elements of a wave can go through V_MFMA_I32_16X16X64_I8:
//     digits radix 2^7 (int8 operands are signed; 7-bit digits stay non-negative): m = 38 digits, p = 37 digits, U = 75 columns
//     A (constant) = 5 row tiles of the 75 x 64 Toeplitz matrix T[i][k] = p_digit[i - k];  B = digits of 16 elements per column tile
//     20 MFMAs (5 row tiles x 4 column tiles) per wave, column sums < 38 * 127 * 127 < 2^20 in i32
// What it costs around the MFMAs is the point of the experiment: the kernels keep ONE ELEMENT PER LANE (limbs in VGPRs), the MFMA
// wants a column of one element spread over four lanes, and the 75 column sums must be carried back into 29-bit limbs:
//     limbs -> 38 digits (VALU) -> LDS -> B fragments (ds_read_b128) ; D fragments -> LDS -> 75 sums per lane -> carry chain (VALU)
// Variant V is the same U = m * p as 81 v_mad_u64_u32 (product scan, 29-bit limbs), i.e. what unsat_dev.hpp does today.
// Both are checked against each other limb for limb (V is the library's product scan) and timed on every CU.
//
//   hipcc -O3 --offload-arch=gfx950 -I ckb_zkp_amd/csrc -o tools/ubench/mfma_redc tools/ubench/mfma_redc.hip && tools/ubench/mfma_redc
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>
#define ZKP_INLINE_MUL
#include "field_dev.hpp"
using namespace zkp;

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int L = 9, B = 29, ND = 38, NP = 37, NC = 75;      // limbs, limb bits, digits of m / p, columns of U
constexpr uint32_t MASK = (1u << B) - 1;

__host__ __device__ constexpr uint32_t p_limb(int i) {        // 29-bit limb i of the BN254 base-field modulus
  int bit = i * B, w = bit >> 5, o = bit & 31;
  uint64_t lo = w < 8 ? Bn254Fq::MOD[w] : 0, hi = w + 1 < 8 ? Bn254Fq::MOD[w + 1] : 0;
  return (uint32_t)(((lo | (hi << 32)) >> o) & MASK);
}
__host__ __device__ constexpr int p_digit(int i) {            // 7-bit digit i of the modulus (0 outside [0, NP))
  if (i < 0 || i >= NP) return 0;
  int bit = i * 7, w = bit >> 5, o = bit & 31;
  uint64_t lo = w < 8 ? Bn254Fq::MOD[w] : 0, hi = w + 1 < 8 ? Bn254Fq::MOD[w + 1] : 0;
  return (int)(((lo | (hi << 32)) >> o) & 127);
}

// ---- variant V: U = m * p, product scan on 29-bit limbs (81 v_mad_u64_u32), 18 limbs out
__device__ __forceinline__ void mulp_valu(const uint32_t* m, uint32_t* u) {
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L) acc += (uint64_t)m[i] * p_limb(j);
    }
    u[k] = (uint32_t)acc & MASK;
    acc >>= B;
  }
  u[2 * L - 1] = (uint32_t)acc;
}

// ---- variant M: the same through V_MFMA_I32_16X16X64_I8
// LDS per wave: digits [64 elements][64 bytes] (4 KiB) + column sums [64 elements][80 i32] (20 KiB)
constexpr int DIG_STRIDE = 64, COL_STRIDE = 80;
struct MfmaConst {
  v4i a[5];                                                     // the constant operand: 5 row tiles of the Toeplitz matrix
};
__device__ __forceinline__ MfmaConst make_const() {
  // A fragment of row tile t: lane l supplies row i = 16 t + (l & 15), k-slots 16 (l >> 4) + b, b < 16 (B uses the same slots)
  MfmaConst c;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < 5; t++) {
    int w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 16; b++) {
      const int i = 16 * t + (lane & 15), k = 16 * (lane >> 4) + b;
      int d = 0;
#pragma unroll
      for (int q = 0; q < NP; q++) d = (i - k == q) ? p_digit(q) : d;
      w[b >> 2] |= d << (8 * (b & 3));
    }
    c.a[t] = v4i{w[0], w[1], w[2], w[3]};
  }
  return c;
}
__device__ __forceinline__ void mulp_mfma(const MfmaConst& c, const uint32_t* m, uint32_t* u, uint8_t* dig, int* col) {
  const int lane = threadIdx.x & 63;
  // (1) limbs -> 7-bit digits, 4 per dword, 10 dwords (40 digit slots, 38 used); the rest of the 64-byte row stays zero
  {
    uint32_t w[10];
#pragma unroll
    for (int q = 0; q < 10; q++) w[q] = 0;
#pragma unroll
    for (int d = 0; d < ND; d++) {
      const int bit = 7 * d, li = bit / B, o = bit % B;
      uint32_t x = m[li] >> o;
      if (o + 7 > B && li + 1 < L) x |= m[li + 1] << (B - o);
      w[d >> 2] |= (x & 127u) << (8 * (d & 3));
    }
    uint32_t* row = reinterpret_cast<uint32_t*>(dig + lane * DIG_STRIDE);
#pragma unroll
    for (int q = 0; q < 10; q++) row[q] = w[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // (2) 4 column tiles x 5 row tiles
#pragma unroll
  for (int nt = 0; nt < 4; nt++) {
    const v4i bfrag = *reinterpret_cast<const v4i*>(dig + (16 * nt + (lane & 15)) * DIG_STRIDE + 16 * (lane >> 4));
#pragma unroll
    for (int t = 0; t < 5; t++) {
      v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(c.a[t], bfrag, v4i{0, 0, 0, 0}, 0, 0, 0);
      // D: column (element) = lane & 15, rows 4 (lane >> 4) + r
      *reinterpret_cast<v4i*>(col + (16 * nt + (lane & 15)) * COL_STRIDE + 16 * t + 4 * (lane >> 4)) = d;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // (3) this lane's 75 column sums (weights 2^(7 i)) -> 18 limbs of 29 bits
  {
    const int* my = col + lane * COL_STRIDE;
    uint64_t acc = 0;
    int limb = 0;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int sh = 7 * i - B * limb;                        // compile-time after unrolling
      acc += (uint64_t)(uint32_t)my[i] << sh;
      if (7 * (i + 1) - B * limb >= B) {
        u[limb++] = (uint32_t)acc & MASK;
        acc >>= B;
      }
    }
#pragma unroll
    for (; limb < 2 * L; limb++) {
      u[limb] = (uint32_t)acc & MASK;
      acc >>= B;
    }
  }
}

template <int MODE>   // 0 = VALU, 1 = MFMA, 2 = both + compare
__global__ __launch_bounds__(256) void redc_kernel(uint32_t* out, int iters, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = threadIdx.x >> 6;
  uint8_t* dig = smem + wave * (64 * DIG_STRIDE + 64 * COL_STRIDE * 4);
  int* col = reinterpret_cast<int*>(dig + 64 * DIG_STRIDE);
  if (MODE != 0) {
    for (int i = threadIdx.x & 63; i < 64 * DIG_STRIDE / 4; i += 64) reinterpret_cast<uint32_t*>(dig)[i] = 0;
  }
  MfmaConst c = make_const();
  uint32_t m[L];
  uint32_t x = seed ^ (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
#pragma unroll
  for (int i = 0; i < L; i++) {
    x = x * 1664525u + 1013904223u;
    m[i] = (x >> 3) & MASK;
  }
  uint32_t bad = 0, fold = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t u[2 * L], v[2 * L];
    if (MODE == 0 || MODE == 2) mulp_valu(m, u);
    if (MODE == 1 || MODE == 2) mulp_mfma(c, m, v, dig, col);
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 2 * L; i++) bad |= u[i] ^ v[i];
    }
    const uint32_t* r = MODE == 1 ? v : u;
#pragma unroll
    for (int i = 0; i < L; i++) {                               // next operand: the high half xor the low half (keeps limbs < 2^29)
      m[i] = (r[i] ^ r[i + L]) & MASK;
      fold ^= r[i];
    }
  }
  out[(blockIdx.x * 256 + threadIdx.x) * 2] = bad;
  out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = fold ^ m[0];
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("# tools/ubench/mfma_redc.hip on %s (%d CUs): U = m * p (261-bit m, BN254 Fq modulus p) for every lane,\n", prop.gcnArchName, cus);
  printf("# V = 81 v_mad_u64_u32 on 29-bit limbs (what unsat_dev.hpp does) vs M = V_MFMA_I32_16X16X64_I8 on 7-bit digits\n");
  const size_t lds = 4 * (64 * DIG_STRIDE + 64 * COL_STRIDE * 4);       // 96 KiB per 256-thread workgroup
  uint32_t* out;
  const int blocks = cus * 1;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  // correctness: both variants on the same chained operands
  hipLaunchKernelGGL(redc_kernel<2>, dim3(blocks), dim3(256), lds, 0, out, 64, 12345u);
  std::vector<uint32_t> h((size_t)blocks * 256 * 2);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < h.size(); i += 2) bad += h[i] != 0;
  printf("mismatching lanes (V vs M, 64 chained products each, %d lanes): %zu\n", blocks * 256, bad);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 2; mode++) {
    for (int bpc = 1; bpc <= (mode == 0 ? 8 : 1); bpc *= 2) {
      const int nb = cus * bpc;
      uint32_t* o2;
      hipMalloc(&o2, (size_t)nb * 256 * 8);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(redc_kernel<0>, dim3(nb), dim3(256), 0, 0, o2, iters, 7u);
        else hipLaunchKernelGGL(redc_kernel<1>, dim3(nb), dim3(256), lds, 0, o2, iters, 7u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s workgroups/CU %d: %.2f G (m * p)/s\n", mode == 0 ? "V (v_mad_u64_u32)" : "M (int8 MFMA)    ", bpc,
             (double)nb * 256 * iters / (ms * 1e-3) / 1e9);
      hipFree(o2);
    }
  }
  return bad != 0;
}
