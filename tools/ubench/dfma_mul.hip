// Experiment (VERDICT r3 task 2): a Montgomery multiplier on the FP64 FMA pipe instead of v_mad_u64_u32.
//
// v_fma_f64 issues at the full vector rate on gfx950 (78.6 TFLOP/s FP64 vector = 64 lanes x 4 SIMDs x 256 CUs x 2.4 GHz x 2), the
// same rate the integer multiplier v_mad_u64_u32 sustains (profiles/r03: 0.76-0.85 of the vector rate), and one FMA pair splits a
// 52 x 52-bit product into two 52-bit halves: 5 x 5 = 25 partial products per 254-bit operand pair instead of 9 x 9 = 81 with
// 29-bit limbs.  What it costs per partial product decides the experiment (Emmart, Zheng, Weems, ARITH 2018):
//     p_hi = fma_rz(a, b, C1)       C1 = 2^104           -> 2^104 + hi * 2^52     (hi = floor(a*b / 2^52): round toward zero)
//     t    = C2 - p_hi              C2 = 2^104 + 2^52    -> 2^52 - hi * 2^52
//     p_lo = fma_rz(a, b, t)                             -> 2^52 + lo              (lo = a*b mod 2^52, exact, in one binade)
//     col[i+j+1] += bits(p_hi);  col[i+j] += bits(p_lo)     two 64-bit integer additions; the offsets are removed per column
// i.e. 3 FP64 + 2 integer-add instructions per partial product (the kernels run with MODE.fp_round[3:2] = toward zero, set once by
// s_setreg: no per-instruction cost) against ONE v_mad_u64_u32 per 29 x 29-bit partial product.  Limbs: 5 x 52 bits, Montgomery radix R = 2^260, word-serial
// reduction (q_i = col_i * (-p^-1) mod 2^52 with a 52-bit integer low product, then 5 more split products per word).
// Checked bit for bit against big-integer arithmetic (tools/ubench/dfma_check.py reads the dumped operands / results) and timed like
// zkp_bench_mulmod: CH independent product chains per lane, 8 workgroups per CU.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I ckb_zkp_amd/csrc -o tools/ubench/dfma_mul tools/ubench/dfma_mul.hip
//   tools/ubench/dfma_mul /tmp/dfma_dump.bin && python tools/ubench/dfma_check.py /tmp/dfma_dump.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>
#define ZKP_INLINE_MUL
#include "field_dev.hpp"
#include "unsat_dev.hpp"
using namespace zkp;

constexpr int L = 5, W = 52;
constexpr uint64_t MASK52 = (1ull << W) - 1;

// 52-bit limb i of the BN254 base-field modulus (from the library's 32-bit words)
__host__ __device__ constexpr uint64_t p_limb(int i) {
  uint64_t v = 0;
  for (int b = 0; b < W; b++) {
    const int bit = i * W + b;
    if (bit < 256 && ((Bn254Fq::MOD[bit >> 5] >> (bit & 31)) & 1u)) v |= 1ull << b;
  }
  return v;
}
// -p^-1 mod 2^52 (Newton iteration on the low limb)
__host__ __device__ constexpr uint64_t p_inv52() {
  const uint64_t p0 = p_limb(0);
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x = x * (2 - p0 * x);
  return (0 - x) & MASK52;
}

struct D5 {
  double v[L];
};

__device__ __forceinline__ double u52_to_double(uint64_t x) {          // exact: x < 2^52
  return __longlong_as_double((long long)(x | 0x4330000000000000ull)) - 4503599627370496.0;
}

// out = a * b * 2^-260 mod p (value < 2p for inputs < 2p); limbs are doubles holding integers in [0, 2^52)
__device__ __forceinline__ D5 dfma_mont_mul(const D5& a, const D5& b) {
  const double C1 = 20282409603651670423947251286016.0;                 // 2^104
  const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;   // 2^104 + 2^52
  const long long B1 = 0x4670000000000000ll;                            // bits(2^104): p_hi = B1 + hi (ulp = 2^52)
  const long long B2 = 0x4330000000000000ll;                            // bits(2^52): p_lo = B2 + lo (ulp = 1)
  // compile-time constants (constexpr variables force the evaluation: a plain call of the constexpr functions is emitted as run-time bit loops)
  constexpr double PD0 = (double)p_limb(0), PD1 = (double)p_limb(1), PD2 = (double)p_limb(2), PD3 = (double)p_limb(3), PD4 = (double)p_limb(4);
  constexpr uint64_t PINV = p_inv52();
  long long col[2 * L + 1];
#pragma unroll
  for (int k = 0; k <= 2 * L; k++) col[k] = 0;
  // ---- a * b
#pragma unroll
  for (int i = 0; i < L; i++) {
#pragma unroll
    for (int j = 0; j < L; j++) {
      const double ph = __builtin_fma(a.v[i], b.v[j], C1);
      const double pl = __builtin_fma(a.v[i], b.v[j], C2 - ph);
      col[i + j + 1] += __double_as_longlong(ph) - B1;
      col[i + j] += __double_as_longlong(pl) - B2;
    }
  }
  // ---- word-serial Montgomery reduction
#pragma unroll
  for (int i = 0; i < L; i++) {
    const uint64_t t = (uint64_t)col[i] & MASK52;
    const uint64_t q = (t * PINV) & MASK52;                        // 52-bit low product on the integer pipe
    const double qd = u52_to_double(q);
#pragma unroll
    for (int j = 0; j < L; j++) {
      const double pj = j == 0 ? PD0 : j == 1 ? PD1 : j == 2 ? PD2 : j == 3 ? PD3 : PD4;
      const double ph = __builtin_fma(qd, pj, C1);
      const double pl = __builtin_fma(qd, pj, C2 - ph);
      col[i + j + 1] += __double_as_longlong(ph) - B1;
      col[i + j] += __double_as_longlong(pl) - B2;
    }
    col[i + 1] += col[i] >> W;                                          // the low 52 bits are zero by construction
  }
  // ---- carry the upper columns into 52-bit limbs
  D5 r;
  long long carry = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    const long long s = col[L + k] + carry;
    r.v[k] = u52_to_double((uint64_t)s & MASK52);
    carry = s >> W;
  }
  return r;
}

// MODE.fp_round bits [3:2] (FP64 / FP16) := 3 = round toward zero; hwreg(HW_REG_MODE = 1, offset 2, width 2)
// (inline asm, issued AFTER the prologue's integer -> double conversions: with the builtin, LLVM's mode-register pass resets the field
// to round-to-nearest behind the first v_cvt_f64_u32 it meets)
__device__ __forceinline__ void fp64_round_toward_zero() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory"); }

template <int CH>
__global__ __launch_bounds__(256) void dfma_rate_kernel(double* __restrict__ out, int iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  D5 a[CH], c[CH];
#pragma unroll
  for (int k = 0; k < CH; k++) {
#pragma unroll
    for (int i = 0; i < L; i++) a[k].v[i] = u52_to_double(((uint64_t)t * 2654435761ull + 40503ull * k + i * 977ull) & (i == L - 1 ? (1ull << 45) - 1 : MASK52));
    c[k] = a[k];
  }
  fp64_round_toward_zero();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) c[k] = dfma_mont_mul(c[k], a[k]);
  }
#pragma unroll
  for (int i = 0; i < L; i++) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < CH; k++) s += c[k].v[i];
    out[(size_t)t * L + i] = s;
  }
}

// the library's unsaturated multiplier in the same harness (what zkp_bench_mulmod measures)
template <int CH>
__global__ __launch_bounds__(256) void unsat_rate_kernel(uint32_t* __restrict__ out, int iters) {
  using F = Fp<Bn254Fq>;
  using U = Fu<Bn254Fq>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  U ua[CH], c[CH];
#pragma unroll
  for (int k = 0; k < CH; k++) {
    F a = F::one();
    a.v[k & 1] ^= t * (2654435761u + 40503u * k) + k;
    a = F::reduce_once(a);
    c[k] = ua[k] = U::from_sat_reduced(a);
  }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < CH; k++) c[k] = U::mul(c[k], ua[k]);
  }
  U s = c[0];
#pragma unroll
  for (int k = 1; k < CH; k++) s = U::add(s, c[k]);
#pragma unroll
  for (int i = 0; i < Bn254Fq::N; i++) out[(size_t)t * Bn254Fq::N + i] = s.v[i];
}

// one product per lane on given operands: a, b, out as 5 u64 limbs each (the check dump)
__global__ void dfma_check_kernel(const uint64_t* __restrict__ ab, uint64_t* __restrict__ out, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  D5 a, b;
#pragma unroll
  for (int i = 0; i < L; i++) {
    a.v[i] = u52_to_double(ab[t * 10 + i]);
    b.v[i] = u52_to_double(ab[t * 10 + 5 + i]);
  }
  fp64_round_toward_zero();
  const D5 r = dfma_mont_mul(a, b);
#pragma unroll
  for (int i = 0; i < L; i++) out[t * L + i] = (uint64_t)__double_as_longlong(r.v[i] + 4503599627370496.0) & MASK52;   // exact: limb < 2^52
}

#define HIPCHECK(x)                                                               \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));                   \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

static uint64_t sm64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# tools/ubench/dfma_mul.hip on %s (%d CUs): 254-bit Montgomery product, 5 x 52-bit limbs on v_fma_f64 vs 9 x 29-bit limbs on v_mad_u64_u32\n",
         prop.gcnArchName, cus);
  // ---- bit-exactness dump: n random operand pairs < 2p plus the edge set, one product each
  const size_t n = 1u << 20;
  std::vector<uint64_t> ab(n * 10);
  uint64_t p[L], p2[L];
  for (int i = 0; i < L; i++) p[i] = p_limb(i);
  {                                                                      // 2p - 1
    unsigned __int128 c = 0;
    for (int i = 0; i < L; i++) {
      c += (unsigned __int128)p[i] * 2;
      p2[i] = (uint64_t)c & MASK52;
      c >>= W;
    }
    p2[0] -= 1;                                                          // 2p is even and nonzero in limb 0 (p odd)
  }
  uint64_t seed = 0xD0FA;
  auto lt2p = [&](const uint64_t* x) {                                   // x <= 2p - 1
    for (int i = L - 1; i >= 0; i--)
      if (x[i] != p2[i]) return x[i] < p2[i];
    return true;
  };
  for (size_t t = 0; t < n; t++) {
    for (int h = 0; h < 2; h++) {
      uint64_t* x = &ab[t * 10 + 5 * h];
      do {
        for (int i = 0; i < L; i++) x[i] = sm64(seed) & MASK52;
        x[L - 1] &= (1ull << (255 - 4 * W)) - 1;                         // < 2^255
      } while (!lt2p(x));
    }
  }
  // edge set in the first lanes: {0, 1, p - 1, p, 2p - 1} x {0, 1, p - 1, p, 2p - 1}
  uint64_t edge[5][L] = {};
  edge[1][0] = 1;
  for (int i = 0; i < L; i++) edge[2][i] = p[i], edge[3][i] = p[i], edge[4][i] = p2[i];
  edge[2][0] -= 1;
  for (int x = 0; x < 5; x++)
    for (int y = 0; y < 5; y++)
      for (int i = 0; i < L; i++) ab[(x * 5 + y) * 10 + i] = edge[x][i], ab[(x * 5 + y) * 10 + 5 + i] = edge[y][i];
  uint64_t *d_ab, *d_out;
  HIPCHECK(hipMalloc(&d_ab, n * 80));
  HIPCHECK(hipMalloc(&d_out, n * 40));
  HIPCHECK(hipMemcpy(d_ab, ab.data(), n * 80, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(dfma_check_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_ab, d_out, n);
  HIPCHECK(hipDeviceSynchronize());
  std::vector<uint64_t> res(n * 5);
  HIPCHECK(hipMemcpy(res.data(), d_out, n * 40, hipMemcpyDeviceToHost));
  if (argc > 1) {
    FILE* f = fopen(argv[1], "wb");
    if (!f) return 1;
    uint64_t hdr[2] = {n, L};
    fwrite(hdr, 8, 2, f);
    fwrite(p, 8, L, f);
    fwrite(ab.data(), 8, n * 10, f);
    fwrite(res.data(), 8, n * 5, f);
    fclose(f);
    printf("dumped %zu operand pairs + products to %s (check: python tools/ubench/dfma_check.py %s)\n", n, argv[1], argv[1]);
  }
  // ---- rates
  const int threads = 256, iters = 800;
  void* out;
  HIPCHECK(hipMalloc(&out, (size_t)cus * 8 * threads * 12 * 8));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  auto time = [&](auto launch) {
    launch();
    HIPCHECK(hipEventRecord(e0, 0));
    launch();
    HIPCHECK(hipEventRecord(e1, 0));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms;
  };
  for (int wg : {4, 8}) {
    const int blocks = cus * wg;
    const double lanes = (double)blocks * threads * iters;
    double m;
    m = time([&] { hipLaunchKernelGGL((dfma_rate_kernel<1>), dim3(blocks), dim3(threads), 0, 0, (double*)out, iters); });
    printf("FP64 FMA   multiplier, 1 chain /lane, %d workgroups/CU: %7.2f G products/s\n", wg, lanes * 1 / (m * 1e-3) / 1e9);
    m = time([&] { hipLaunchKernelGGL((dfma_rate_kernel<2>), dim3(blocks), dim3(threads), 0, 0, (double*)out, iters); });
    printf("FP64 FMA   multiplier, 2 chains/lane, %d workgroups/CU: %7.2f G products/s\n", wg, lanes * 2 / (m * 1e-3) / 1e9);
    m = time([&] { hipLaunchKernelGGL((dfma_rate_kernel<4>), dim3(blocks), dim3(threads), 0, 0, (double*)out, iters); });
    printf("FP64 FMA   multiplier, 4 chains/lane, %d workgroups/CU: %7.2f G products/s\n", wg, lanes * 4 / (m * 1e-3) / 1e9);
    m = time([&] { hipLaunchKernelGGL((unsat_rate_kernel<2>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)out, iters); });
    printf("v_mad_u64  multiplier, 2 chains/lane, %d workgroups/CU: %7.2f G products/s   (unsat_dev.hpp, = zkp_bench_mulmod)\n", wg,
           lanes * 2 / (m * 1e-3) / 1e9);
    m = time([&] { hipLaunchKernelGGL((unsat_rate_kernel<4>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)out, iters); });
    printf("v_mad_u64  multiplier, 4 chains/lane, %d workgroups/CU: %7.2f G products/s\n", wg, lanes * 4 / (m * 1e-3) / 1e9);
  }
  return 0;
}
