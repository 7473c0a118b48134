/* zkp_accel.h — C ABI of the MI355X-native MSM + NTT proving backend for sec-bit/ckb-zkp.
 *
 * The reference (Rust) has no FFI for this path; this header *defines* the drop-in
 * boundary.  Each entry point names the reference call it replaces; INTEGRATION.md shows the Rust
 * `extern "C"` shim a maintainer would add.
 *
 * Conventions
 *   - every function returns int32_t: ZKP_OK (0) or a negative zkp_status; nothing throws across the ABI;
 *   - field elements: little-endian u64 limbs, MONTGOMERY form (R = 2^256, or 2^384 for BLS12-381 Fq)
 *     == ark-ff `Fp256/Fp384` in memory;  Fr = 4 limbs; Fq = 4 (BN254) / 6 (BLS12-381) limbs;
 *   - MSM scalars: 4 x u64 CANONICAL integers < r  == ark `BigInteger256` from `into_repr()`;
 *   - affine points: AoS (x, y), G2 coordinates (c0, c1); identity = separate u8 flag array (NULL = none);
 *   - projective results: ark Jacobian (X, Y, Z), identity = (0, 1, 0) — compare after `into_affine()`;
 *   - `*_dev` variants take DEVICE pointers (hipMalloc / torch tensors) and run on the ctx stream;
 *     host variants copy in/out and synchronise;
 *   - one in-flight call per ctx; create one ctx per prover thread / per GPU (one process per GPU): contexts on one device run
 *     concurrently from different host threads (the reference may prove from several rayon threads).  Since 0.5 every entry point
 *     that takes a ctx holds its lock for the whole call, so two threads entering the SAME ctx are serialised, never interleaved
 *     (0.6: zkp_set_profiling / zkp_*_last_timing too; the zkp_*_multi entry points hold the root's and every member's lock;
 *     zkp_ctx_destroy waits for a call in flight — destroying a context another thread still uses stays the caller's bug).
 */
#ifndef ZKP_ACCEL_H
#define ZKP_ACCEL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { ZKP_BN254 = 0, ZKP_BLS12_381 = 1 } zkp_curve_t;

typedef enum {
  ZKP_OK = 0,
  ZKP_ERR_BAD_ARG = -1,
  ZKP_ERR_UNSUPPORTED_CURVE = -2,
  ZKP_ERR_DOMAIN_TOO_LARGE = -3, /* == SynthesisError::PolynomialDegreeTooLarge (groth16/src/r1cs_to_qap.rs:123-125) */
  ZKP_ERR_OOM = -4,
  ZKP_ERR_DEVICE = -5,           /* HIP runtime error, or no MI355X/gfx950 device: there is NO CPU fallback */
  ZKP_ERR_BAD_HANDLE = -6,
  ZKP_ERR_INVALID_POINT = -7 /* zkp_g*_decompress / zkp_g*_subgroup_check: a point is malformed / off the curve / outside the subgroup
                               (== ark-serialize SerializationError::InvalidData); *bad_index holds its index */
} zkp_status;

/* ops of zkp_ntt: ark-poly `EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place`
 * (groth16/src/r1cs_to_qap.rs:144-148,161-162,169) */
typedef enum { ZKP_NTT_FFT = 0, ZKP_NTT_IFFT = 1, ZKP_NTT_COSET_FFT = 2, ZKP_NTT_COSET_IFFT = 3 } zkp_ntt_op;

typedef struct zkp_ctx zkp_ctx; /* opaque: device, stream, twiddle tables, scratch, resident bases */

const char* zkp_status_string(int32_t status);
/* "zkp_accel <major.minor> (gfx950)".  0.6 (round 6): zkp_ctx_config / zkp_ctx_create_ex / zkp_ctx_create_multi_ex / zkp_ctx_get_config (the
 * prover switches are per context; the environment only supplies defaults, read when the context is created); RCCL bring-up behind a
 * watchdog (zkp_groth16_multi_info info[0] == 2); the multi-GPU entry points lock every member context.  0.5 (round 5): per-context lock (see Conventions); zkp_groth16_pk_upload_ex (ZKP_PK_KEEP_FORM); ZKP_MULTI_EXCHANGE=rccl also takes the RCCL
 * exchange with one rank; slots L / H of zkp_groth16_prove_partials_dev are only defined as a SUM for folded / evaluation-form /
 * bucket-chained keys (zkp_groth16_pk_info info[7] says which).  0.4 (round 4): ZKP_ERR_INVALID_POINT for malformed / out-of-subgroup points (0.2 used
 * ZKP_ERR_BAD_ARG), zkp_groth16_multi_info, zkp_bench_hbm_copy, zkp_groth16_points_into_affine; since 0.3 a bucket-chained key returns slot L of
 * zkp_groth16_prove_partials_dev as the identity and slot H as h + l (their sum is what prover.rs:189-196 consumes). */
const char* zkp_version(void);

/* ---- context & device memory ------------------------------------------------------------------ */
int32_t zkp_ctx_create(zkp_ctx** out, int device_id);
/* Per-context configuration (since 0.6; SURVEY §5 "Config / flags": struct, with the environment as the default).  A field left
 * at 0 means "default": the environment variable named beside it when the process has it set, else the built-in choice — so
 * zkp_ctx_create(out, dev) == zkp_ctx_create_ex(out, dev, NULL) and the A/B variables keep working.  A field that is set wins
 * over the environment and belongs to THIS context only: two contexts of one process may differ (the library keeps no other
 * prover state outside zkp_ctx).  Switches are tri-state: 0 default, ZKP_ON, ZKP_OFF. */
typedef enum { ZKP_DEFAULT = 0, ZKP_ON = 1, ZKP_OFF = 2 } zkp_tristate;
typedef enum { ZKP_EXCHANGE_AUTO = 0, ZKP_EXCHANGE_RCCL = 1, ZKP_EXCHANGE_PEER = 2 } zkp_exchange;
typedef struct {
  uint32_t struct_size;        /* sizeof(zkp_ctx_config) as the caller compiled it (fields beyond it are defaults); 0 is rejected */
  int32_t lanes;               /* proofs in flight inside zkp_groth16_prove_batch*: 1..8            [ZKP_LANES; 8, 4 above 2^22] */
  int32_t msm_batch_lanes;     /* lanes zkp_msm_g1_mont_batch_dev / Marlin's commitments rotate over [ZKP_BATCH_LANES; 1] */
  int32_t msm_window_bits;     /* bucket-index bits c of resident-table MSMs, 2..22                 [ZKP_MSM_C; round(log2 n) <= 20] */
  int32_t msm_window_bits_g2;  /* the same for G2 bases                                             [ZKP_MSM_C_G2; as G1] */
  int64_t msm_chunk_points;    /* lone G1 MSMs of >= 2x this many points run in chunks that share one bucket array; -1 = never
                                  chunk                                                             [ZKP_MSM_CHUNK; 3 * 2^19] */
  double table_budget_gb;      /* budget for resident window tables of this context, GiB            [ZKP_TABLE_BUDGET_GB; free
                                  device memory minus a quarter of the device] */
  int32_t h_evaluation_form;   /* tri-state: H query transformed to evaluation form at key upload   [ZKP_H_LAGRANGE; on] */
  int32_t c_fold;              /* tri-state: C matrix folded into the L query at key upload         [ZKP_C_FOLD; on] */
  int32_t host_affine;         /* tri-state: into_affine of the three proof points on the host      [ZKP_HOST_AFFINE; on] */
  int64_t c_fold_heavy_cost;   /* a C column above this cost (1 per +-1 coefficient, 380 per general one) leaves the fold kernel
                                  for one MSM of its own                                            [ZKP_LFOLD_HEAVY_COST; chosen per
                                  key: the cut that minimises longest kernel chain + 150 per heavy column, never below 1000] */
  int32_t multi_exchange;      /* ZKP_EXCHANGE_*: partial sums of zkp_groth16_prove_multi           [ZKP_MULTI_EXCHANGE; auto =
                                  RCCL all-gather when the devices are distinct and librccl loads, else peer copies] */
  int32_t multi_exchange_timeout_ms; /* watchdog of the RCCL setup and of the first all-gather: when it expires the key falls
                                  back to peer copies (said on stderr and in zkp_groth16_multi_info)  [ZKP_MULTI_EXCHANGE_TIMEOUT_MS; 30000] */
  int32_t multi_witness_split; /* tri-state: a / b / c chains of the witness map on devices 0 / 1 / 2 [ZKP_MULTI_WM_SPLIT;
                                  measured per key on proofs 3-4] */
} zkp_ctx_config;
/* cfg == NULL: all defaults.  ZKP_ERR_BAD_ARG for struct_size == 0 or a field out of range. */
int32_t zkp_ctx_create_ex(zkp_ctx** out, int device_id, const zkp_ctx_config* cfg);
/* the RESOLVED configuration of a context (defaults replaced by what the context actually uses; 0 where the choice depends on the
 * job, e.g. lanes / msm_window_bits) */
int32_t zkp_ctx_get_config(zkp_ctx* ctx, zkp_ctx_config* out);
int32_t zkp_ctx_destroy(zkp_ctx* ctx);
/* run on an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int32_t zkp_ctx_set_stream(zkp_ctx* ctx, void* hip_stream);
int32_t zkp_ctx_sync(zkp_ctx* ctx);
int32_t zkp_dev_alloc(zkp_ctx* ctx, size_t bytes, void** dptr);
int32_t zkp_dev_free(zkp_ctx* ctx, void* dptr);
int32_t zkp_h2d(zkp_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int32_t zkp_d2h(zkp_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* stream-ordered timing helpers (HIP events on the ctx stream) used by bench.py */
int32_t zkp_timer_start(zkp_ctx* ctx);
int32_t zkp_timer_stop_ms(zkp_ctx* ctx, float* ms);

/* ---- NTT: replaces ark-poly GeneralEvaluationDomain::<Fr> ops ----------------------------------
 * data: 2^log_n Fr elements (Montgomery), natural order in and out, in place.
 * log_n > TWO_ADICITY (28 BN254 / 32 BLS12-381) -> ZKP_ERR_DOMAIN_TOO_LARGE.
 * coset generator = Fr::multiplicative_generator() = 5 (BN254) / 7 (BLS12-381). */
int32_t zkp_ntt(zkp_ctx* ctx, zkp_curve_t curve, uint64_t* data_host, uint32_t log_n, int32_t op);
int32_t zkp_ntt_dev(zkp_ctx* ctx, zkp_curve_t curve, uint64_t* data_dev, uint32_t log_n, int32_t op);

/* ---- Fr vector / polynomial primitives around the Marlin prover's NTTs and KZG10 MSMs (device pointers) -----
 * vectors: Fr Montgomery, 4 x u64 per element.  k / z: one Fr element in HOST memory. */
typedef enum { ZKP_VEC_MUL = 0, ZKP_VEC_ADD = 1, ZKP_VEC_SUB = 2, ZKP_VEC_SCALE = 3, ZKP_VEC_AXPY = 4,
               ZKP_VEC_ADDC = 5 } zkp_vec_op;
/* out[i] = a[i]*b[i] | a[i]+b[i] | a[i]-b[i] | k*a[i] | a[i]+k*b[i] | a[i]+k   (marlin/src/ahp/prover.rs:248-252,298-305,399-411);
 * out may alias a or b */
int32_t zkp_fr_vec_op_dev(zkp_ctx* ctx, zkp_curve_t curve, int32_t op, const uint64_t* a, const uint64_t* b,
                          const uint64_t* k_host, uint64_t* out, size_t n);
/* sparse matrix-vector product, CSR on the device: out[i] = sum_k coeff[k] * x[col[k]]  (z_a = A z of
 * marlin/src/ahp/prover.rs:110-123; the `t` accumulation of :259-269 as a transposed product) */
int32_t zkp_fr_spmv_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint32_t* row_ptr_dev, const uint32_t* col_dev,
                        const uint64_t* coeff_dev, size_t nrows, const uint64_t* x_dev, uint64_t* out_dev);
/* out[i] = idx[i] < 0 ? 0 : in[idx[i]]   (interleaving of inputs/witness over H, prover.rs:176-186) */
int32_t zkp_fr_gather_dev(zkp_ctx* ctx, const uint64_t* in_dev, const int32_t* idx_dev, size_t n, uint64_t* out_dev);
/* DensePolynomial::divide_by_vanishing_poly (prover.rs:191,204,307,415): p = q (X^n - 1) + rem;
 * q_dev: len - n coefficients (or NULL), rem_dev: n coefficients (or NULL) */
int32_t zkp_poly_divide_by_vanishing_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* p_dev, size_t len, size_t n,
                                         uint64_t* q_dev, uint64_t* rem_dev);
/* stream-ordered device-to-device copy / zero fill (polynomial shifts, paddings) */
int32_t zkp_d2d(zkp_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int32_t zkp_dev_zero(zkp_ctx* ctx, void* dst_dev, size_t bytes);
/* ark_ff::fields::batch_inversion, in place; zeros stay zero (marlin/src/ahp/prover.rs:357-367) */
int32_t zkp_fr_batch_inverse_dev(zkp_ctx* ctx, zkp_curve_t curve, uint64_t* v, size_t n);
/* DensePolynomial::evaluate (marlin/src/lib.rs:147-156): eval_out_host = sum_i p[i] z^i */
int32_t zkp_poly_evaluate_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* p, size_t n, const uint64_t* z_host,
                              uint64_t* eval_out_host);
/* KZG10 witness polynomial p / (X - z), remainder discarded (marlin/src/pc/kzg10.rs:211-226):
 * q_dev receives n-1 coefficients (q_dev != p); eval_out_host (optional) receives p(z) */
int32_t zkp_poly_div_linear_dev(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* p, size_t n, const uint64_t* z_host,
                                uint64_t* q_dev, uint64_t* eval_out_host);

/* ---- bases (proving-key queries / SRS powers): upload once, prove many -------------------------
 * xy: n affine points (Montgomery); inf: n flags or NULL.  The library copies (and pre-computes its
 * window tables in HBM); the caller keeps ownership of the host buffers. */
int32_t zkp_bases_upload_g1(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n,
                            uint64_t* handle);
int32_t zkp_bases_upload_g2(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n,
                            uint64_t* handle);
int32_t zkp_bases_free(zkp_ctx* ctx, uint64_t handle);
/* Make a resident base vector of `src` usable through `dst` as well (same device, same process): the window tables are
 * shared, not copied — one SRS / proving key serves several contexts (one per prover thread).  The entry lives until
 * every context that holds it has freed it. */
int32_t zkp_bases_share(zkp_ctx* dst, zkp_ctx* src, uint64_t src_handle, uint64_t* dst_handle);
int32_t zkp_bases_len(zkp_ctx* ctx, uint64_t handle, size_t* n);

/* ---- MSM: replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul ------------------------------
 * (groth16/src/prover.rs:187,190,220; marlin/src/pc/kzg10.rs:109,118,137,146; curve/src/lib.rs:44)
 * result = sum_{i<n} scalars[i] * bases[offset+i]; n is clamped to the bases available (ark's
 * min(len) truncation, prover.rs:186-187); n == 0 -> identity.
 * out_xyz: 3 (G1) or 6 (G2) field elements, Jacobian, Montgomery, host memory. */
int32_t zkp_msm_g1(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* scalars_host, size_t n,
                   uint64_t* out_xyz);
int32_t zkp_msm_g2(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* scalars_host, size_t n,
                   uint64_t* out_xyz);
int32_t zkp_msm_g1_dev(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* scalars_dev, size_t n,
                       uint64_t* out_xyz_host);
int32_t zkp_msm_g2_dev(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* scalars_dev, size_t n,
                       uint64_t* out_xyz_host);
/* zkp_curve::Curve::vartime_multiscalar_mul (curve/src/lib.rs:38-45): scalars are Fr in MONTGOMERY form;
 * the into_repr() map is fused into the digit scan on the device. */
int32_t zkp_vartime_multiscalar_mul_g1(zkp_ctx* ctx, uint64_t handle, const uint64_t* fr_scalars_host, size_t n,
                                       uint64_t* out_xyz);
int32_t zkp_vartime_multiscalar_mul_g2(zkp_ctx* ctx, uint64_t handle, const uint64_t* fr_scalars_host, size_t n,
                                       uint64_t* out_xyz);
/* TRUE variable-base MSM — the literal semantics of `VariableBaseMSM::multi_scalar_mul(bases, scalars)` and of
 * `Curve::vartime_multiscalar_mul(scalars, points)` (curve/src/lib.rs:38-45) when the bases are FRESH on every call
 * (bulletproofs / spartan / hyrax generators): points and scalars come from HOST memory, nothing is precomputed and
 * nothing stays resident (no window tables: W = 256 / c separate bucket sets, c = 16 from 2^12 points on, and a
 * <= 255-doubling tail).  montgomery != 0: scalars are Fr elements (into_repr() fused), else canonical BigInteger256.
 * Use the resident-handle entry points above when the same bases serve many MSMs (proving keys, SRS): they are ~3x faster. */
int32_t zkp_msm_g1_var(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy_host, const uint8_t* inf_host,
                       const uint64_t* scalars_host, size_t n, int32_t montgomery, uint64_t* out_xyz);
int32_t zkp_msm_g2_var(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy_host, const uint8_t* inf_host,
                       const uint64_t* scalars_host, size_t n, int32_t montgomery, uint64_t* out_xyz);
/* KZG10::commit / open (marlin/src/pc/kzg10.rs:108-109,137-140): MSM of Montgomery Fr coefficients that are already
 * on the DEVICE against powers[offset ..] (offset = number of skipped leading zeros) */
int32_t zkp_msm_g1_mont_dev(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* fr_scalars_dev, size_t n,
                            uint64_t* out_xyz_host);
int32_t zkp_msm_g2_mont_dev(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* fr_scalars_dev, size_t n,
                            uint64_t* out_xyz_host);
/* PC::commit over a LIST of polynomials (marlin/src/pc/mod.rs:34-71; one KZG10::commit = one MSM each, kzg10.rs:100-123):
 * `count` MSMs of device-resident Montgomery coefficient vectors against powers[offsets[k] ..], min(ns[k], len - offset)
 * terms each, three in flight at a time on the context's MSM streams.  out_xyz: count Jacobian results. */
int32_t zkp_msm_g1_mont_batch_dev(zkp_ctx* ctx, uint64_t bases_handle, size_t count, const size_t* offsets,
                                  const uint64_t* const* scalars_dev, const size_t* ns, uint64_t* out_xyz);
/* The same for MSMs against DIFFERENT resident base vectors (G1 and G2 mixed): the five partial MSMs of a base-sharded
 * Groth16 proof (prover.rs:164-190) in one call, four in flight.  Result k is written at out_xyz + k * slot_u64 (its
 * Jacobian limbs first); slot_u64 >= 3 * limbs of the largest group involved (18 for BN254 G2, 36 for BLS12-381 G2). */
int32_t zkp_msm_mont_multi_dev(zkp_ctx* ctx, size_t count, const uint64_t* bases_handles, const size_t* offsets,
                               const uint64_t* const* scalars_dev, const size_t* ns, uint64_t* out_xyz, size_t slot_u64);
/* fold k Jacobian points (host) into one: the local step after the multi-GPU all-gather of partial MSM
 * results (EC addition is not an RCCL reduction op) */
int32_t zkp_g1_fold(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz_host, size_t k, uint64_t* out_xyz);
int32_t zkp_g2_fold(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz_host, size_t k, uint64_t* out_xyz);
/* Jacobian -> affine (x, y) + identity flag: ark `into_affine()` */
int32_t zkp_g1_into_affine(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz_host, uint64_t* xy_out,
                           uint8_t* inf_out);
int32_t zkp_g2_into_affine(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xyz_host, uint64_t* xy_out,
                           uint8_t* inf_out);
/* HOST ONLY (no device, ctx may be NULL-free: there is none): the into_affine() of the three proof points (prover.rs:205-209) as
 * zkp_groth16_prove* runs it since ABI 0.4 — a, c: G1 XYZZ (x = X / ZZ, y = Y / ZZZ; ZZ == 0: the identity; 4 coordinates of
 * L u64 limbs, Montgomery), b: G2 XYZZ (8 coordinates, (c0, c1) pairs) -> proof_out = A | B | C affine Montgomery words exactly as
 * zkp_groth16_prove writes them (ark's (0, 0) for the identity), inf_out[3].  One field inversion for all three points. */
int32_t zkp_groth16_points_into_affine(zkp_curve_t curve, const uint64_t* a_xyzz, const uint64_t* b_xyzz, const uint64_t* c_xyzz,
                                       uint64_t* proof_out, uint8_t* inf_out);

/* ark-serialize 0.2 COMPRESSED short-Weierstrass points <-> the affine Montgomery arrays of zkp_groth16_pk_desc /
 * zkp_bases_upload_*: what `Parameters::serialize` writes into a .pk file (cli/src/setup.rs:41-45) and `Proof::serialize` into
 * a proof (cli/src/zkp_prove.rs:45-49).  A point is its canonical little-endian x (G2: c0 then c1; 32 / 64 bytes on BN254,
 * 48 / 96 on BLS12-381) with two flags in the top bits of the last byte: bit 7 = y is the larger of {y, -y} (Fq: as integers;
 * Fq2: c1 first, then c0), bit 6 = the identity.  Decompression (a square root per point) runs on the device, one lane per point:
 * a 2^20 key loads in tens of milliseconds.  On a malformed point (both flags, x >= p, x^3 + b not a square) the call returns
 * ZKP_ERR_INVALID_POINT, *bad_index (if not NULL) receives its index and xy_out / inf_out are left untouched (*bad_index is
 * SIZE_MAX after any other outcome, so an argument error is never mistaken for "point 0").  No subgroup check (= ark's `deserialize_unchecked` for the
 * cofactor groups; BN254 G1 has cofactor 1).  The container framing (Vec length prefixes, field order of `Parameters`) stays with
 * the caller: ckb_zkp_amd/serialize.py, rust/zkp-accel. */
int32_t zkp_g1_decompress(zkp_ctx* ctx, zkp_curve_t curve, const uint8_t* bytes, size_t n, uint64_t* xy_out, uint8_t* inf_out,
                          size_t* bad_index);
int32_t zkp_g2_decompress(zkp_ctx* ctx, zkp_curve_t curve, const uint8_t* bytes, size_t n, uint64_t* xy_out, uint8_t* inf_out,
                          size_t* bad_index);
int32_t zkp_g1_compress(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* bytes_out);
int32_t zkp_g2_compress(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* bytes_out);
/* The CHECKED half of ark-serialize's `deserialize` for curve points (ark-ec 0.2 `GroupAffine::deserialize`: on the curve and
 * `is_in_correct_subgroup_assuming_on_curve`, i.e. [r]P = O; what `Parameters::deserialize` / `Proof::deserialize` /
 * `VerifyKey::deserialize` run per element — /root/reference/cli/src/zkp_prove.rs:45, zkp_verify.rs:61-62).  xy: n affine Montgomery points
 * (the layout zkp_g*_decompress writes), inf: optional identity flags (identities pass).  One lane per point, a double-and-add over
 * the group order: ~0.2 s for a 2^20-element G2 query.  ZKP_OK if every point passes; otherwise ZKP_ERR_INVALID_POINT and *bad_index
 * (if not NULL) = index of the first point that is off the curve or outside the prime-order subgroup. */
int32_t zkp_g1_subgroup_check(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, size_t* bad_index);
int32_t zkp_g2_subgroup_check(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* xy, const uint8_t* inf, size_t n, size_t* bad_index);

/* ---- fixed-base multiples k_i * P (setup side, generator.rs:205-256 `FixedBaseMSM`).
 * Used to build synthetic proving keys from a known trapdoor at 2^20+ scale.  scalars canonical. */
int32_t zkp_fixed_base_mul_g1(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* base_xy, const uint64_t* scalars_host,
                              size_t n, uint64_t* out_xy, uint8_t* out_inf);
int32_t zkp_fixed_base_mul_g2(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* base_xy, const uint64_t* scalars_host,
                              size_t n, uint64_t* out_xy, uint8_t* out_inf);

/* ---- Groth16: replaces zkp_groth16::create_proof (groth16/src/prover.rs:124-211) ---------------
 * The circuit closures stay on the caller's side (Rust); the boundary receives what
 * `ProvingAssignment` holds after synthesis (prover.rs:16-25): the three sparse matrices and the
 * assignment.  Matrices are fixed per circuit -> uploaded with the key.
 * A descriptor with a_query == NULL uploads the matrices only (witness_map works, prove is refused): used by
 * the base-sharded multi-GPU prover, where every rank holds the matrices and a slice of each query. */
typedef struct {
  /* CSR over constraints; coeffs Fr Montgomery; col = index into z = input_assignment ++ aux_assignment */
  const uint32_t* row_ptr; /* num_constraints + 1 */
  const uint32_t* col;     /* nnz */
  const uint64_t* coeff;   /* nnz x 4 */
} zkp_csr;

typedef struct {
  zkp_curve_t curve;
  uint32_t num_inputs;      /* incl. the constant one */
  uint32_t num_aux;
  uint32_t num_constraints;
  zkp_csr at, bt, ct;       /* prover.rs:17-19 */
  /* Parameters<E> (groth16/src/lib.rs:79-91), host affine Montgomery + identity flags (NULL = none) */
  const uint64_t* alpha_g1; const uint64_t* beta_g1; const uint64_t* delta_g1; /* 1 G1 point each */
  const uint64_t* beta_g2; const uint64_t* delta_g2;                           /* 1 G2 point each */
  const uint64_t* a_query; const uint8_t* a_inf; size_t a_len;
  const uint64_t* b_g1_query; const uint8_t* b_g1_inf; size_t b_g1_len;
  const uint64_t* b_g2_query; const uint8_t* b_g2_inf; size_t b_g2_len;
  const uint64_t* h_query; const uint8_t* h_inf; size_t h_len;
  const uint64_t* l_query; const uint8_t* l_inf; size_t l_len;
} zkp_groth16_pk_desc;

typedef struct zkp_groth16_pk zkp_groth16_pk; /* opaque device-resident proving key */

/* Uploads `Parameters<E>` + the circuit matrices once; every zkp_groth16_prove* call then runs on the resident key.  Besides the
 * window tables the upload derives (round 4, one-time, on the device: + 0.65 s at 2^20, linear in N log N; sharded keys likewise):
 *   - the h_query in EVALUATION form over the coset of r1cs_to_qap.rs:164-169, so that the H MSM of prover.rs:186-187 takes the
 *     pointwise values (a b - c) / Z(g) and the closing coset_ifft is not run;
 *   - the l_query with the C matrix folded in (L'_m = L_m - sum_k C_km G_k), so that neither C z nor the ifft / coset_fft of c
 *     (r1cs_to_qap.rs:155-162) is run.
 * Proofs are the group elements of prover.rs:192-210 for every assignment, satisfying or not; zkp_groth16_witness_map still
 * returns h in coefficient form.  ZKP_H_LAGRANGE=0 / ZKP_C_FOLD=0 keep the key as given (21 instead of 12 transform passes). */
int32_t zkp_groth16_pk_upload(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, zkp_groth16_pk** out);
/* The same with flags (ABI 0.5).  ZKP_PK_KEEP_FORM: skip the evaluation-form transforms above — the key stays as `Parameters<E>` holds it
 * (upload ~0.65 s shorter per 2^20 constraints, a proof runs all 7 transforms of r1cs_to_qap.rs:144-169: 7.0 instead of 6.5 ms at 2^20).
 * For callers that prove once or a few times per key, e.g. the reference CLI (cli/src/zkp_prove.rs loads a key, proves, exits): the
 * transforms pay back after ~2000 proofs.  Same proof bytes either way.  Unknown flag bits -> ZKP_ERR_BAD_ARG. */
#define ZKP_PK_KEEP_FORM 1u
int32_t zkp_groth16_pk_upload_ex(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, uint32_t flags, zkp_groth16_pk** out);
int32_t zkp_groth16_pk_free(zkp_ctx* ctx, zkp_groth16_pk* pk);

/* R1CStoQAP::witness_map (groth16/src/r1cs_to_qap.rs:113-172): z (num_inputs+num_aux Fr, Montgomery)
 * -> h (domain_size Fr, Montgomery).  *_dev: both pointers are device memory. */
int32_t zkp_groth16_witness_map(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_host, uint64_t* h_host);
int32_t zkp_groth16_witness_map_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_dev, uint64_t* h_dev);
int32_t zkp_groth16_domain_size(zkp_groth16_pk* pk, uint64_t* n);
/* Window-table plan of a resident key: info[0] = window-group size k (1 = one table copy per window; > 1: the tables did not
 * fit ZKP_TABLE_BUDGET_GB / the free device memory and k consecutive windows share a copy), info[1] = bytes of the five
 * queries' tables, info[2..4] = window bits / windows / resident copies of the A query, info[5] = window bits of the B queries,
 * info[6] = sort sharing (bit 0: B1 reuses B2's bucket sort, bit 1: L reuses A's, bit 2: shared level-1 pass), info[7] = key form (bit 0: H query in evaluation form, bit 1: C folded into the L query, bit 2: L and H
 * bucket-chained).  With any bit of info[7] set, slots L (3) and H (4) of zkp_groth16_prove_partials_dev are NOT the reference's
 * l_aux_acc / h_acc individually — only slot 3 + slot 4 equals l_aux_acc + h_acc, which is all prover.rs:189-196 consumes
 * (ZKP_H_LAGRANGE=0 ZKP_C_FOLD=0 ZKP_CHAIN_LH=0 at upload restores the individual values). */
int32_t zkp_groth16_pk_info(zkp_ctx* ctx, zkp_groth16_pk* pk, uint64_t info[8]);

/* create_proof(params, circuit, r, s): z = full assignment (Montgomery), r/s Fr Montgomery (4 limbs).
 * proof_out: A (G1 affine) | B (G2 affine) | C (G1 affine), Montgomery; inf_out[3] identity flags. */
int32_t zkp_groth16_prove(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_host, const uint64_t* r,
                          const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);
int32_t zkp_groth16_prove_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_dev, const uint64_t* r,
                              const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);

/* Throughput mode: n independent proofs with the same key, software-pipelined inside the library over two
 * "lanes" (stream sets + scratch), so that the latency-bound tail of proof i (log-depth bucket reduction, the two
 * dynamic scalar multiplications of the assembly) overlaps the throughput-bound kernels of proof i+1.
 * z_dev: n device pointers (may repeat); r, s: n x 4 limbs (host); proofs_out: n x (A|B|C); inf_out: n x 3. */
int32_t zkp_groth16_prove_batch_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, size_t n, const uint64_t* const* z_dev,
                                    const uint64_t* r, const uint64_t* s, uint64_t* proofs_out, uint8_t* inf_out);

/* The same with the witnesses in HOST memory: each proof's assignment (num_inputs + num_aux Fr) is copied to the device on
 * its lane's stream in front of the proof, so the PCIe transfer of proof i+1 overlaps the kernels of proof i when the
 * host buffers are pinned (hipHostMalloc / hipHostRegister); pageable buffers work but the copy then blocks the caller. */
int32_t zkp_groth16_prove_batch(zkp_ctx* ctx, zkp_groth16_pk* pk, size_t n, const uint64_t* const* z_host,
                                const uint64_t* r, const uint64_t* s, uint64_t* proofs_out, uint8_t* inf_out);

/* Multi-GPU (one process per GPU, bases sharded by index): every rank computes partial sums with zkp_msm_*,
 * the host all-gathers them (RCCL / any transport: 5 points, < 2 KiB), folds them with zkp_g*_fold and finishes
 * here.  sums_xyz: Jacobian Montgomery  g_a (G1) | g1_b (G1) | g2_b (G2) | h_acc (G1) | l_acc (G1)  where the
 * key-point terms of prover.rs:165-177,183 are already folded in (see DESIGN.md "Folding").  Computes
 * C = s*g_a + r*g1_b + l_acc + h_acc and the three into_affine() (prover.rs:192-210). */
int32_t zkp_groth16_assemble(zkp_ctx* ctx, zkp_curve_t curve, const uint64_t* sums_xyz, const uint64_t* r,
                             const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);

/* Device-resident base-sharded step (BASELINE configs[4]) — nothing but the collective between the calls:
 *   zkp_groth16_pk_upload_shard     rank `rank` of `world` keeps elements [lo, hi) of every query of prover.rs:164-190
 *                                   resident (contiguous, balanced ranges over the extended queries; 1/world of the key)
 *                                   plus the circuit matrices;
 *   zkp_groth16_prove_partials_dev  witness map (replicated) + the five partial MSMs of this rank; r, s (host, Montgomery)
 *                                   must be the same on every rank.  partials_dev (DEVICE, zkp_groth16_partials_bytes()
 *                                   bytes: 5 XYZZ slots A|B1|B2|H|L) is complete when the call returns; only the slot-wise
 *                                   SUMS over the ranks are specified — since round 3 the H MSM reduces L's buckets
 *                                   together with its own (slot H = h_acc + l', slot L = the identity; C uses their sum);
 *   -- ncclAllGather(partials_dev -> gathered_dev, world x partials bytes) over RCCL, by the caller --
 *   zkp_groth16_fold_assemble_dev   slot-wise sum over the ranks (EC addition is not an RCCL reduction op) + assembly of
 *                                   prover.rs:192-210 -> proof_out (host).
 * A sharded key is refused by zkp_groth16_prove*. */
int32_t zkp_groth16_pk_upload_shard(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, int32_t rank, int32_t world,
                                    zkp_groth16_pk** out);
int32_t zkp_groth16_partials_bytes(zkp_curve_t curve, size_t* bytes);
int32_t zkp_groth16_prove_partials_dev(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_dev, const uint64_t* r,
                                       const uint64_t* s, void* partials_dev);
int32_t zkp_groth16_fold_assemble_dev(zkp_ctx* ctx, zkp_curve_t curve, const void* gathered_dev, int32_t world,
                                      const uint64_t* r, const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);

/* ---- Single-process multi-GPU: `create_proof` (groth16/src/prover.rs:124) is ONE call in ONE process, so the library owns
 * one context per device and the exchange step (zkp_ctx_create(ctx**, device_ids, n_devices)).
 *   zkp_ctx_create_multi   root context = rank 0 (device_ids[0]) + one member context per further id, rank order = argument
 *                          order.  Ids may repeat (several ranks on one GPU: one-GPU test boxes).  Peer access between
 *                          distinct devices is enabled when the topology offers it (xGMI).  The root is an ordinary
 *                          zkp_ctx for every other call; zkp_ctx_destroy(root) destroys the members.
 *   zkp_ctx_device         borrow member `rank` (rank 0 = the root itself) for zkp_dev_alloc / zkp_h2d / ... on that device.
 *   zkp_groth16_pk_upload_multi
 *       ZKP_MULTI_SHARD      every (extended) query of prover.rs:164-190 split by index, rank k keeps 1/n of the window
 *                            tables (BASELINE configs[4]: keys that do not fit one GPU; lower single-proof latency);
 *       ZKP_MULTI_REPLICATE  the whole key on every device (throughput: independent proofs on independent GPUs).
 *   zkp_groth16_prove_multi        (SHARD key) ONE proof over all devices: partial MSMs per device, the five partial sums
 *                          exchanged over xGMI — an RCCL ncclAllGather of n x zkp_groth16_partials_bytes whenever the
 *                          devices are distinct and librccl.so loads (dlopen, no link-time dependency), else
 *                          hipMemcpyPeerAsync to rank 0, said once on stderr (ZKP_MULTI_EXCHANGE=rccl / peer forces one) —
 *                          then slot-wise fold + assembly on rank 0.  With n >= 3 the witness map can be task-split: the
 *                          a / b / c chains of r1cs_to_qap.rs:144-162 on ranks 0 / 1 / 2, rank 0 finishes h and every rank
 *                          fetches its slice of h for its share of the H MSM.  Whether that beats the replicated map depends
 *                          on the link, so the key measures both on its own proofs 3 and 4 (1 and 2 warm them up) and keeps
 *                          the faster from proof 5 on; the proof bytes do not depend on it (ZKP_MULTI_WM_SPLIT=0 / 1
 *                          forces).  z_on_device == 0: z[0] is the host assignment; != 0: z[k] is a device pointer on rank
 *                          k's device, k < n.
 *   zkp_groth16_multi_info  what the last zkp_groth16_prove_multi did: info[0] = exchange (0 peer copies, 1 RCCL all-gather, 2 peer copies because
 *                          the RCCL watchdog gave up: ncclCommInitAll or the probe all-gather failed or did not return within
 *                          zkp_ctx_config.multi_exchange_timeout_ms — RCCL then stays off for the process),
 *                          info[1] = RCCL ranks, info[2] = witness map (0 replicated, 1 split, 2 still measuring), info[3] /
 *                          info[4] = microseconds of the timed proof with the replicated / split map (0 = not measured),
 *                          info[5] = devices.
 *   zkp_groth16_prove_batch_multi  (REPLICATE key) `count` independent proofs, proof i on rank i % n (z[i] host, or a device
 *                          pointer on that rank's device), one host thread per device driving its lanes as
 *                          zkp_groth16_prove_batch does; outputs in input order. */
typedef struct zkp_groth16_pk_multi zkp_groth16_pk_multi;
typedef enum { ZKP_MULTI_SHARD = 0, ZKP_MULTI_REPLICATE = 1 } zkp_multi_mode;
int32_t zkp_ctx_create_multi(zkp_ctx** out, const int* device_ids, int n_devices);
/* the same with a configuration applied to the root and every member (cfg == NULL: defaults) */
int32_t zkp_ctx_create_multi_ex(zkp_ctx** out, const int* device_ids, int n_devices, const zkp_ctx_config* cfg);
int32_t zkp_ctx_num_devices(zkp_ctx* ctx, int32_t* n);
int32_t zkp_ctx_device(zkp_ctx* ctx, int32_t rank, zkp_ctx** member);
int32_t zkp_groth16_pk_upload_multi(zkp_ctx* ctx, const zkp_groth16_pk_desc* desc, int32_t mode,
                                    zkp_groth16_pk_multi** out);
int32_t zkp_groth16_pk_multi_free(zkp_ctx* ctx, zkp_groth16_pk_multi* pk);
int32_t zkp_groth16_multi_info(zkp_ctx* ctx, zkp_groth16_pk_multi* pk, uint64_t info[6]);
int32_t zkp_groth16_prove_multi(zkp_ctx* ctx, zkp_groth16_pk_multi* pk, const uint64_t* const* z, int32_t z_on_device,
                                const uint64_t* r, const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);
int32_t zkp_groth16_prove_batch_multi(zkp_ctx* ctx, zkp_groth16_pk_multi* pk, size_t count, const uint64_t* const* z,
                                      int32_t z_on_device, const uint64_t* r, const uint64_t* s, uint64_t* proofs_out,
                                      uint8_t* inf_out);

/* ---- Marlin Fiat–Shamir RNG (host code): replaces marlin/src/fs_rng.rs:11-70 `FiatShamirRng` ----------------------------
 * merlin 2.0 transcript "MARLINSEED" -> 32-byte seed -> ChaCha20 RNG (rand_chacha 0.2).  The byte strings absorbed are
 * the caller's `to_bytes![...]` (marlin/src/lib.rs:105-158); samples come back in the ABI's field layout.
 *   new                  FiatShamirRng::from_seed(&bytes)                       fs_rng.rs:41-53
 *   absorb               seed = H(bytes || seed), re-key the ChaCha stream      fs_rng.rs:57-69
 *   rand_fr              `Fr::rand(&mut fs_rng)` (ark-ff 0.2 rejection sampling; Montgomery limbs out)
 *   sample_outside_domain  AHP::sample_element_outside_domain for a domain of 2^log_domain (ahp/verifier.rs:118-127)
 *   rand_u128            `u128::rand(&mut fs_rng)` -> out[0] = low, out[1] = high 64 bits (lib.rs:158)
 *   seed / next_u64      introspection for the parity tests
 *   zkp_merlin_oneshot   Transcript::new(label); append_message(msg_label, msg); challenge_bytes(chal_label, out) */
typedef struct zkp_fs_rng zkp_fs_rng;
int32_t zkp_fs_rng_new(const uint8_t* seed_material, size_t len, zkp_fs_rng** out);
int32_t zkp_fs_rng_free(zkp_fs_rng* rng);
int32_t zkp_fs_rng_absorb(zkp_fs_rng* rng, const uint8_t* material, size_t len);
int32_t zkp_fs_rng_seed(const zkp_fs_rng* rng, uint8_t out32[32]);
int32_t zkp_fs_rng_next_u64(zkp_fs_rng* rng, uint64_t* out);
int32_t zkp_fs_rng_rand_u128(zkp_fs_rng* rng, uint64_t out[2]);
int32_t zkp_fs_rng_rand_fr(zkp_fs_rng* rng, zkp_curve_t curve, uint64_t* out_mont);
int32_t zkp_fs_rng_sample_outside_domain(zkp_fs_rng* rng, zkp_curve_t curve, uint32_t log_domain, uint64_t* out_mont);
int32_t zkp_merlin_oneshot(const uint8_t* label, size_t label_len, const uint8_t* msg_label, size_t msg_label_len,
                           const uint8_t* msg, size_t msg_len, const uint8_t* chal_label, size_t chal_label_len,
                           uint8_t* out, size_t out_len);

/* ---- Marlin: replaces zkp_marlin::index (device half) and zkp_marlin::create_random_proof (marlin/src/lib.rs:69-181) ----
 * The caller keeps synthesis and the index-manipulation half of AHP::index (make_matrices_square, balance_matrices, per-row
 * column sort: ahp/constraint_systems.rs:9-31,100-133) and hands over the three square matrices as CSR; the library computes
 * the arithmetization (row / col / val / row_col over K, their interpolations and evaluations over B: arithmetic.rs:98-172)
 * on the device and keeps it resident.  `zkp_marlin_prove` runs prover_init and the three AHP rounds (ahp/prover.rs:86-427),
 * PC::commit after each round (pc/mod.rs:34-71, MSMs on the resident SRS powers), the Fiat–Shamir transcript
 * (fs_rng.rs; lib.rs:105-158), the 21 evaluations and PC::batch_open (pc/mod.rs:73-160).  The zk randomness the reference
 * draws from `zk_rng` (mask polynomial, the three degree-0 masks, the hiding-bound-1 commitment blinders) is an input. */
typedef struct zkp_marlin_index zkp_marlin_index;
typedef struct {
  zkp_curve_t curve;
  uint32_t num_inputs; /* formatted public inputs incl. the leading one */
  uint32_t n;          /* rows == columns after make_matrices_square */
  uint32_t pad_aux;    /* dummy witness variables (value one) make_matrices_square appended */
  zkp_csr a, b, c;     /* n rows each, balanced, columns ascending per row, col < n; coeffs Fr Montgomery */
} zkp_marlin_index_desc;
#define ZKP_MARLIN_NUM_EVALS 21
typedef struct {
  const uint64_t* w;   /* 1 Fr (Montgomery) each: the degree-0 masks of prover.rs:190,196,200 */
  const uint64_t* z_a;
  const uint64_t* z_b;
  const uint64_t* mask;    /* 3|H| Fr: DensePolynomial::rand of prover.rs:202-203 (host, or device if mask_on_device) */
  int32_t mask_on_device;
  const uint64_t* blind_w; /* 2 Fr each: `Rand::rand(hiding_bound = 1)` of KZG10::commit for w, z_a, z_b, g_1 and g_1's shifted commitment */
  const uint64_t* blind_z_a;
  const uint64_t* blind_z_b;
  const uint64_t* blind_g_1;
  const uint64_t* blind_shifted_g_1;
} zkp_marlin_rand;
typedef struct {
  /* commitments in oracle order w, z_a, z_b, mask | t, g_1, h_1 | g_2, h_2: G1 affine Montgomery in 12-u64 slots */
  uint64_t comm[9 * 12];
  uint8_t comm_inf[9];
  uint64_t shifted[2 * 12]; /* degree-bound commitments of g_1 and g_2 */
  uint8_t shifted_inf[2];
  uint64_t evaluations[ZKP_MARLIN_NUM_EVALS * 4]; /* query-set order = labels ascending (lib.rs:147-156), Fr Montgomery */
  uint32_t num_opening_proofs;                    /* 2 (1 if beta == gamma); query points ascending */
  uint64_t opening_w[2 * 12];
  uint8_t opening_w_inf[2];
  uint8_t opening_has_rand[2];
  uint64_t opening_rand_v[2 * 4];
  uint64_t challenges[7 * 4]; /* alpha, eta_a, eta_b, eta_c, beta, gamma, opening challenge (Fr Montgomery) as used */
} zkp_marlin_proof;
int32_t zkp_marlin_index_upload(zkp_ctx* ctx, const zkp_marlin_index_desc* desc, zkp_marlin_index** out);
int32_t zkp_marlin_index_free(zkp_ctx* ctx, zkp_marlin_index* index);
/* info[6] = |X|, |H|, |K|, |B|, max_degree (= the SRS degree `index.max_degree()` needs), num_non_zeros */
int32_t zkp_marlin_index_info(const zkp_marlin_index* index, uint64_t info[6]);
/* the 12 index commitments (lib.rs:77-83; a_row, a_col, a_val, a_row_col, b_..., c_...): G1 affine Montgomery, 12-u64 slots */
int32_t zkp_marlin_index_commit(zkp_ctx* ctx, zkp_marlin_index* index, uint64_t powers_of_g, uint64_t* comms_xy,
                                uint8_t* inf);
/* powers_of_g / powers_of_gamma_g: resident bases (zkp_bases_upload_g1) of the committer key, >= max_degree + 1 / >= 2 points.
 * ivk_bytes = to_bytes![index_verifier_key] (the caller owns and serialises the key); x: num_inputs formatted inputs
 * (leading one included), w: the witness without the make_matrices_square padding — Fr Montgomery, host.
 * fixed_challenges == NULL: create_random_proof (messages derived from the transcript).  Non-NULL (7 Fr Montgomery: alpha,
 * eta_a, eta_b, eta_c, beta, gamma, xi): TEST HOOK — the messages are taken as given, the transcript is not consulted.
 * An index owns the scratch pool and the pinned landing slots of its proofs: ONE zkp_marlin_prove at a time per index (prove
 * with the same circuit from several threads = one zkp_marlin_index_upload per thread).  On any error the call drains its
 * streams before the pool is recycled, so a failed proof never leaves work in flight on the index. */
int32_t zkp_marlin_prove(zkp_ctx* ctx, zkp_marlin_index* index, uint64_t powers_of_g, uint64_t powers_of_gamma_g,
                         const uint8_t* ivk_bytes, size_t ivk_len, const uint64_t* x, const uint64_t* w, size_t n_w,
                         const zkp_marlin_rand* rnd, const uint64_t* fixed_challenges, zkp_marlin_proof* out);

/* ---- introspection for bench.py / rocprof bookkeeping ------------------------------------------ */
typedef struct {
  float ms_total;          /* last zkp_groth16_prove*: stream time, HIP events */
  float ms_witness_map;
  float ms_msm[5];         /* A, B1, B2, H, L */
  float ms_assemble;
  float ms_msm_accumulate; /* sum over the five MSMs of the bucket-accumulate kernel (dominant kernel) */
  uint64_t msm_accumulate_launches;
  uint64_t msm_points;     /* (scalar, window) entries of the five MSMs = mixed additions of the accumulate kernels */
  float ms_msm_scan;       /* MSM "scalar scan" (digit extraction fused into the level-1 histogram + scatter passes), summed */
  uint64_t msm_scan_launches; /* MSMs that ran their own scan (B1 reuses B2's) */
  uint64_t msm_scan_bytes; /* algorithmic bytes of those scans: 2 x 32 B per scalar read + 8 B per entry written */
  float ms_msm_acc[5];     /* accumulate kernel per MSM (A, B1, B2, H, L) */
  uint64_t msm_entries[5]; /* entries (= mixed additions: identity bases and zero digits are dropped by the scan) per MSM */
} zkp_groth16_timing;
int32_t zkp_groth16_last_timing(zkp_ctx* ctx, zkp_groth16_timing* out);
/* Phases of the last zkp_marlin_prove on this context (host clock around stream synchronisations that the prover performs
 * anyway between a round and its PC::commit, marlin/src/lib.rs:105-181): rounds = AHP prover rounds (sparse products, NTTs,
 * pointwise work), commits = the batched commitment MSMs of the round + the transcript, evaluations = 21 Horner evaluations,
 * open = batch_open (linear combinations, two witness divisions, two opening MSMs). */
typedef struct {
  double ms_round[3];
  double ms_commit[3];
  double ms_evaluations;
  double ms_open;
  double ms_total;
  uint64_t commit_points;  /* sum of the lengths of the committed coefficient vectors (incl. shifted commitments) */
  uint64_t open_points;    /* lengths of the opening-witness MSMs */
  uint64_t ntt_count;      /* transforms run by the three rounds */
  uint64_t ntt_elements;   /* sum of their sizes */
} zkp_marlin_timing;
int32_t zkp_marlin_last_timing(zkp_ctx* ctx, zkp_marlin_timing* out);
int32_t zkp_set_profiling(zkp_ctx* ctx, int32_t enable); /* per-phase HIP events (adds sync points) */
/* Sustained rate (1e9 products / s) of the library's own Montgomery multipliers with every CU saturated: the integer-VALU
 * roof the MSM / NTT kernels are bound by, measured in the calling process (bench.py `valu_roof`).
 * field: 0 = Fr, 1 = Fq; unsaturated: 0 = 32-bit saturated limbs (field_dev.hpp), 1 = 29/28-bit limbs (unsat_dev.hpp). */
int32_t zkp_bench_mulmod(zkp_ctx* ctx, zkp_curve_t curve, int32_t field, int32_t unsaturated, double* gmulmod_per_s);
/* HBM bandwidth (GB/s, bytes read + bytes written) a plain streaming copy kernel reaches on this device between two freshly
 * allocated `bytes`-byte buffers (>= 1 MiB; use >= 1 GiB so that the 256 MB of Infinity Cache cannot hold it): the measured
 * peak bench.py quotes next to the nominal 8 TB/s ("also measure an on-box copy kernel and quote both"). */
int32_t zkp_bench_hbm_copy(zkp_ctx* ctx, size_t bytes, double* gbytes_per_s);

#ifdef __cplusplus
}
#endif
#endif /* ZKP_ACCEL_H */
