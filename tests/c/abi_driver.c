/* Plain C99 consumer of include/zkp_accel.h, linked against libzkp_accel.so (no ctypes, no C++): what a cgo / Rust
 * `extern "C"` binding sees.  Usage:
 *     abi_driver probe                -> creates a context; prints "ctx=<status>"; exit 0 (status may be ZKP_ERR_DEVICE)
 *     abi_driver run <in.bin> <out.bin>
 *     abi_driver groth16 <in.bin> <out.bin>   the seam of zkp_groth16::create_proof (prover.rs:124): zkp_groth16_pk_upload ->
 *                                             zkp_groth16_witness_map -> zkp_groth16_prove -> zkp_groth16_prove_batch
 *     abi_driver marlin <in.bin> <out.bin>    the seam of zkp_marlin::create_random_proof (marlin/src/lib.rs:97): resident SRS,
 *                                             zkp_marlin_index_upload -> zkp_marlin_prove (transcript-derived challenges)
 * groth16 / marlin in.bin = a sequence of sections [u64 byte count][payload padded to 8 bytes], in the order the functions
 * below read them; out.bin = raw little-endian words (see the fwrite calls).  tests/test_gpu_cabi.py writes / checks them.
 * run mode:
 * in.bin  (little-endian u64 words): curve, log_n, n_points, n_scalars, then 2^log_n x 4 (Fr, Montgomery),
 *         n_points x 8 (G1 affine, Montgomery; BN254 only), n_points bytes padded to 8 (identity flags),
 *         n_scalars x 4 (canonical scalars)
 * out.bin: 2^log_n x 4 (coset_fft of the input), 2^log_n x 4 (coset_ifft of that = the input again),
 *          8 words affine MSM result, 1 word identity flag, 12 words Jacobian of the Montgomery-scalar entry point
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkp_accel.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int32_t st_ = (call);                                                        \
    if (st_ != ZKP_OK) {                                                         \
      fprintf(stderr, "%s -> %d (%s)\n", #call, (int)st_, zkp_status_string(st_)); \
      return 2;                                                                  \
    }                                                                            \
  } while (0)

/* ---- section reader: [u64 nbytes][payload, zero-padded to a multiple of 8] */
static void* section(FILE* f, size_t* nbytes) {
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1) {
    fprintf(stderr, "section header missing\n");
    exit(4);
  }
  const size_t padded = (size_t)((n + 7) & ~(uint64_t)7);
  void* p = malloc(padded ? padded : 8);
  if (!p || (padded && fread(p, 1, padded, f) != padded)) {
    fprintf(stderr, "short section (%lu bytes)\n", (unsigned long)n);
    exit(4);
  }
  if (nbytes) *nbytes = (size_t)n;
  return p;
}
static void read_csr(FILE* f, zkp_csr* m) {
  m->row_ptr = (const uint32_t*)section(f, NULL);
  m->col = (const uint32_t*)section(f, NULL);
  m->coeff = (const uint64_t*)section(f, NULL);
}

/* zkp_groth16::create_proof through the C boundary.  Sections: hdr (u64: curve, num_inputs, num_aux, num_constraints,
 * n_proofs) | at, bt, ct (row_ptr, col, coeff each) | alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2 | a, b_g1, b_g2, h, l
 * (xy then identity flags each) | z | r (n_proofs x 4) | s (n_proofs x 4).
 * out: N | h (N x 4) | proof of (r0, s0) by zkp_groth16_prove (8 fq words + 3 flag words) | n_proofs x the same by
 * zkp_groth16_prove_batch | the proof of (r0, s0) once more by a second zkp_groth16_prove after the batch (key reuse) | the
 * proof of (r0, s0) by zkp_groth16_prove_multi over three ranks on device 0 (zkp_ctx_create_multi with duplicate ids) */
static int groth16_mode(const char* in, const char* out) {
  FILE* f = fopen(in, "rb");
  if (!f) return 1;
  uint64_t* hdr = (uint64_t*)section(f, NULL);
  zkp_groth16_pk_desc d;
  memset(&d, 0, sizeof d);
  d.curve = (zkp_curve_t)hdr[0];
  d.num_inputs = (uint32_t)hdr[1];
  d.num_aux = (uint32_t)hdr[2];
  d.num_constraints = (uint32_t)hdr[3];
  const size_t n_proofs = (size_t)hdr[4];
  const size_t fq = d.curve == ZKP_BN254 ? 4 : 6, pw = 8 * fq;
  read_csr(f, &d.at);
  read_csr(f, &d.bt);
  read_csr(f, &d.ct);
  d.alpha_g1 = (const uint64_t*)section(f, NULL);
  d.beta_g1 = (const uint64_t*)section(f, NULL);
  d.delta_g1 = (const uint64_t*)section(f, NULL);
  d.beta_g2 = (const uint64_t*)section(f, NULL);
  d.delta_g2 = (const uint64_t*)section(f, NULL);
  size_t nb = 0;
  d.a_query = (const uint64_t*)section(f, NULL);
  d.a_inf = (const uint8_t*)section(f, &nb);
  d.a_len = nb;
  d.b_g1_query = (const uint64_t*)section(f, NULL);
  d.b_g1_inf = (const uint8_t*)section(f, &nb);
  d.b_g1_len = nb;
  d.b_g2_query = (const uint64_t*)section(f, NULL);
  d.b_g2_inf = (const uint8_t*)section(f, &nb);
  d.b_g2_len = nb;
  d.h_query = (const uint64_t*)section(f, NULL);
  d.h_inf = (const uint8_t*)section(f, &nb);
  d.h_len = nb;
  d.l_query = (const uint64_t*)section(f, NULL);
  d.l_inf = (const uint8_t*)section(f, &nb);
  d.l_len = nb;
  const uint64_t* z = (const uint64_t*)section(f, NULL);
  const uint64_t* r = (const uint64_t*)section(f, NULL);
  const uint64_t* s = (const uint64_t*)section(f, NULL);
  fclose(f);

  zkp_ctx* ctx = NULL;
  zkp_groth16_pk* pk = NULL;
  CHECK(zkp_ctx_create(&ctx, 0));
  CHECK(zkp_groth16_pk_upload(ctx, &d, &pk));
  uint64_t N = 0;
  CHECK(zkp_groth16_domain_size(pk, &N));
  uint64_t* h = (uint64_t*)malloc((size_t)N * 32);
  CHECK(zkp_groth16_witness_map(ctx, pk, z, h));
  uint64_t* one = (uint64_t*)calloc(pw, 8);
  uint64_t* again = (uint64_t*)calloc(pw, 8);
  uint8_t one_inf[3] = {9, 9, 9}, again_inf[3] = {9, 9, 9};
  CHECK(zkp_groth16_prove(ctx, pk, z, r, s, one, one_inf));
  uint64_t* batch = (uint64_t*)calloc(n_proofs * pw, 8);
  uint8_t* batch_inf = (uint8_t*)malloc(n_proofs * 3);
  const uint64_t** zs = (const uint64_t**)malloc(n_proofs * sizeof *zs);
  for (size_t i = 0; i < n_proofs; i++) zs[i] = z;
  CHECK(zkp_groth16_prove_batch(ctx, pk, n_proofs, zs, r, s, batch, batch_inf));
  CHECK(zkp_groth16_prove(ctx, pk, z, r, s, again, again_inf));
  /* error behaviour of the seam: NULL witness, and a freed key must not be usable through a stale context call sequence */
  if (zkp_groth16_prove(ctx, pk, NULL, r, s, one, one_inf) != ZKP_ERR_BAD_ARG) return 3;
  CHECK(zkp_groth16_pk_free(ctx, pk));
  /* the single-process multi-GPU seam from C (duplicate device ids: three ranks on the one GPU of a test box): a base-sharded
   * key proves (r0, s0) over all ranks — host witness and one device witness per rank — and must reproduce `one` */
  uint64_t* multi = (uint64_t*)calloc(pw, 8);
  uint8_t multi_inf[3] = {9, 9, 9};
  {
    const int ids[3] = {0, 0, 0};
    zkp_ctx* root = NULL;
    zkp_groth16_pk_multi* mpk = NULL;
    int32_t ndev = 0;
    uint64_t minfo[6];
    CHECK(zkp_ctx_create_multi(&root, ids, 3));
    CHECK(zkp_ctx_num_devices(root, &ndev));
    if (ndev != 3) return 3;
    CHECK(zkp_groth16_pk_upload_multi(root, &d, ZKP_MULTI_SHARD, &mpk));
    const uint64_t* zh[1] = {z};
    CHECK(zkp_groth16_prove_multi(root, mpk, zh, 0, r, s, multi, multi_inf));
    if (memcmp(multi, one, pw * 8) != 0 || memcmp(multi_inf, one_inf, 3) != 0) { fprintf(stderr, "sharded proof != single-GPU proof\n"); return 5; }
    const size_t nzb = ((size_t)d.num_inputs + d.num_aux) * 32;
    void* zdev[3];
    const uint64_t* zd[3];
    for (int k = 0; k < 3; k++) {
      zkp_ctx* mem = NULL;
      CHECK(zkp_ctx_device(root, k, &mem));
      CHECK(zkp_dev_alloc(mem, nzb, &zdev[k]));
      CHECK(zkp_h2d(mem, zdev[k], z, nzb));
      zd[k] = (const uint64_t*)zdev[k];
    }
    memset(multi, 0, pw * 8);
    CHECK(zkp_groth16_prove_multi(root, mpk, zd, 1, r, s, multi, multi_inf));
    if (memcmp(multi, one, pw * 8) != 0 || memcmp(multi_inf, one_inf, 3) != 0) { fprintf(stderr, "sharded proof (device witnesses) != single-GPU proof\n"); return 5; }
    CHECK(zkp_groth16_multi_info(root, mpk, minfo));
    if (minfo[5] != 3 || minfo[0] != 0) return 3;                /* duplicate ids: peer (device-to-device) copies, never RCCL */
    if (zkp_groth16_prove_batch_multi(root, mpk, 1, zh, 0, r, s, multi, multi_inf) != ZKP_ERR_BAD_ARG) return 3;   /* SHARD key */
    for (int k = 0; k < 3; k++) {
      zkp_ctx* mem = NULL;
      CHECK(zkp_ctx_device(root, k, &mem));
      CHECK(zkp_dev_free(mem, zdev[k]));
    }
    CHECK(zkp_groth16_pk_multi_free(root, mpk));
    CHECK(zkp_ctx_destroy(root));
  }

  /* per-context configuration from C (ABI 0.6): a second context with two lanes, the key kept in coefficient form and the proof
   * points normalised on the device reads its configuration back and proves the same (r0, s0) and the same batch */
  {
    zkp_ctx_config cfg, got;
    zkp_ctx* c2 = NULL;
    zkp_groth16_pk* pk2 = NULL;
    uint64_t* p2 = (uint64_t*)calloc(n_proofs * pw, 8);
    uint8_t* p2_inf = (uint8_t*)malloc(n_proofs * 3);
    memset(&cfg, 0, sizeof cfg);
    if (zkp_ctx_create_ex(&c2, 0, &cfg) != ZKP_ERR_BAD_ARG) return 3;      /* struct_size == 0 */
    cfg.struct_size = (uint32_t)sizeof cfg;
    cfg.lanes = 2;
    cfg.h_evaluation_form = ZKP_OFF;
    cfg.host_affine = ZKP_OFF;
    CHECK(zkp_ctx_create_ex(&c2, 0, &cfg));
    CHECK(zkp_ctx_get_config(c2, &got));
    if (got.struct_size != sizeof got || got.lanes != 2 || got.h_evaluation_form != ZKP_OFF || got.host_affine != ZKP_OFF) return 3;
    CHECK(zkp_groth16_pk_upload(c2, &d, &pk2));
    CHECK(zkp_groth16_prove(c2, pk2, z, r, s, p2, p2_inf));
    if (memcmp(p2, one, pw * 8) != 0 || memcmp(p2_inf, one_inf, 3) != 0) { fprintf(stderr, "proof of the configured context differs\n"); return 5; }
    CHECK(zkp_groth16_prove_batch(c2, pk2, n_proofs, zs, r, s, p2, p2_inf));
    if (memcmp(p2, batch, n_proofs * pw * 8) != 0 || memcmp(p2_inf, batch_inf, n_proofs * 3) != 0) { fprintf(stderr, "batch of the configured context differs\n"); return 5; }
    CHECK(zkp_groth16_pk_free(c2, pk2));
    CHECK(zkp_ctx_destroy(c2));
    free(p2);
    free(p2_inf);
  }

  f = fopen(out, "wb");
  if (!f) return 1;
  fwrite(&N, 8, 1, f);
  fwrite(h, 32, (size_t)N, f);
  uint64_t w3[3];
  fwrite(one, 8, pw, f);
  for (int k = 0; k < 3; k++) w3[k] = one_inf[k];
  fwrite(w3, 8, 3, f);
  for (size_t i = 0; i < n_proofs; i++) {
    fwrite(batch + i * pw, 8, pw, f);
    for (int k = 0; k < 3; k++) w3[k] = batch_inf[3 * i + k];
    fwrite(w3, 8, 3, f);
  }
  fwrite(again, 8, pw, f);
  for (int k = 0; k < 3; k++) w3[k] = again_inf[k];
  fwrite(w3, 8, 3, f);
  fwrite(multi, 8, pw, f);
  for (int k = 0; k < 3; k++) w3[k] = multi_inf[k];
  fwrite(w3, 8, 3, f);
  fclose(f);
  CHECK(zkp_ctx_destroy(ctx));
  printf("ok\n");
  return 0;
}

/* zkp_marlin::create_random_proof through the C boundary (BN254 or BLS12-381 G1 SRS).  Sections: hdr (u64: curve, num_inputs,
 * n, pad_aux, n_w, srs_g points, srs_gamma_g points) | a, b, c (row_ptr, col, coeff each) | powers_of_g xy | powers_of_gamma_g
 * xy | ivk bytes | x | w | rand.w | rand.z_a | rand.z_b | rand.mask | blind_w | blind_z_a | blind_z_b | blind_g_1 |
 * blind_shifted_g_1.   out: info[6] | the zkp_marlin_proof struct, raw | 12 index commitments (12 x 12 words) + 12 flag words */
static int marlin_mode(const char* in, const char* out) {
  FILE* f = fopen(in, "rb");
  if (!f) return 1;
  uint64_t* hdr = (uint64_t*)section(f, NULL);
  zkp_marlin_index_desc d;
  memset(&d, 0, sizeof d);
  d.curve = (zkp_curve_t)hdr[0];
  d.num_inputs = (uint32_t)hdr[1];
  d.n = (uint32_t)hdr[2];
  d.pad_aux = (uint32_t)hdr[3];
  const size_t n_w = (size_t)hdr[4], n_g = (size_t)hdr[5], n_gg = (size_t)hdr[6];
  read_csr(f, &d.a);
  read_csr(f, &d.b);
  read_csr(f, &d.c);
  const uint64_t* g_xy = (const uint64_t*)section(f, NULL);
  const uint64_t* gg_xy = (const uint64_t*)section(f, NULL);
  size_t ivk_len = 0;
  const uint8_t* ivk = (const uint8_t*)section(f, &ivk_len);
  const uint64_t* x = (const uint64_t*)section(f, NULL);
  const uint64_t* w = (const uint64_t*)section(f, NULL);
  zkp_marlin_rand R;
  memset(&R, 0, sizeof R);
  R.w = (const uint64_t*)section(f, NULL);
  R.z_a = (const uint64_t*)section(f, NULL);
  R.z_b = (const uint64_t*)section(f, NULL);
  R.mask = (const uint64_t*)section(f, NULL);
  R.mask_on_device = 0;
  R.blind_w = (const uint64_t*)section(f, NULL);
  R.blind_z_a = (const uint64_t*)section(f, NULL);
  R.blind_z_b = (const uint64_t*)section(f, NULL);
  R.blind_g_1 = (const uint64_t*)section(f, NULL);
  R.blind_shifted_g_1 = (const uint64_t*)section(f, NULL);
  fclose(f);

  zkp_ctx* ctx = NULL;
  zkp_marlin_index* ix = NULL;
  uint64_t h_g = 0, h_gg = 0, info[6];
  CHECK(zkp_ctx_create(&ctx, 0));
  CHECK(zkp_bases_upload_g1(ctx, d.curve, g_xy, NULL, n_g, &h_g));
  CHECK(zkp_bases_upload_g1(ctx, d.curve, gg_xy, NULL, n_gg, &h_gg));
  CHECK(zkp_marlin_index_upload(ctx, &d, &ix));
  CHECK(zkp_marlin_index_info(ix, info));
  uint64_t* icomm = (uint64_t*)calloc(12 * 12, 8);
  uint8_t icomm_inf[12];
  CHECK(zkp_marlin_index_commit(ctx, ix, h_g, icomm, icomm_inf));
  zkp_marlin_proof* proof = (zkp_marlin_proof*)calloc(1, sizeof *proof);
  CHECK(zkp_marlin_prove(ctx, ix, h_g, h_gg, ivk, ivk_len, x, w, n_w, &R, NULL, proof));
  if (zkp_marlin_prove(ctx, ix, h_g, h_gg, NULL, 0, x, w, n_w, &R, NULL, proof) != ZKP_ERR_BAD_ARG) return 3; /* no key bytes, no fixed challenges */
  CHECK(zkp_marlin_index_free(ctx, ix));
  CHECK(zkp_bases_free(ctx, h_g));
  CHECK(zkp_bases_free(ctx, h_gg));

  f = fopen(out, "wb");
  if (!f) return 1;
  fwrite(info, 8, 6, f);
  uint64_t sz = sizeof *proof;
  fwrite(&sz, 8, 1, f);
  fwrite(proof, 1, sizeof *proof, f);
  fwrite(icomm, 8, 12 * 12, f);
  for (int k = 0; k < 12; k++) {
    uint64_t v = icomm_inf[k];
    fwrite(&v, 8, 1, f);
  }
  fclose(f);
  CHECK(zkp_ctx_destroy(ctx));
  printf("ok\n");
  return 0;
}

int main(int argc, char** argv) {
  zkp_ctx* ctx = NULL;
  if (argc == 4 && strcmp(argv[1], "groth16") == 0) return groth16_mode(argv[2], argv[3]);
  if (argc == 4 && strcmp(argv[1], "marlin") == 0) return marlin_mode(argv[2], argv[3]);
  if (argc >= 2 && strcmp(argv[1], "probe") == 0) {
    int32_t st = zkp_ctx_create(&ctx, 0);
    printf("version=%s\nctx=%d (%s)\n", zkp_version(), (int)st, zkp_status_string(st));
    if (st == ZKP_OK) zkp_ctx_destroy(ctx);
    return 0;
  }
  if (argc != 4 || strcmp(argv[1], "run") != 0) {
    fprintf(stderr, "usage: %s probe | run|groth16|marlin in.bin out.bin\n", argv[0]);
    return 1;
  }
  FILE* f = fopen(argv[2], "rb");
  if (!f) return 1;
  uint64_t hdr[4];
  if (fread(hdr, 8, 4, f) != 4) return 1;
  const zkp_curve_t curve = (zkp_curve_t)hdr[0];
  const uint32_t log_n = (uint32_t)hdr[1];
  const size_t n = (size_t)1 << log_n, np = (size_t)hdr[2], ns = (size_t)hdr[3];
  const size_t inf_words = (np + 7) / 8;
  uint64_t* fr = (uint64_t*)malloc(n * 32);
  uint64_t* pts = (uint64_t*)malloc(np * 64 + 8);
  uint64_t* infw = (uint64_t*)malloc(inf_words * 8 + 8);
  uint64_t* sc = (uint64_t*)malloc(ns * 32 + 8);
  if (fread(fr, 32, n, f) != n || fread(pts, 64, np, f) != np || fread(infw, 8, inf_words, f) != inf_words ||
      fread(sc, 32, ns, f) != ns)
    return 1;
  fclose(f);

  CHECK(zkp_ctx_create(&ctx, 0));
  uint64_t* a = (uint64_t*)malloc(n * 32);
  uint64_t* b = (uint64_t*)malloc(n * 32);
  memcpy(a, fr, n * 32);
  CHECK(zkp_ntt(ctx, curve, a, log_n, ZKP_NTT_COSET_FFT));
  memcpy(b, a, n * 32);
  CHECK(zkp_ntt(ctx, curve, b, log_n, ZKP_NTT_COSET_IFFT));
  if (zkp_ntt(ctx, curve, b, 40, ZKP_NTT_FFT) != ZKP_ERR_DOMAIN_TOO_LARGE) return 3;   /* PolynomialDegreeTooLarge */

  uint64_t handle = 0, xyz[12], xy[8], xyz_mont[12];
  uint8_t inf = 0;
  size_t len = 0;
  CHECK(zkp_bases_upload_g1(ctx, curve, pts, (const uint8_t*)infw, np, &handle));
  CHECK(zkp_bases_len(ctx, handle, &len));
  if (len != np) return 3;
  CHECK(zkp_msm_g1(ctx, handle, 0, sc, ns, xyz));
  CHECK(zkp_g1_into_affine(ctx, curve, xyz, xy, &inf));
  /* the same scalars through the device-pointer entry point (explicit allocation + copy) */
  void* sdev = NULL;
  CHECK(zkp_dev_alloc(ctx, ns * 32, &sdev));
  CHECK(zkp_h2d(ctx, sdev, sc, ns * 32));
  CHECK(zkp_msm_g1_dev(ctx, handle, 0, (const uint64_t*)sdev, ns, xyz_mont));
  CHECK(zkp_dev_free(ctx, sdev));
  if (zkp_msm_g2(ctx, handle, 0, sc, ns, xyz) != ZKP_ERR_BAD_HANDLE) return 3;          /* G1 handle on a G2 entry point */
  CHECK(zkp_bases_free(ctx, handle));
  if (zkp_bases_free(ctx, handle) != ZKP_ERR_BAD_HANDLE) return 3;

  f = fopen(argv[3], "wb");
  if (!f) return 1;
  uint64_t infw_out = inf;
  fwrite(a, 32, n, f);
  fwrite(b, 32, n, f);
  fwrite(xy, 8, 8, f);
  fwrite(&infw_out, 8, 1, f);
  fwrite(xyz_mont, 8, 12, f);
  fclose(f);
  CHECK(zkp_ctx_destroy(ctx));
  printf("ok\n");
  return 0;
}
