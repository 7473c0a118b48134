/* Plain C99 consumer of include/zkp_accel.h, linked against libzkp_accel.so (no ctypes, no C++): what a cgo / Rust
 * `extern "C"` binding sees.  Usage:
 *     abi_driver probe                -> creates a context; prints "ctx=<status>"; exit 0 (status may be ZKP_ERR_DEVICE)
 *     abi_driver run <in.bin> <out.bin>
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

int main(int argc, char** argv) {
  zkp_ctx* ctx = NULL;
  if (argc >= 2 && strcmp(argv[1], "probe") == 0) {
    int32_t st = zkp_ctx_create(&ctx, 0);
    printf("version=%s\nctx=%d (%s)\n", zkp_version(), (int)st, zkp_status_string(st));
    if (st == ZKP_OK) zkp_ctx_destroy(ctx);
    return 0;
  }
  if (argc != 4 || strcmp(argv[1], "run") != 0) {
    fprintf(stderr, "usage: %s probe | run in.bin out.bin\n", argv[0]);
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
