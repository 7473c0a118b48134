// Internal (non-ABI) entry points shared between translation units.
#pragma once
#include "ctx.hpp"

namespace zkp {

// ntt.hip
void ntt_run(zkp_ctx* ctx, int curve, uint32_t* data_dev, int log_n, int op);
// count <= 4 independent transforms of the same size and kind in the same launches (grid.y)
void ntt_run_batch(zkp_ctx* ctx, int curve, uint32_t* const* data_dev, int count, int log_n, int op);
// fused witness-map chains (ntt.hip); return false / nullptr when the fused form is unavailable (domain above the full-table
// limit, ZKP_NTT_FUSE=0): the caller then uses the separate transforms
bool ntt_ifft_coset_fft(zkp_ctx* ctx, int curve, uint32_t* data, int log_n);
uint32_t* ntt_qap_coset_ifft(zkp_ctx* ctx, int curve, uint32_t* a, uint32_t* b, uint32_t* c, const uint32_t* zinv, int log_n);
void ntt_free_tables(zkp_ctx* ctx);

// poly.hip (all pointers device memory unless *_host)
void fr_vec_op(zkp_ctx* ctx, int curve, int op, const uint64_t* a, const uint64_t* b, const uint64_t* k_host,
               uint64_t* out, size_t n);
void fr_batch_inverse(zkp_ctx* ctx, int curve, uint64_t* v, size_t n);
void fr_spmv(zkp_ctx* ctx, int curve, const uint32_t* row_ptr, const uint32_t* col, const uint64_t* coeff, size_t nrows,
             const uint64_t* x, uint64_t* out);
void fr_gather(zkp_ctx* ctx, const uint64_t* in, const int32_t* idx, size_t n, uint64_t* out);
void poly_vanishing_fold(zkp_ctx* ctx, int curve, const uint64_t* p, size_t len, size_t n, uint64_t* q, uint64_t* rem);
void poly_div_linear(zkp_ctx* ctx, int curve, const uint64_t* p, size_t n, const uint64_t* z_host, uint64_t* q,
                     uint64_t* eval_out_host);
void marlin_round2_prod(zkp_ctx* ctx, int curve, const uint64_t* ra, const uint64_t* za, const uint64_t* zb, const uint64_t* t,
                        const uint64_t* z, const uint64_t* k_host, uint64_t* out, size_t n);
void marlin_t3_evals(zkp_ctx* ctx, int curve, const uint64_t* const* on_k, const uint64_t* k_host, uint64_t* out, size_t n);
void marlin_h2_numerator(zkp_ctx* ctx, int curve, const uint64_t* const* on_b, const uint64_t* t, const uint64_t* k_host, uint64_t* out,
                         size_t n);
void poly_evaluate_batch(zkp_ctx* ctx, int curve, size_t count, const uint64_t* const* p, const size_t* n, const uint64_t* z_host,
                         uint64_t* out_host);

// msm.hip
// c_hint == -1: a stand-alone base vector (zkp_bases_upload_*): the lone-MSM window rule (msm.hip pick_window_bits);
// c_hint > 0: window bits chosen by the caller (Groth16: the B queries are sized by their NON-identity bases); cap_hint > 0:
// entries per accumulate task
// lgk_hint >= 0: window-group size 2^lgk chosen by the caller (a Groth16 key sizes all five queries together); -1: chosen here
// from ZKP_TABLE_BUDGET_GB / the free device memory (msm.hip BasesEntry::lgk)
uint64_t bases_upload(zkp_ctx* ctx, int curve, int group, const uint64_t* xy_host, const uint8_t* inf_host, size_t n,
                      int c_hint = 0, int cap_hint = 0, int lgk_hint = -1);
// bytes of the resident window tables of n points at group size 2^lgk, and the smallest lgk whose tables (sum over `count`
// queries) fit the budget
size_t bases_table_bytes(zkp_ctx* ctx, int curve, int group, size_t n, int lgk);
int bases_plan_lgk(zkp_ctx* ctx, int curve, const int* groups, const size_t* ns, int count);
void bases_free(zkp_ctx* ctx, uint64_t handle);
uint64_t bases_share(zkp_ctx* dst, zkp_ctx* src, uint64_t handle);
size_t bases_len(zkp_ctx* ctx, uint64_t handle);
int bases_group(zkp_ctx* ctx, uint64_t handle);
// result -> host Jacobian (out_xyz_host) ; if out_dev_xyzz != nullptr the XYZZ result is also left on device
// runs on workspace `ws` (its stream + scratch); does not synchronise unless out_xyz_host != nullptr
void msm_run(zkp_ctx* ctx, uint64_t handle, size_t offset, const uint64_t* scalars_dev, size_t n, bool montgomery,
             uint64_t* out_xyz_host, void* out_dev_xyzz = nullptr, float* ms_accumulate = nullptr,
             uint64_t* n_entries = nullptr, int ws = 0, int sort_src = -1, float* ms_scan = nullptr, int l1_src = -1);
// bench_kern.hip: sustained rate (1e9 products/s) of the library's Montgomery multipliers; field 0 = Fr, 1 = Fq
double bench_mulmod(zkp_ctx* ctx, int curve, int field, bool unsaturated);
// bench_kern.hip: GB/s (read + written) of a streaming device-to-device copy kernel over two `bytes`-byte buffers
double bench_hbm_copy(zkp_ctx* ctx, size_t bytes);
// sort_src >= 0: reuse the bucket sort + task schedule that workspace `sort_src` of the same lane computed for the SAME
// scalars, length, window configuration and identity flags (Groth16: b_g1_query / b_g2_query) instead of redoing it
bool bases_same_shape(zkp_ctx* ctx, uint64_t h1, uint64_t h2);
// flags (n bytes, host) the digit scan of `handle` uses instead of the entry's own identity flags when it sorts for a group
// of MSMs over the same scalars (a point is dropped only if it is the identity in every member)
void bases_set_sort_flags(zkp_ctx* ctx, uint64_t handle, const uint8_t* flags_host, size_t n);
// l1_src (msm_run): workspace of the same lane whose LEVEL-1 sort output (entries of a group of queries over the same scalars,
// scattered into bins) this MSM shares; it then runs its own level 2 and drops its own identities there.
// bases_set_group: on the query that runs the shared pass; member_flags_host[i] bit k = base i is the identity in member k
// (k < 3) although the group's scan (bases_set_sort_flags) keeps it.  bases_set_filter_bit: a member's k (-1 = none).
void bases_set_group(zkp_ctx* ctx, uint64_t owner, const uint8_t* member_flags_host, size_t n);
void bases_set_filter_bit(zkp_ctx* ctx, uint64_t handle, int bit);
void msm_run_batch(zkp_ctx* ctx, uint64_t handle, size_t count, const size_t* offsets, const uint64_t* const* scalars_dev,
                   const size_t* ns, bool montgomery, uint64_t* out_xyz_host);
void msm_run_multi(zkp_ctx* ctx, size_t count, const uint64_t* handles, const size_t* offsets,
                   const uint64_t* const* scalars_dev, const size_t* ns, bool montgomery, uint64_t* out_xyz_host,
                   size_t slot_words);
void msm_var_run(zkp_ctx* ctx, int curve, int group, const uint64_t* xy_host, const uint8_t* inf_host,
                 const uint64_t* scalars_host, size_t n, bool montgomery, uint64_t* out_xyz_host);
void msm_free_all(zkp_ctx* ctx);
void bases_drop(zkp_ctx* ctx, uint64_t handle);
void point_fold(zkp_ctx* ctx, int curve, int group, const uint64_t* xyz_host, size_t k, uint64_t* out_xyz_host);
void point_into_affine(zkp_ctx* ctx, int curve, int group, const uint64_t* xyz_host, uint64_t* xy_out, uint8_t* inf_out);
size_t points_decompress(zkp_ctx* ctx, int curve, int group, const uint8_t* bytes, size_t n, uint64_t* xy_out, uint8_t* inf_out);
void points_compress(zkp_ctx* ctx, int curve, int group, const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* bytes_out);
size_t points_subgroup_check(zkp_ctx* ctx, int curve, int group, const uint64_t* xy, const uint8_t* inf, size_t n);
// k Jacobian points a_i (+ b_i where has_b[i]) -> affine, one launch; host in / host out (slot strides: 3 fN / 2 fN words)
void points_fold_into_affine(zkp_ctx* ctx, int curve, int group, const uint64_t* a_xyz_host, const uint64_t* b_xyz_host,
                             const uint8_t* has_b, size_t k, uint64_t* xy_out, uint8_t* inf_out);
void fixed_base_mul(zkp_ctx* ctx, int curve, int group, const uint64_t* base_xy, const uint64_t* scalars_host, size_t n,
                    uint64_t* out_xy, uint8_t* out_inf);

// groth16.hip
zkp_groth16_pk* groth16_pk_upload(zkp_ctx* ctx, const zkp_groth16_pk_desc* d, int rank = 0, int world = 0, int flags = 0);
size_t groth16_partials_bytes(int curve);
void groth16_prove_partials(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z_dev, const uint64_t* r, const uint64_t* s,
                            void* out_dev);
void groth16_fold_assemble(zkp_ctx* ctx, int curve, const void* gathered_dev, int world, const uint64_t* r,
                           const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);
void groth16_pk_free(zkp_ctx* ctx, zkp_groth16_pk* pk);
uint64_t groth16_domain_size(zkp_groth16_pk* pk);
void groth16_pk_info(zkp_ctx* ctx, zkp_groth16_pk* pk, uint64_t info[8]);
void bases_info(zkp_ctx* ctx, uint64_t handle, uint64_t info[5]);   // c, W, k, copies, table bytes
void groth16_witness_map(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, uint64_t* h, bool on_device);
void groth16_prove(zkp_ctx* ctx, zkp_groth16_pk* pk, const uint64_t* z, bool z_on_device, const uint64_t* r,
                   const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);

void groth16_prove_batch(zkp_ctx* ctx, zkp_groth16_pk* pk, size_t n, const uint64_t* const* z_dev, const uint64_t* r,
                         const uint64_t* s, uint64_t* proofs_out, uint8_t* inf_out, bool z_on_device = true);
void groth16_assemble(zkp_ctx* ctx, int curve, const uint64_t* sums_xyz, const uint64_t* r, const uint64_t* s,
                      uint64_t* proof_out, uint8_t* inf_out);

// single-process multi-GPU (groth16.hip): `root` is a context made by zkp_ctx_create_multi; mode 0 = base-sharded key (one
// proof over all devices), 1 = replicated key (independent proofs round-robin)
zkp_groth16_pk_multi* groth16_pk_upload_multi(zkp_ctx* root, const zkp_groth16_pk_desc* d, int mode);
void groth16_pk_multi_free(zkp_ctx* root, zkp_groth16_pk_multi* pk);
void groth16_multi_info(zkp_ctx* root, zkp_groth16_pk_multi* pk, uint64_t info[6]);
void groth16_prove_multi(zkp_ctx* root, zkp_groth16_pk_multi* pk, const uint64_t* const* z, bool z_on_device, const uint64_t* r,
                         const uint64_t* s, uint64_t* proof_out, uint8_t* inf_out);
void groth16_prove_batch_multi(zkp_ctx* root, zkp_groth16_pk_multi* pk, size_t n, const uint64_t* const* z, bool z_on_device,
                               const uint64_t* r, const uint64_t* s, uint64_t* proofs_out, uint8_t* inf_out);

// marlin.hip
zkp_marlin_index* marlin_index_upload(zkp_ctx* ctx, const zkp_marlin_index_desc* d);
void marlin_index_free(zkp_ctx* ctx, zkp_marlin_index* ix);
void marlin_index_commit(zkp_ctx* ctx, zkp_marlin_index* ix, uint64_t powers_g, uint64_t* comms_xy, uint8_t* inf);
void marlin_index_info(const zkp_marlin_index* ix, uint64_t info[6]);
void marlin_prove(zkp_ctx* ctx, zkp_marlin_index* ix, uint64_t powers_g, uint64_t powers_gamma_g, const uint8_t* ivk_bytes,
                  size_t ivk_len, const uint64_t* x_mont, const uint64_t* w_mont, size_t n_w, const zkp_marlin_rand* rnd,
                  const uint64_t* fixed_challenges, zkp_marlin_proof* out);

}  // namespace zkp
