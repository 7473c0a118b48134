//! Raw `extern "C"` declarations — a 1:1 transcription of `include/zkp_accel.h` (the entry points the Rust side needs).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct zkp_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct zkp_groth16_pk {
    _p: [u8; 0],
}

pub const ZKP_BN254: c_int = 0;
pub const ZKP_BLS12_381: c_int = 1;

pub const ZKP_OK: i32 = 0;
pub const ZKP_ERR_BAD_ARG: i32 = -1;
pub const ZKP_ERR_UNSUPPORTED_CURVE: i32 = -2;
pub const ZKP_ERR_DOMAIN_TOO_LARGE: i32 = -3;
pub const ZKP_ERR_OOM: i32 = -4;
pub const ZKP_ERR_DEVICE: i32 = -5;
pub const ZKP_ERR_BAD_HANDLE: i32 = -6;

pub const ZKP_NTT_FFT: i32 = 0;
pub const ZKP_NTT_IFFT: i32 = 1;
pub const ZKP_NTT_COSET_FFT: i32 = 2;
pub const ZKP_NTT_COSET_IFFT: i32 = 3;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkp_csr {
    pub row_ptr: *const u32,
    pub col: *const u32,
    pub coeff: *const u64,
}

#[repr(C)]
pub struct zkp_groth16_pk_desc {
    pub curve: c_int,
    pub num_inputs: u32,
    pub num_aux: u32,
    pub num_constraints: u32,
    pub at: zkp_csr,
    pub bt: zkp_csr,
    pub ct: zkp_csr,
    pub alpha_g1: *const u64,
    pub beta_g1: *const u64,
    pub delta_g1: *const u64,
    pub beta_g2: *const u64,
    pub delta_g2: *const u64,
    pub a_query: *const u64,
    pub a_inf: *const u8,
    pub a_len: usize,
    pub b_g1_query: *const u64,
    pub b_g1_inf: *const u8,
    pub b_g1_len: usize,
    pub b_g2_query: *const u64,
    pub b_g2_inf: *const u8,
    pub b_g2_len: usize,
    pub h_query: *const u64,
    pub h_inf: *const u8,
    pub h_len: usize,
    pub l_query: *const u64,
    pub l_inf: *const u8,
    pub l_len: usize,
}

extern "C" {
    pub fn zkp_status_string(status: i32) -> *const c_char;
    pub fn zkp_ctx_create(out: *mut *mut zkp_ctx, device_id: c_int) -> i32;
    pub fn zkp_ctx_destroy(ctx: *mut zkp_ctx) -> i32;
    pub fn zkp_ctx_sync(ctx: *mut zkp_ctx) -> i32;

    pub fn zkp_ntt(ctx: *mut zkp_ctx, curve: c_int, data_host: *mut u64, log_n: u32, op: i32) -> i32;

    pub fn zkp_bases_upload_g1(ctx: *mut zkp_ctx, curve: c_int, xy: *const u64, inf: *const u8, n: usize, handle: *mut u64) -> i32;
    pub fn zkp_bases_upload_g2(ctx: *mut zkp_ctx, curve: c_int, xy: *const u64, inf: *const u8, n: usize, handle: *mut u64) -> i32;
    pub fn zkp_bases_free(ctx: *mut zkp_ctx, handle: u64) -> i32;
    pub fn zkp_bases_len(ctx: *mut zkp_ctx, handle: u64, n: *mut usize) -> i32;

    pub fn zkp_msm_g1(ctx: *mut zkp_ctx, handle: u64, offset: usize, scalars: *const u64, n: usize, out_xyz: *mut u64) -> i32;
    pub fn zkp_msm_g2(ctx: *mut zkp_ctx, handle: u64, offset: usize, scalars: *const u64, n: usize, out_xyz: *mut u64) -> i32;
    pub fn zkp_vartime_multiscalar_mul_g1(ctx: *mut zkp_ctx, handle: u64, fr: *const u64, n: usize, out_xyz: *mut u64) -> i32;
    pub fn zkp_vartime_multiscalar_mul_g2(ctx: *mut zkp_ctx, handle: u64, fr: *const u64, n: usize, out_xyz: *mut u64) -> i32;
    pub fn zkp_msm_g1_var(ctx: *mut zkp_ctx, curve: c_int, xy: *const u64, inf: *const u8, scalars: *const u64, n: usize,
                          montgomery: i32, out_xyz: *mut u64) -> i32;
    pub fn zkp_msm_g2_var(ctx: *mut zkp_ctx, curve: c_int, xy: *const u64, inf: *const u8, scalars: *const u64, n: usize,
                          montgomery: i32, out_xyz: *mut u64) -> i32;

    pub fn zkp_groth16_pk_upload(ctx: *mut zkp_ctx, desc: *const zkp_groth16_pk_desc, out: *mut *mut zkp_groth16_pk) -> i32;
    pub fn zkp_groth16_pk_free(ctx: *mut zkp_ctx, pk: *mut zkp_groth16_pk) -> i32;
    pub fn zkp_groth16_witness_map(ctx: *mut zkp_ctx, pk: *mut zkp_groth16_pk, z: *const u64, h: *mut u64) -> i32;
    pub fn zkp_groth16_prove(ctx: *mut zkp_ctx, pk: *mut zkp_groth16_pk, z: *const u64, r: *const u64, s: *const u64,
                             proof_out: *mut u64, inf_out: *mut u8) -> i32;

    // base-sharded multi-GPU step (one process per GPU; the caller runs ncclAllGather between the two calls)
    pub fn zkp_groth16_pk_upload_shard(ctx: *mut zkp_ctx, desc: *const zkp_groth16_pk_desc, rank: i32, world: i32,
                                       out: *mut *mut zkp_groth16_pk) -> i32;
    pub fn zkp_groth16_partials_bytes(curve: c_int, bytes: *mut usize) -> i32;
    pub fn zkp_groth16_prove_partials_dev(ctx: *mut zkp_ctx, pk: *mut zkp_groth16_pk, z_dev: *const u64, r: *const u64,
                                          s: *const u64, partials_dev: *mut c_void) -> i32;
    pub fn zkp_groth16_fold_assemble_dev(ctx: *mut zkp_ctx, curve: c_int, gathered_dev: *const c_void, world: i32,
                                         r: *const u64, s: *const u64, proof_out: *mut u64, inf_out: *mut u8) -> i32;
}
