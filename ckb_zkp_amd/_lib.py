"""ctypes binding of libzkp_accel.so (include/zkp_accel.h).  There is NO CPU fallback: if the library is
missing or no gfx950 device is present, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(os.environ.get("ZKP_ACCEL_LIB") or Path(__file__).resolve().parent / "lib" / "libzkp_accel.so")   # override: A/B builds

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class ZkpError(RuntimeError):
    def __init__(self, status: int, where: str, text: str):
        super().__init__(f"{where}: status {status} ({text})")
        self.status = status


class Csr(C.Structure):
    _fields_ = [("row_ptr", vp), ("col", vp), ("coeff", vp)]


class Groth16PkDesc(C.Structure):
    _fields_ = [
        ("curve", C.c_int), ("num_inputs", C.c_uint32), ("num_aux", C.c_uint32), ("num_constraints", C.c_uint32),
        ("at", Csr), ("bt", Csr), ("ct", Csr),
        ("alpha_g1", vp), ("beta_g1", vp), ("delta_g1", vp), ("beta_g2", vp), ("delta_g2", vp),
        ("a_query", vp), ("a_inf", vp), ("a_len", C.c_size_t),
        ("b_g1_query", vp), ("b_g1_inf", vp), ("b_g1_len", C.c_size_t),
        ("b_g2_query", vp), ("b_g2_inf", vp), ("b_g2_len", C.c_size_t),
        ("h_query", vp), ("h_inf", vp), ("h_len", C.c_size_t),
        ("l_query", vp), ("l_inf", vp), ("l_len", C.c_size_t),
    ]


class Groth16Timing(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_witness_map", C.c_float), ("ms_msm", C.c_float * 5),
                ("ms_assemble", C.c_float), ("ms_msm_accumulate", C.c_float),
                ("msm_accumulate_launches", C.c_uint64), ("msm_points", C.c_uint64),
                ("ms_msm_scan", C.c_float), ("msm_scan_launches", C.c_uint64), ("msm_scan_bytes", C.c_uint64),
                ("ms_msm_acc", C.c_float * 5), ("msm_entries", C.c_uint64 * 5)]


class MarlinTiming(C.Structure):
    _fields_ = [("ms_round", C.c_double * 3), ("ms_commit", C.c_double * 3), ("ms_evaluations", C.c_double),
                ("ms_open", C.c_double), ("ms_total", C.c_double), ("commit_points", C.c_uint64),
                ("open_points", C.c_uint64), ("ntt_count", C.c_uint64), ("ntt_elements", C.c_uint64)]


class MarlinIndexDesc(C.Structure):
    _fields_ = [("curve", C.c_int), ("num_inputs", C.c_uint32), ("n", C.c_uint32), ("pad_aux", C.c_uint32),
                ("a", Csr), ("b", Csr), ("c", Csr)]


class MarlinRand(C.Structure):
    _fields_ = [("w", vp), ("z_a", vp), ("z_b", vp), ("mask", vp), ("mask_on_device", C.c_int32),
                ("blind_w", vp), ("blind_z_a", vp), ("blind_z_b", vp), ("blind_g_1", vp), ("blind_shifted_g_1", vp)]


class CtxConfig(C.Structure):
    """zkp_ctx_config (include/zkp_accel.h): 0 = default (environment variable, else built-in); tri-states 1 on / 2 off"""
    _fields_ = [("struct_size", C.c_uint32), ("lanes", C.c_int32), ("msm_batch_lanes", C.c_int32),
                ("msm_window_bits", C.c_int32), ("msm_window_bits_g2", C.c_int32), ("msm_chunk_points", C.c_int64),
                ("table_budget_gb", C.c_double), ("h_evaluation_form", C.c_int32), ("c_fold", C.c_int32),
                ("host_affine", C.c_int32), ("c_fold_heavy_cost", C.c_int64), ("multi_exchange", C.c_int32),
                ("multi_exchange_timeout_ms", C.c_int32), ("multi_witness_split", C.c_int32)]


ZKP_ON, ZKP_OFF = 1, 2
ZKP_EXCHANGE_AUTO, ZKP_EXCHANGE_RCCL, ZKP_EXCHANGE_PEER = 0, 1, 2


def make_config(config) -> "CtxConfig | None":
    """dict / CtxConfig / None -> CtxConfig with struct_size set.  Booleans map to the tri-state (True = on, False = off)."""
    if config is None:
        return None
    if isinstance(config, CtxConfig):
        cfg = config
    else:
        cfg = CtxConfig()
        names = {f[0] for f in CtxConfig._fields_} - {"struct_size"}
        for k, v in dict(config).items():
            if k not in names:
                raise KeyError(f"zkp_ctx_config has no field {k!r} (fields: {sorted(names)})")
            if isinstance(v, bool):
                v = ZKP_ON if v else ZKP_OFF
            if k == "multi_exchange" and isinstance(v, str):
                v = {"auto": ZKP_EXCHANGE_AUTO, "rccl": ZKP_EXCHANGE_RCCL, "peer": ZKP_EXCHANGE_PEER}[v]
            setattr(cfg, k, v)
    cfg.struct_size = C.sizeof(CtxConfig)
    return cfg


MARLIN_NUM_EVALS = 21


class MarlinProof(C.Structure):
    _fields_ = [("comm", C.c_uint64 * (9 * 12)), ("comm_inf", C.c_uint8 * 9),
                ("shifted", C.c_uint64 * (2 * 12)), ("shifted_inf", C.c_uint8 * 2),
                ("evaluations", C.c_uint64 * (MARLIN_NUM_EVALS * 4)), ("num_opening_proofs", C.c_uint32),
                ("opening_w", C.c_uint64 * (2 * 12)), ("opening_w_inf", C.c_uint8 * 2), ("opening_has_rand", C.c_uint8 * 2),
                ("opening_rand_v", C.c_uint64 * (2 * 4)), ("challenges", C.c_uint64 * (7 * 4))]


# name -> (restype, argtypes).  Must list every symbol declared in include/zkp_accel.h
# (tests/test_abi.py parses the header and checks both directions).
SIGNATURES = {
    "zkp_status_string": (C.c_char_p, [C.c_int32]),
    "zkp_version": (C.c_char_p, []),
    "zkp_ctx_create": (C.c_int32, [C.POINTER(vp), C.c_int]),
    "zkp_ctx_create_ex": (C.c_int32, [C.POINTER(vp), C.c_int, C.POINTER(CtxConfig)]),
    "zkp_ctx_get_config": (C.c_int32, [vp, C.POINTER(CtxConfig)]),
    "zkp_ctx_destroy": (C.c_int32, [vp]),
    "zkp_ctx_create_multi": (C.c_int32, [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]),
    "zkp_ctx_create_multi_ex": (C.c_int32, [C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.POINTER(CtxConfig)]),
    "zkp_ctx_num_devices": (C.c_int32, [vp, C.POINTER(C.c_int32)]),
    "zkp_ctx_device": (C.c_int32, [vp, C.c_int32, C.POINTER(vp)]),
    "zkp_groth16_pk_upload_multi": (C.c_int32, [vp, C.POINTER(Groth16PkDesc), C.c_int32, C.POINTER(vp)]),
    "zkp_groth16_pk_multi_free": (C.c_int32, [vp, vp]),
    "zkp_groth16_multi_info": (C.c_int32, [vp, vp, u64p]),
    "zkp_groth16_prove_multi": (C.c_int32, [vp, vp, vp, C.c_int32, vp, vp, vp, vp]),
    "zkp_groth16_prove_batch_multi": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_int32, vp, vp, vp, vp]),
    "zkp_ctx_set_stream": (C.c_int32, [vp, vp]),
    "zkp_ctx_sync": (C.c_int32, [vp]),
    "zkp_dev_alloc": (C.c_int32, [vp, C.c_size_t, C.POINTER(vp)]),
    "zkp_dev_free": (C.c_int32, [vp, vp]),
    "zkp_h2d": (C.c_int32, [vp, vp, vp, C.c_size_t]),
    "zkp_d2h": (C.c_int32, [vp, vp, vp, C.c_size_t]),
    "zkp_timer_start": (C.c_int32, [vp]),
    "zkp_timer_stop_ms": (C.c_int32, [vp, C.POINTER(C.c_float)]),
    "zkp_ntt": (C.c_int32, [vp, C.c_int, vp, C.c_uint32, C.c_int32]),
    "zkp_ntt_dev": (C.c_int32, [vp, C.c_int, vp, C.c_uint32, C.c_int32]),
    "zkp_bases_upload_g1": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_uint64)]),
    "zkp_bases_upload_g2": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_uint64)]),
    "zkp_bases_share": (C.c_int32, [vp, vp, C.c_uint64, u64p]),
    "zkp_bases_free": (C.c_int32, [vp, C.c_uint64]),
    "zkp_bases_len": (C.c_int32, [vp, C.c_uint64, C.POINTER(C.c_size_t)]),
    "zkp_msm_g1": (C.c_int32, [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, vp]),
    "zkp_msm_g2": (C.c_int32, [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, vp]),
    "zkp_msm_g1_dev": (C.c_int32, [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, vp]),
    "zkp_msm_g2_dev": (C.c_int32, [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, vp]),
    "zkp_vartime_multiscalar_mul_g1": (C.c_int32, [vp, C.c_uint64, vp, C.c_size_t, vp]),
    "zkp_vartime_multiscalar_mul_g2": (C.c_int32, [vp, C.c_uint64, vp, C.c_size_t, vp]),
    "zkp_msm_g1_var": (C.c_int32, [vp, C.c_int, vp, vp, vp, C.c_size_t, C.c_int32, vp]),
    "zkp_msm_g2_var": (C.c_int32, [vp, C.c_int, vp, vp, vp, C.c_size_t, C.c_int32, vp]),
    "zkp_msm_g1_mont_dev": (C.c_int32, [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, vp]),
    "zkp_msm_g2_mont_dev": (C.c_int32, [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, vp]),
    "zkp_msm_g1_mont_batch_dev": (C.c_int32, [vp, C.c_uint64, C.c_size_t, vp, vp, vp, vp]),
    "zkp_msm_mont_multi_dev": (C.c_int32, [vp, C.c_size_t, vp, vp, vp, vp, vp, C.c_size_t]),
    "zkp_fr_vec_op_dev": (C.c_int32, [vp, C.c_int, C.c_int32, vp, vp, vp, vp, C.c_size_t]),
    "zkp_fr_spmv_dev": (C.c_int32, [vp, C.c_int, vp, vp, vp, C.c_size_t, vp, vp]),
    "zkp_fr_gather_dev": (C.c_int32, [vp, vp, vp, C.c_size_t, vp]),
    "zkp_poly_divide_by_vanishing_dev": (C.c_int32, [vp, C.c_int, vp, C.c_size_t, C.c_size_t, vp, vp]),
    "zkp_d2d": (C.c_int32, [vp, vp, vp, C.c_size_t]),
    "zkp_dev_zero": (C.c_int32, [vp, vp, C.c_size_t]),
    "zkp_fr_batch_inverse_dev": (C.c_int32, [vp, C.c_int, vp, C.c_size_t]),
    "zkp_poly_evaluate_dev": (C.c_int32, [vp, C.c_int, vp, C.c_size_t, vp, vp]),
    "zkp_poly_div_linear_dev": (C.c_int32, [vp, C.c_int, vp, C.c_size_t, vp, vp, vp]),
    "zkp_g1_fold": (C.c_int32, [vp, C.c_int, vp, C.c_size_t, vp]),
    "zkp_g2_fold": (C.c_int32, [vp, C.c_int, vp, C.c_size_t, vp]),
    "zkp_g1_into_affine": (C.c_int32, [vp, C.c_int, vp, vp, vp]),
    "zkp_g1_decompress": (C.c_int32, [vp, C.c_int, vp, C.c_size_t, vp, vp, C.POINTER(C.c_size_t)]),
    "zkp_g2_decompress": (C.c_int32, [vp, C.c_int, vp, C.c_size_t, vp, vp, C.POINTER(C.c_size_t)]),
    "zkp_g1_compress": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, vp]),
    "zkp_g2_compress": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, vp]),
    "zkp_g1_subgroup_check": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zkp_g2_subgroup_check": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zkp_g2_into_affine": (C.c_int32, [vp, C.c_int, vp, vp, vp]),
    "zkp_groth16_points_into_affine": (C.c_int32, [C.c_int, vp, vp, vp, vp, vp]),
    "zkp_fixed_base_mul_g1": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, vp, vp]),
    "zkp_fixed_base_mul_g2": (C.c_int32, [vp, C.c_int, vp, vp, C.c_size_t, vp, vp]),
    "zkp_groth16_pk_upload": (C.c_int32, [vp, C.POINTER(Groth16PkDesc), C.POINTER(vp)]),
    "zkp_groth16_pk_upload_ex": (C.c_int32, [vp, C.POINTER(Groth16PkDesc), C.c_uint32, C.POINTER(vp)]),
    "zkp_groth16_pk_upload_shard": (C.c_int32, [vp, C.POINTER(Groth16PkDesc), C.c_int32, C.c_int32, C.POINTER(vp)]),
    "zkp_groth16_partials_bytes": (C.c_int32, [C.c_int, C.POINTER(C.c_size_t)]),
    "zkp_groth16_prove_partials_dev": (C.c_int32, [vp, vp, vp, vp, vp, vp]),
    "zkp_groth16_fold_assemble_dev": (C.c_int32, [vp, C.c_int, vp, C.c_int32, vp, vp, vp, vp]),
    "zkp_groth16_pk_free": (C.c_int32, [vp, vp]),
    "zkp_groth16_witness_map": (C.c_int32, [vp, vp, vp, vp]),
    "zkp_groth16_witness_map_dev": (C.c_int32, [vp, vp, vp, vp]),
    "zkp_groth16_domain_size": (C.c_int32, [vp, C.POINTER(C.c_uint64)]),
    "zkp_groth16_pk_info": (C.c_int32, [vp, vp, C.POINTER(C.c_uint64)]),
    "zkp_groth16_prove": (C.c_int32, [vp, vp, vp, vp, vp, vp, vp]),
    "zkp_groth16_prove_dev": (C.c_int32, [vp, vp, vp, vp, vp, vp, vp]),
    "zkp_groth16_prove_batch_dev": (C.c_int32, [vp, vp, C.c_size_t, vp, vp, vp, vp, vp]),
    "zkp_groth16_prove_batch": (C.c_int32, [vp, vp, C.c_size_t, vp, vp, vp, vp, vp]),
    "zkp_groth16_assemble": (C.c_int32, [vp, C.c_int, vp, vp, vp, vp, vp]),
    "zkp_groth16_last_timing": (C.c_int32, [vp, C.POINTER(Groth16Timing)]),
    "zkp_marlin_last_timing": (C.c_int32, [vp, C.POINTER(MarlinTiming)]),
    "zkp_set_profiling": (C.c_int32, [vp, C.c_int32]),
    "zkp_marlin_index_upload": (C.c_int32, [vp, C.POINTER(MarlinIndexDesc), C.POINTER(vp)]),
    "zkp_marlin_index_free": (C.c_int32, [vp, vp]),
    "zkp_marlin_index_info": (C.c_int32, [vp, u64p]),
    "zkp_marlin_index_commit": (C.c_int32, [vp, vp, C.c_uint64, vp, vp]),
    "zkp_marlin_prove": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, vp, C.c_size_t, vp, vp, C.c_size_t,
                                     C.POINTER(MarlinRand), vp, C.POINTER(MarlinProof)]),
    "zkp_fs_rng_new": (C.c_int32, [vp, C.c_size_t, C.POINTER(vp)]),
    "zkp_fs_rng_free": (C.c_int32, [vp]),
    "zkp_fs_rng_absorb": (C.c_int32, [vp, vp, C.c_size_t]),
    "zkp_fs_rng_seed": (C.c_int32, [vp, vp]),
    "zkp_fs_rng_next_u64": (C.c_int32, [vp, u64p]),
    "zkp_fs_rng_rand_u128": (C.c_int32, [vp, vp]),
    "zkp_fs_rng_rand_fr": (C.c_int32, [vp, C.c_int, vp]),
    "zkp_fs_rng_sample_outside_domain": (C.c_int32, [vp, C.c_int, C.c_uint32, vp]),
    "zkp_merlin_oneshot": (C.c_int32, [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t]),
    "zkp_bench_mulmod": (C.c_int32, [vp, C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "zkp_bench_hbm_copy": (C.c_int32, [vp, C.c_size_t, C.POINTER(C.c_double)]),
}

_lib = None


def load(path: os.PathLike | None = None) -> C.CDLL:
    """dlopen the in-tree library and attach signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise ImportError(f"{p} not found — build it with `python -m ckb_zkp_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


ZKP_ERR_INVALID_POINT = -7        # include/zkp_accel.h zkp_status


def check(status: int, where: str) -> None:
    if status != 0:
        raise ZkpError(status, where, load().zkp_status_string(status).decode())
