//! `zkp-accel`: safe Rust face of `libzkp_accel.so` (include/zkp_accel.h) for sec-bit/ckb-zkp.
//!
//! What it replaces in the reference (file:line in the ckb-zkp tree):
//!   * `ark_ec::msm::VariableBaseMSM::multi_scalar_mul`          groth16/src/prover.rs:187,190,220;
//!                                                               marlin/src/pc/kzg10.rs:109,118,137,146
//!   * `zkp_curve::Curve::vartime_multiscalar_mul`               curve/src/lib.rs:38-45
//!   * `EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place`   groth16/src/r1cs_to_qap.rs:144-169
//!   * `zkp_groth16::create_proof` after synthesis               groth16/src/prover.rs:148-210
//!
//! SOURCE ONLY — never compiled (no Rust toolchain in the authoring image).  arkworks 0.2 facts relied upon
//! [recalled, to be confirmed by the first `cargo check`]:
//!   * `Fp256<P>(pub BigInteger256, PhantomData<P>)`, `BigInteger256(pub [u64; 4])`: the public tuple field holds the
//!     MONTGOMERY representation a*R mod p, little-endian limbs — exactly the ABI's field-element layout;
//!     `Fp256::new(BigInteger256)` builds an element from that raw representation (what `field_new!` expands to);
//!     the same for `Fp384` / `BigInteger384` (BLS12-381 Fq);
//!   * `QuadExtField { pub c0, pub c1 }`;
//!   * `GroupAffine { pub x, pub y, pub infinity }` (repr(Rust): copied field by field, never transmuted),
//!     `GroupProjective::new(x, y, z)` is Jacobian (x = X/Z^2, y = Y/Z^3), identity = any Z == 0.
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;

use ark_ec::models::short_weierstrass_jacobian::{GroupAffine, GroupProjective};
use ark_ec::models::SWModelParameters;
use ark_ff::{BigInteger256, BigInteger384, Fp256, Fp256Parameters, Fp384, Fp384Parameters, QuadExtField, QuadExtParameters, Zero};

pub mod accel_cache;
pub mod ffi;
pub mod groth16;
pub mod marlin;

/// Errors of the device library (zkp_status) — see `Error::into_synthesis_error` for the mapping the reference expects.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Error {
    BadArg,
    UnsupportedCurve,
    /// == `SynthesisError::PolynomialDegreeTooLarge` (groth16/src/r1cs_to_qap.rs:123-125)
    DomainTooLarge,
    OutOfDeviceMemory,
    /// HIP runtime error or no gfx950 device.  The library has NO CPU fallback; callers that want one keep the arkworks
    /// call and choose it here (see `msm_or_arkworks` in the patch under rust/patches/).
    Device(String),
    BadHandle,
    /// a point handed to the codec entry points is malformed / off the curve / outside the subgroup (the wrappers below turn
    /// it into `PointError::Invalid(index)`)
    InvalidPoint,
    Unknown(i32),
}

impl Error {
    pub fn from_status(st: i32) -> Self {
        match st {
            ffi::ZKP_ERR_BAD_ARG => Error::BadArg,
            ffi::ZKP_ERR_UNSUPPORTED_CURVE => Error::UnsupportedCurve,
            ffi::ZKP_ERR_DOMAIN_TOO_LARGE => Error::DomainTooLarge,
            ffi::ZKP_ERR_OOM => Error::OutOfDeviceMemory,
            ffi::ZKP_ERR_DEVICE => {
                let s = unsafe { CStr::from_ptr(ffi::zkp_status_string(st)) };
                Error::Device(s.to_string_lossy().into_owned())
            }
            ffi::ZKP_ERR_BAD_HANDLE => Error::BadHandle,
            ffi::ZKP_ERR_INVALID_POINT => Error::InvalidPoint,
            other => Error::Unknown(other),
        }
    }
    /// groth16 / marlin return `SynthesisError`; only the domain-size error has a counterpart there
    /// (r1cs/src/error.rs): everything else surfaces as `IoError`-like failure of the backend.
    pub fn into_synthesis_error(self) -> zkp_r1cs::SynthesisError {
        match self {
            Error::DomainTooLarge => zkp_r1cs::SynthesisError::PolynomialDegreeTooLarge,
            other => zkp_r1cs::SynthesisError::IoError(std::io::Error::new(std::io::ErrorKind::Other, format!("zkp-accel: {:?}", other))),
        }
    }
}

fn check(st: i32) -> Result<(), Error> {
    if st == ffi::ZKP_OK {
        Ok(())
    } else {
        Err(Error::from_status(st))
    }
}

/// One device context (`zkp_ctx`): one per prover thread / per GPU is what runs concurrently.  Since ABI 0.5 the library holds a
/// per-context lock in every entry point, so a `&Ctx` shared between threads (`accel_cache::ctx()`) is safe: calls are serialised.
pub struct Ctx(pub(crate) *mut ffi::zkp_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}

impl Ctx {
    /// Export `GPU_MAX_HW_QUEUES=16` in the process environment BEFORE the first HIP call: the library
    /// does not touch the environment.
    pub fn new(device_id: i32) -> Result<Self, Error> {
        let mut p: *mut ffi::zkp_ctx = ptr::null_mut();
        check(unsafe { ffi::zkp_ctx_create(&mut p, device_id as c_int) })?;
        Ok(Ctx(p))
    }

    /// `zkp_ctx_create_ex`: a context with its own configuration (ABI 0.6).  Fields left at 0 are defaults (the environment
    /// variable named in the header when set, else the built-in choice); two contexts of one process may differ:
    /// `Ctx::with_config(0, &CtxConfig { lanes: 2, h_evaluation_form: ffi::ZKP_OFF, ..CtxConfig::default() })`.
    pub fn with_config(device_id: i32, cfg: &CtxConfig) -> Result<Self, Error> {
        let mut p: *mut ffi::zkp_ctx = ptr::null_mut();
        let raw = cfg.to_ffi();
        check(unsafe { ffi::zkp_ctx_create_ex(&mut p, device_id as c_int, &raw) })?;
        Ok(Ctx(p))
    }

    /// The resolved configuration of this context (`zkp_ctx_get_config`).
    pub fn config(&self) -> Result<ffi::zkp_ctx_config, Error> {
        let mut raw: ffi::zkp_ctx_config = unsafe { std::mem::zeroed() };
        check(unsafe { ffi::zkp_ctx_get_config(self.0, &mut raw) })?;
        Ok(raw)
    }
}

/// `zkp_ctx_config` without the `struct_size` bookkeeping (`to_ffi` fills it in).  `Default` = every field default.
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct CtxConfig {
    pub lanes: i32,
    pub msm_batch_lanes: i32,
    pub msm_window_bits: i32,
    pub msm_window_bits_g2: i32,
    pub msm_chunk_points: i64,
    pub table_budget_gb: f64,
    pub h_evaluation_form: i32,
    pub c_fold: i32,
    pub host_affine: i32,
    pub c_fold_heavy_cost: i64,
    pub multi_exchange: i32,
    pub multi_exchange_timeout_ms: i32,
    pub multi_witness_split: i32,
}
impl CtxConfig {
    pub fn to_ffi(&self) -> ffi::zkp_ctx_config {
        ffi::zkp_ctx_config {
            struct_size: std::mem::size_of::<ffi::zkp_ctx_config>() as u32,
            lanes: self.lanes,
            msm_batch_lanes: self.msm_batch_lanes,
            msm_window_bits: self.msm_window_bits,
            msm_window_bits_g2: self.msm_window_bits_g2,
            msm_chunk_points: self.msm_chunk_points,
            table_budget_gb: self.table_budget_gb,
            h_evaluation_form: self.h_evaluation_form,
            c_fold: self.c_fold,
            host_affine: self.host_affine,
            c_fold_heavy_cost: self.c_fold_heavy_cost,
            multi_exchange: self.multi_exchange,
            multi_exchange_timeout_ms: self.multi_exchange_timeout_ms,
            multi_witness_split: self.multi_witness_split,
        }
    }
}
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { ffi::zkp_ctx_destroy(self.0) };
    }
}

/// `zkp_ctx_create_multi`: ONE process, one context per listed device; the root is an ordinary context (rank 0).
/// `create_proof` (groth16/src/prover.rs:124) is one call in one process — this is how it reaches 8 GPUs without `torchrun`:
/// see `groth16::MultiDeviceProvingKey`.
pub struct MultiCtx {
    pub root: Ctx,
    pub num_devices: usize,
}
impl MultiCtx {
    pub fn new(device_ids: &[i32]) -> Result<Self, Error> {
        let ids: Vec<c_int> = device_ids.iter().map(|&d| d as c_int).collect();
        let mut p: *mut ffi::zkp_ctx = ptr::null_mut();
        check(unsafe { ffi::zkp_ctx_create_multi(&mut p, ids.as_ptr(), ids.len() as c_int) })?;
        Ok(MultiCtx { root: Ctx(p), num_devices: ids.len() })
    }

    /// `zkp_ctx_create_multi_ex`: the configuration applies to the root and to every member (e.g. `multi_exchange`,
    /// `multi_exchange_timeout_ms` for the RCCL bring-up watchdog).
    pub fn with_config(device_ids: &[i32], cfg: &CtxConfig) -> Result<Self, Error> {
        let ids: Vec<c_int> = device_ids.iter().map(|&d| d as c_int).collect();
        let mut p: *mut ffi::zkp_ctx = ptr::null_mut();
        let raw = cfg.to_ffi();
        check(unsafe { ffi::zkp_ctx_create_multi_ex(&mut p, ids.as_ptr(), ids.len() as c_int, &raw) })?;
        Ok(MultiCtx { root: Ctx(p), num_devices: ids.len() })
    }
}

// ------------------------------------------------------------------------------------------------ marshalling
/// A base-field element that can be copied to / from the ABI's little-endian Montgomery limbs.
pub trait AbiField: Sized + Copy {
    /// u64 limbs per element (4: 256-bit fields, 6: BLS12-381 Fq, 8 / 12: their quadratic extensions)
    const LIMBS: usize;
    fn write_limbs(&self, out: &mut [u64]);
    fn read_limbs(limbs: &[u64]) -> Self;
}
impl<P: Fp256Parameters> AbiField for Fp256<P> {
    const LIMBS: usize = 4;
    fn write_limbs(&self, out: &mut [u64]) {
        out[..4].copy_from_slice(&(self.0).0); // Montgomery representation, as stored
    }
    fn read_limbs(l: &[u64]) -> Self {
        Fp256::new(BigInteger256([l[0], l[1], l[2], l[3]]))
    }
}
impl<P: Fp384Parameters> AbiField for Fp384<P> {
    const LIMBS: usize = 6;
    fn write_limbs(&self, out: &mut [u64]) {
        out[..6].copy_from_slice(&(self.0).0);
    }
    fn read_limbs(l: &[u64]) -> Self {
        Fp384::new(BigInteger384([l[0], l[1], l[2], l[3], l[4], l[5]]))
    }
}
impl<P: QuadExtParameters> AbiField for QuadExtField<P>
where
    P::BaseField: AbiField,
{
    const LIMBS: usize = 2 * <P::BaseField as AbiField>::LIMBS;
    fn write_limbs(&self, out: &mut [u64]) {
        let h = <P::BaseField as AbiField>::LIMBS;
        self.c0.write_limbs(&mut out[..h]); // ABI: (c0, c1)
        self.c1.write_limbs(&mut out[h..2 * h]);
    }
    fn read_limbs(l: &[u64]) -> Self {
        let h = <P::BaseField as AbiField>::LIMBS;
        QuadExtField::new(<P::BaseField as AbiField>::read_limbs(&l[..h]), <P::BaseField as AbiField>::read_limbs(&l[h..2 * h]))
    }
}

/// `&[GroupAffine]` -> AoS (x, y) limbs + identity flags (the ABI never looks at the coordinates of a flagged point).
pub fn marshal_points<P: SWModelParameters>(pts: &[GroupAffine<P>]) -> (Vec<u64>, Vec<u8>)
where
    P::BaseField: AbiField,
{
    let l = <P::BaseField as AbiField>::LIMBS;
    let mut xy = vec![0u64; pts.len() * 2 * l];
    let mut inf = vec![0u8; pts.len()];
    for (i, p) in pts.iter().enumerate() {
        if p.infinity {
            inf[i] = 1;
        } else {
            p.x.write_limbs(&mut xy[i * 2 * l..i * 2 * l + l]);
            p.y.write_limbs(&mut xy[i * 2 * l + l..(i + 1) * 2 * l]);
        }
    }
    (xy, inf)
}

/// Jacobian (X, Y, Z) limbs as returned by `zkp_msm_*` -> `GroupProjective` (identity: Z == 0 -> `zero()`).
pub fn unmarshal_projective<P: SWModelParameters>(xyz: &[u64]) -> GroupProjective<P>
where
    P::BaseField: AbiField,
{
    let l = <P::BaseField as AbiField>::LIMBS;
    let z = <P::BaseField as AbiField>::read_limbs(&xyz[2 * l..3 * l]);
    if z.is_zero() {
        return GroupProjective::<P>::zero();
    }
    GroupProjective::new(
        <P::BaseField as AbiField>::read_limbs(&xyz[..l]),
        <P::BaseField as AbiField>::read_limbs(&xyz[l..2 * l]),
        z,
    )
}

/// affine (x, y) limbs + flag -> `GroupAffine`
pub fn unmarshal_affine<P: SWModelParameters>(xy: &[u64], infinity: bool) -> GroupAffine<P>
where
    P::BaseField: AbiField,
{
    let l = <P::BaseField as AbiField>::LIMBS;
    if infinity {
        return GroupAffine::<P>::zero();
    }
    GroupAffine::new(<P::BaseField as AbiField>::read_limbs(&xy[..l]), <P::BaseField as AbiField>::read_limbs(&xy[l..2 * l]), false)
}

/// Scalar element types whose in-memory form is exactly 4 little-endian u64 limbs (32 bytes): the only types the ABI's
/// scalar / NTT arguments accept.  SEALED: implemented here for `BigInteger256` (canonical, after `into_repr()`) and
/// `Fp256<P>` (Montgomery limbs) and nowhere else, so safe generic callers cannot hand the C library a slice of any other
/// element size (which would make it read or write `32 * n` bytes past the slice).
pub trait AbiScalar: sealed::Sealed + Copy {}
mod sealed {
    pub trait Sealed {}
}
impl sealed::Sealed for BigInteger256 {}
impl AbiScalar for BigInteger256 {}
impl<P: Fp256Parameters> sealed::Sealed for Fp256<P> {}
impl<P: Fp256Parameters> AbiScalar for Fp256<P> {}

/// A slice of scalars as the ABI's `const uint64_t*` (4 limbs per element, no copy).  The size check is a hard assertion
/// (also in release builds) on top of the sealed trait: a layout change in arkworks must fail loudly, not corrupt memory.
pub fn scalars_ptr<T: AbiScalar>(s: &[T]) -> *const u64 {
    assert_eq!(std::mem::size_of::<T>(), 32, "ABI scalars are 4 x u64");
    assert_eq!(std::mem::align_of::<T>() % std::mem::align_of::<u64>(), 0);
    s.as_ptr() as *const u64
}
fn scalars_mut_ptr<T: AbiScalar>(s: &mut [T]) -> *mut u64 {
    assert_eq!(std::mem::size_of::<T>(), 32, "ABI scalars are 4 x u64");
    s.as_mut_ptr() as *mut u64
}

// ------------------------------------------------------------------------------------------------ curves
/// Which ABI curve id / group a short-Weierstrass parameter set maps to.
pub trait AccelGroup: SWModelParameters
where
    Self::BaseField: AbiField,
{
    const CURVE: c_int;
    /// 1 = G1, 2 = G2
    const GROUP: u8;
}
#[cfg(feature = "bn254")]
impl AccelGroup for ark_bn254::g1::Parameters {
    const CURVE: c_int = ffi::ZKP_BN254;
    const GROUP: u8 = 1;
}
#[cfg(feature = "bn254")]
impl AccelGroup for ark_bn254::g2::Parameters {
    const CURVE: c_int = ffi::ZKP_BN254;
    const GROUP: u8 = 2;
}
#[cfg(feature = "bls12_381")]
impl AccelGroup for ark_bls12_381::g1::Parameters {
    const CURVE: c_int = ffi::ZKP_BLS12_381;
    const GROUP: u8 = 1;
}
#[cfg(feature = "bls12_381")]
impl AccelGroup for ark_bls12_381::g2::Parameters {
    const CURVE: c_int = ffi::ZKP_BLS12_381;
    const GROUP: u8 = 2;
}

// ------------------------------------------------------------------------------------------------ MSM
/// A query / SRS resident in HBM (`zkp_bases_upload_*`): upload once per `Parameters` / `CommitterKey`, prove many.
pub struct ResidentBases<'c, P: AccelGroup>
where
    P::BaseField: AbiField,
{
    ctx: &'c Ctx,
    handle: u64,
    len: usize,
    _p: std::marker::PhantomData<P>,
}

impl<'c, P: AccelGroup> ResidentBases<'c, P>
where
    P::BaseField: AbiField,
{
    pub fn upload(ctx: &'c Ctx, pts: &[GroupAffine<P>]) -> Result<Self, Error> {
        let (xy, inf) = marshal_points(pts);
        let mut handle = 0u64;
        let st = unsafe {
            if P::GROUP == 1 {
                ffi::zkp_bases_upload_g1(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), pts.len(), &mut handle)
            } else {
                ffi::zkp_bases_upload_g2(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), pts.len(), &mut handle)
            }
        };
        check(st)?;
        Ok(ResidentBases { ctx, handle, len: pts.len(), _p: std::marker::PhantomData })
    }
    pub fn len(&self) -> usize {
        self.len
    }

    /// drop-in for `VariableBaseMSM::multi_scalar_mul(&bases[offset..], scalars)` — canonical `BigInteger256` scalars
    /// (the output of `into_repr()`); min(len) truncation as in arkworks.
    pub fn msm(&self, offset: usize, scalars: &[BigInteger256]) -> Result<GroupProjective<P>, Error> {
        let l = <P::BaseField as AbiField>::LIMBS;
        let mut out = vec![0u64; 3 * l];
        let st = unsafe {
            if P::GROUP == 1 {
                ffi::zkp_msm_g1(self.ctx.0, self.handle, offset, scalars_ptr(scalars), scalars.len(), out.as_mut_ptr())
            } else {
                ffi::zkp_msm_g2(self.ctx.0, self.handle, offset, scalars_ptr(scalars), scalars.len(), out.as_mut_ptr())
            }
        };
        check(st)?;
        Ok(unmarshal_projective::<P>(&out))
    }

    /// `Curve::vartime_multiscalar_mul` with resident points: `Fr` elements as they sit in memory (Montgomery);
    /// `into_repr()` is fused into the device's digit scan.
    pub fn vartime_multiscalar_mul<F: AbiScalar>(&self, scalars: &[F]) -> Result<GroupProjective<P>, Error> {
        let l = <P::BaseField as AbiField>::LIMBS;
        let mut out = vec![0u64; 3 * l];
        let st = unsafe {
            if P::GROUP == 1 {
                ffi::zkp_vartime_multiscalar_mul_g1(self.ctx.0, self.handle, scalars_ptr(scalars), scalars.len(), out.as_mut_ptr())
            } else {
                ffi::zkp_vartime_multiscalar_mul_g2(self.ctx.0, self.handle, scalars_ptr(scalars), scalars.len(), out.as_mut_ptr())
            }
        };
        check(st)?;
        Ok(unmarshal_projective::<P>(&out))
    }
}
impl<'c, P: AccelGroup> ResidentBases<'c, P>
where
    P::BaseField: AbiField,
{
    /// `KZG10::commit` / `open` on a coefficient vector that already lives in HBM (marlin/src/pc/kzg10.rs:108-109,137-140):
    /// Montgomery `Fr` coefficients at `coeffs_dev` against `powers[offset ..]`, `offset` = the leading zeros skipped by
    /// `skip_leading_zeros_and_convert_to_bigints` (`into_repr()` is fused into the device's digit scan).
    ///
    /// # Safety
    /// `coeffs_dev` must be a device pointer to `n` `Fr` elements on this context's device.
    pub unsafe fn msm_mont_dev(&self, offset: usize, coeffs_dev: *const u64, n: usize) -> Result<GroupProjective<P>, Error> {
        let l = <P::BaseField as AbiField>::LIMBS;
        let mut out = vec![0u64; 3 * l];
        let st = if P::GROUP == 1 {
            ffi::zkp_msm_g1_mont_dev(self.ctx.0, self.handle, offset, coeffs_dev, n, out.as_mut_ptr())
        } else {
            ffi::zkp_msm_g2_mont_dev(self.ctx.0, self.handle, offset, coeffs_dev, n, out.as_mut_ptr())
        };
        check(st)?;
        Ok(unmarshal_projective::<P>(&out))
    }
    /// The same for host-resident Montgomery coefficients: uploads them and runs `msm_mont_dev` (what the KZG10 seam in
    /// rust/patches/marlin-accel.diff calls: no `into_repr()` pass over the polynomial on the CPU).
    pub fn msm_mont<F: AbiScalar>(&self, offset: usize, coeffs: &[F]) -> Result<GroupProjective<P>, Error> {
        if coeffs.is_empty() {
            return Ok(GroupProjective::<P>::zero());
        }
        let bytes = coeffs.len() * 32;
        let mut d: *mut std::os::raw::c_void = ptr::null_mut();
        check(unsafe { ffi::zkp_dev_alloc(self.ctx.0, bytes, &mut d) })?;
        let r = (|| {
            check(unsafe { ffi::zkp_h2d(self.ctx.0, d, coeffs.as_ptr() as *const std::os::raw::c_void, bytes) })?;
            unsafe { self.msm_mont_dev(offset, d as *const u64, coeffs.len()) }
        })();
        unsafe { ffi::zkp_dev_free(self.ctx.0, d) };
        r
    }
    pub fn handle(&self) -> u64 {
        self.handle
    }
}
impl<'c, P: AccelGroup> Drop for ResidentBases<'c, P>
where
    P::BaseField: AbiField,
{
    fn drop(&mut self) {
        unsafe { ffi::zkp_bases_free(self.ctx.0, self.handle) };
    }
}

/// True variable-base MSM (`zkp_msm_g*_var`): fresh bases every call, nothing stays resident — the semantics of
/// `Curve::vartime_multiscalar_mul(scalars, points)` for callers such as bulletproofs / spartan whose generators change.
/// `montgomery = true`: `scalars` are `Fr` elements; `false`: canonical `BigInteger256`.
pub fn msm_var<P: AccelGroup, S: AbiScalar>(ctx: &Ctx, points: &[GroupAffine<P>], scalars: &[S], montgomery: bool) -> Result<GroupProjective<P>, Error>
where
    P::BaseField: AbiField,
{
    let n = points.len().min(scalars.len());
    let (xy, inf) = marshal_points(&points[..n]);
    let l = <P::BaseField as AbiField>::LIMBS;
    let mut out = vec![0u64; 3 * l];
    let st = unsafe {
        if P::GROUP == 1 {
            ffi::zkp_msm_g1_var(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), scalars_ptr(scalars), n, montgomery as i32, out.as_mut_ptr())
        } else {
            ffi::zkp_msm_g2_var(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), scalars_ptr(scalars), n, montgomery as i32, out.as_mut_ptr())
        }
    };
    check(st)?;
    Ok(unmarshal_projective::<P>(&out))
}

// ------------------------------------------------------------------------------------------------ NTT
#[derive(Clone, Copy)]
pub enum NttOp {
    Fft = ffi::ZKP_NTT_FFT as isize,
    Ifft = ffi::ZKP_NTT_IFFT as isize,
    CosetFft = ffi::ZKP_NTT_COSET_FFT as isize,
    CosetIfft = ffi::ZKP_NTT_COSET_IFFT as isize,
}

/// drop-in for `domain.{fft,ifft,coset_fft,coset_ifft}_in_place(&mut v)`: `v.len()` must be the domain size (a power of
/// two; arkworks pads with zeros before calling — do the same), elements are `Fr` in memory layout.
pub fn ntt_in_place<F: AbiScalar>(ctx: &Ctx, curve: c_int, v: &mut [F], op: NttOp) -> Result<(), Error> {
    assert!(v.len().is_power_of_two());
    let log_n = v.len().trailing_zeros();
    check(unsafe { ffi::zkp_ntt(ctx.0, curve, scalars_mut_ptr(v), log_n, op as i32) })
}

// ------------------------------------------------------------------------------------------------ point codec
/// Error of the codec / checked-deserialize calls: the library reports the index of the first offending point.
#[derive(Debug)]
pub enum PointError {
    /// malformed compressed point (both flags set, x >= p, x^3 + b not a square) / point off the curve or outside the
    /// prime-order subgroup, at this index
    Invalid(usize),
    Accel(Error),
}

/// `n` ark-serialize 0.2 COMPRESSED points (what `Parameters::serialize` writes for a query) -> `Vec<GroupAffine<P>>`,
/// decompressed on the device (`zkp_g*_decompress`: one square root per lane).  `checked = true` adds ark's
/// `is_in_correct_subgroup_assuming_on_curve` for every point (`zkp_g*_subgroup_check`), i.e. the semantics of
/// `CanonicalDeserialize::deserialize`; `false` = `deserialize_unchecked`.
pub fn decompress_points<P: AccelGroup>(ctx: &Ctx, bytes: &[u8], checked: bool) -> Result<Vec<GroupAffine<P>>, PointError>
where
    P::BaseField: AbiField,
{
    let l = <P::BaseField as AbiField>::LIMBS;
    let pb = 8 * l; // one coordinate per point
    assert_eq!(bytes.len() % pb, 0, "a whole number of compressed points");
    let n = bytes.len() / pb;
    let mut xy = vec![0u64; n * 2 * l];
    let mut inf = vec![0u8; n];
    let mut bad: usize = 0;
    let st = unsafe {
        if P::GROUP == 1 {
            ffi::zkp_g1_decompress(ctx.0, P::CURVE, bytes.as_ptr(), n, xy.as_mut_ptr(), inf.as_mut_ptr(), &mut bad)
        } else {
            ffi::zkp_g2_decompress(ctx.0, P::CURVE, bytes.as_ptr(), n, xy.as_mut_ptr(), inf.as_mut_ptr(), &mut bad)
        }
    };
    if st == ffi::ZKP_ERR_INVALID_POINT {
        return Err(PointError::Invalid(bad));
    }
    check(st).map_err(PointError::Accel)?;
    if checked {
        subgroup_check_limbs::<P>(ctx, &xy, &inf)?;
    }
    Ok((0..n).map(|i| unmarshal_affine::<P>(&xy[i * 2 * l..(i + 1) * 2 * l], inf[i] != 0)).collect())
}

/// `&[GroupAffine<P>]` -> ark-serialize compressed bytes (`zkp_g*_compress`), e.g. the three points of a `Proof`.
pub fn compress_points<P: AccelGroup>(ctx: &Ctx, pts: &[GroupAffine<P>]) -> Result<Vec<u8>, Error>
where
    P::BaseField: AbiField,
{
    let l = <P::BaseField as AbiField>::LIMBS;
    let (xy, inf) = marshal_points(pts);
    let mut out = vec![0u8; pts.len() * 8 * l];
    let st = unsafe {
        if P::GROUP == 1 {
            ffi::zkp_g1_compress(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), pts.len(), out.as_mut_ptr())
        } else {
            ffi::zkp_g2_compress(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), pts.len(), out.as_mut_ptr())
        }
    };
    check(st)?;
    Ok(out)
}

fn subgroup_check_limbs<P: AccelGroup>(ctx: &Ctx, xy: &[u64], inf: &[u8]) -> Result<(), PointError>
where
    P::BaseField: AbiField,
{
    let mut bad: usize = 0;
    let n = inf.len();
    let st = unsafe {
        if P::GROUP == 1 {
            ffi::zkp_g1_subgroup_check(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), n, &mut bad)
        } else {
            ffi::zkp_g2_subgroup_check(ctx.0, P::CURVE, xy.as_ptr(), inf.as_ptr(), n, &mut bad)
        }
    };
    if st == ffi::ZKP_ERR_INVALID_POINT {
        return Err(PointError::Invalid(bad));
    }
    check(st).map_err(PointError::Accel)
}

/// ark-ec 0.2 `is_in_correct_subgroup_assuming_on_curve` (plus the curve equation) for a whole query at once.
pub fn subgroup_check<P: AccelGroup>(ctx: &Ctx, pts: &[GroupAffine<P>]) -> Result<(), PointError>
where
    P::BaseField: AbiField,
{
    let (xy, inf) = marshal_points(pts);
    subgroup_check_limbs::<P>(ctx, &xy, &inf)
}
