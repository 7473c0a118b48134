//! Process-wide device context + resident-key cache for the `create_proof` seam (rust/patches/groth16-accel.diff).
//!
//! `zkp_groth16::create_proof(circuit, params, r, s)` (groth16/src/prover.rs:124-211) receives `&Parameters<E>` on every call and
//! has nowhere to keep device state, while the device key (window tables, evaluation-form transforms: seconds per 2^20
//! constraints) must be uploaded ONCE per `Parameters` and reused by every later proof.  This module is that "somewhere":
//!   * `ctx()`            one `Ctx` per process (device `ZKP_ACCEL_DEVICE`, default 0), created on first use.  Since ABI 0.5 every
//!                        entry point holds a per-context lock, so rayon threads proving through the shared context are
//!                        serialised, never interleaved; callers that want concurrent proofs create their own `Ctx` per thread
//!                        and call `groth16::DeviceProvingKey::upload` themselves.
//!   * `get_or_upload()`  the resident key of a `Parameters` value, identified by its ADDRESS and the address / length of its
//!                        `h_query` buffer (a `Parameters` that is moved or reloaded uploads again; `evict` drops an entry
//!                        before its `Parameters` is freed so that a later allocation at the same address cannot alias it).
//! SOURCE ONLY — never compiled (no Rust toolchain in the authoring image), like the rest of the crate.
use std::any::Any;
use std::collections::HashMap;
use std::sync::{Arc, Mutex, Once};

use crate::groth16::DeviceProvingKey;
use crate::{AbiField, AccelGroup, Ctx, Error};

static CTX_ONCE: Once = Once::new();
static mut CTX: Option<Result<Ctx, Error>> = None;

/// The process-wide context.  Panics never; a failed creation (no gfx950 device, library missing) is remembered and returned
/// by every `get_or_upload` as its error, so that the caller's arkworks path takes over (the patch's `Err(_) => {}` arm).
pub fn try_ctx() -> Result<&'static Ctx, Error> {
    CTX_ONCE.call_once(|| {
        let dev = std::env::var("ZKP_ACCEL_DEVICE").ok().and_then(|v| v.parse::<i32>().ok()).unwrap_or(0);
        // GPU_MAX_HW_QUEUES must be in the environment before the first HIP call
        if std::env::var_os("GPU_MAX_HW_QUEUES").is_none() {
            std::env::set_var("GPU_MAX_HW_QUEUES", "16");
        }
        unsafe { CTX = Some(Ctx::new(dev)) };
    });
    // written exactly once above, read-only afterwards
    match unsafe { CTX.as_ref() }.expect("initialised by call_once") {
        Ok(c) => Ok(c),
        Err(e) => Err(e.clone()),
    }
}

/// The context for `DeviceProvingKey::upload` inside a `get_or_upload` closure.  Only call it there: `get_or_upload` has
/// already checked that the context exists before it runs the closure.
pub fn ctx() -> &'static Ctx {
    try_ctx().expect("zkp-accel: get_or_upload checks the context before running its closure")
}

/// `Ctx` is shared between threads here; the library serialises the calls that enter one context (ABI 0.5).
struct Shared<T>(T);
unsafe impl<T> Send for Shared<T> {}
unsafe impl<T> Sync for Shared<T> {}

#[derive(Clone, Copy, PartialEq, Eq, Hash)]
struct KeyId {
    params: usize,
    size: usize,
}

fn table() -> &'static Mutex<HashMap<KeyId, Arc<dyn Any + Send + Sync>>> {
    static ONCE: Once = Once::new();
    static mut TABLE: Option<Mutex<HashMap<KeyId, Arc<dyn Any + Send + Sync>>>> = None;
    ONCE.call_once(|| unsafe { TABLE = Some(Mutex::new(HashMap::new())) });
    unsafe { TABLE.as_ref() }.expect("initialised by call_once")
}

fn id_of<P>(params: &P) -> KeyId {
    KeyId { params: params as *const P as usize, size: std::mem::size_of::<P>() }
}

/// A cached resident key: derefs to the `DeviceProvingKey` (so `.prove(..)` reads as in the patch).
pub struct CachedKey<G1: AccelGroup + 'static, G2: AccelGroup + 'static>(Arc<Shared<DeviceProvingKey<'static, G1, G2>>>)
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField;

impl<G1: AccelGroup + 'static, G2: AccelGroup + 'static> std::ops::Deref for CachedKey<G1, G2>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    type Target = DeviceProvingKey<'static, G1, G2>;
    fn deref(&self) -> &Self::Target {
        &(self.0).0
    }
}

/// The resident key of `params`: uploaded by `upload` on the first call, shared afterwards.  `upload` runs under the table's
/// lock (two threads proving with a fresh `Parameters` upload it once, the second waits).
pub fn get_or_upload<P, G1, G2, F>(params: &P, upload: F) -> Result<CachedKey<G1, G2>, Error>
where
    G1: AccelGroup + 'static,
    G2: AccelGroup + 'static,
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
    F: FnOnce() -> Result<DeviceProvingKey<'static, G1, G2>, Error>,
{
    try_ctx()?;
    let id = id_of(params);
    let mut t = table().lock().map_err(|_| Error::Device("zkp-accel key cache poisoned".into()))?;
    if let Some(entry) = t.get(&id) {
        if let Ok(k) = entry.clone().downcast::<Shared<DeviceProvingKey<'static, G1, G2>>>() {
            return Ok(CachedKey(k));
        }
    }
    let key = Arc::new(Shared(upload()?));
    t.insert(id, key.clone() as Arc<dyn Any + Send + Sync>);
    Ok(CachedKey(key))
}

/// Drop the resident key of `params` (call it before `params` is freed; the device memory goes when the last proof that
/// still holds the key returns).
pub fn evict<P>(params: &P) -> bool {
    table().lock().map(|mut t| t.remove(&id_of(params)).is_some()).unwrap_or(false)
}

/// Number of resident keys (diagnostics / tests).
pub fn len() -> usize {
    table().lock().map(|t| t.len()).unwrap_or(0)
}
