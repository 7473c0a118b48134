//! Process-wide device context + resident-key cache for the `create_proof` seam (rust/patches/groth16-accel.diff).
//!
//! `zkp_groth16::create_proof(circuit, params, r, s)` (groth16/src/prover.rs:124-211) receives `&Parameters<E>` on every call and
//! has nowhere to keep device state, while the device key (window tables, evaluation-form transforms: seconds per 2^20
//! constraints) must be uploaded ONCE per `Parameters` and reused by every later proof.  This module is that "somewhere":
//!   * `ctx()`            one `Ctx` per process (device `ZKP_ACCEL_DEVICE`, default 0), created on first use.  Since ABI 0.5 every
//!                        entry point holds a per-context lock, so rayon threads proving through the shared context are
//!                        serialised, never interleaved; callers that want concurrent proofs create their own `Ctx` per thread
//!                        and call `groth16::DeviceProvingKey::upload` themselves.
//!   * `get_or_upload()`  the resident key of a `Parameters` value.  The table is indexed by the value's ADDRESS, but an entry is
//!                        only a HIT when its `Fingerprint` matches too: the addresses and lengths of the `a_query` / `h_query`
//!                        / `l_query` buffers, the circuit shape (inputs, aux, non-zeros of A/B/C) and a 64-bit digest of the
//!                        key's CONTENT (alpha_g1, delta_g1, delta_g2 and the first / middle / last H and L points).  A
//!                        `Parameters` dropped and reloaded at the same address — a server swapping keys, `let params = load(..)`
//!                        in a loop — therefore misses and uploads again instead of proving with the old device key; `evict`
//!                        remains as the way to release device memory early, not as a correctness requirement.
//!   * locking            the table's mutex only guards the map.  Each entry is its own slot (`Mutex<Option<key>>`): the upload
//!                        runs under the SLOT's lock with the table unlocked, so threads proving with other, already resident
//!                        keys never wait for it; two threads arriving with the same fresh `Parameters` upload once.  A
//!                        panicking upload poisons nothing: poisoned locks are recovered with `into_inner` and the slot stays
//!                        empty (the next call retries).
//! `GPU_MAX_HW_QUEUES` is NOT set here (mutating the environment from an arbitrary thread races with `getenv` elsewhere): the
//! host sets it before the process starts, INTEGRATION.md section 4.
//! SOURCE ONLY — never compiled (no Rust toolchain in the authoring image), like the rest of the crate.
use std::any::Any;
use std::collections::HashMap;
use std::sync::{Arc, Mutex, MutexGuard, Once};

use crate::groth16::{DeviceProvingKey, KeyRef};
use crate::{AbiField, AccelGroup, Ctx, Error};

static CTX_ONCE: Once = Once::new();
static mut CTX: Option<Result<Ctx, Error>> = None;

/// The process-wide context.  Panics never; a failed creation (no gfx950 device, library missing) is remembered and returned
/// by every `get_or_upload` as its error, so that the caller's arkworks path takes over (the patch's `Err(_) => {}` arm).
pub fn try_ctx() -> Result<&'static Ctx, Error> {
    CTX_ONCE.call_once(|| {
        let dev = std::env::var("ZKP_ACCEL_DEVICE").ok().and_then(|v| v.parse::<i32>().ok()).unwrap_or(0);
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

/// What makes two `Parameters` + circuit pairs "the same key" for the cache (see the module doc).
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub struct Fingerprint {
    pub a_query: (usize, usize),
    pub h_query: (usize, usize),
    pub l_query: (usize, usize),
    pub num_inputs: usize,
    pub num_aux: usize,
    pub nnz: [usize; 3],
    pub digest: u64,
}

/// FNV-1a over the in-memory bytes of a value (arkworks field elements and affine points are plain old data).
fn fnv<T>(h: &mut u64, v: &T) {
    let bytes = unsafe { std::slice::from_raw_parts(v as *const T as *const u8, std::mem::size_of::<T>()) };
    for b in bytes {
        *h ^= *b as u64;
        *h = h.wrapping_mul(0x0000_0100_0000_01B3);
    }
}

impl Fingerprint {
    /// `nnz` = non-zeros of the A, B, C matrices the device key bakes in (`prover.at/bt/ct`).
    pub fn of<G1: AccelGroup, G2: AccelGroup>(key: &KeyRef<G1, G2>, num_inputs: usize, num_aux: usize, nnz: [usize; 3]) -> Self
    where
        G1::BaseField: AbiField,
        G2::BaseField: AbiField,
    {
        let mut h = 0xCBF2_9CE4_8422_2325u64;
        fnv(&mut h, key.alpha_g1);
        fnv(&mut h, key.delta_g1);
        fnv(&mut h, key.delta_g2);
        for q in [key.h_query, key.l_query, key.a_query] {
            if !q.is_empty() {
                fnv(&mut h, &q[0]);
                fnv(&mut h, &q[q.len() / 2]);
                fnv(&mut h, &q[q.len() - 1]);
            }
        }
        let span = |q: &[_]| (q.as_ptr() as usize, q.len());
        Fingerprint { a_query: span(key.a_query), h_query: span(key.h_query), l_query: span(key.l_query), num_inputs, num_aux, nnz, digest: h }
    }
}

struct Slot {
    print: Fingerprint,
    key: Mutex<Option<Arc<dyn Any + Send + Sync>>>,
}

type Table = Mutex<HashMap<usize, Arc<Slot>>>;

fn table() -> &'static Table {
    static ONCE: Once = Once::new();
    static mut TABLE: Option<Table> = None;
    ONCE.call_once(|| unsafe { TABLE = Some(Mutex::new(HashMap::new())) });
    unsafe { TABLE.as_ref() }.expect("initialised by call_once")
}

/// a panicking holder poisons a std mutex; the protected data (a map / an Option) is valid whatever happened
fn relock<T>(m: &Mutex<T>) -> MutexGuard<'_, T> {
    m.lock().unwrap_or_else(|p| p.into_inner())
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

/// The resident key of `params`: uploaded by `upload` on the first call (or when the entry at this address has another
/// `Fingerprint`: the old device key is dropped when its last proof returns), shared afterwards.
pub fn get_or_upload<P, G1, G2, F>(params: &P, print: Fingerprint, upload: F) -> Result<CachedKey<G1, G2>, Error>
where
    G1: AccelGroup + 'static,
    G2: AccelGroup + 'static,
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
    F: FnOnce() -> Result<DeviceProvingKey<'static, G1, G2>, Error>,
{
    try_ctx()?;
    let addr = params as *const P as usize;
    let slot = {
        let mut t = relock(table());
        match t.get(&addr) {
            Some(s) if s.print == print => s.clone(),
            _ => {
                let s = Arc::new(Slot { print, key: Mutex::new(None) });
                t.insert(addr, s.clone()); // replaces a stale entry of a `Parameters` that lived at this address before
                s
            }
        }
    }; // table unlocked: an upload below blocks only the callers of THIS key
    let mut k = relock(&slot.key);
    if let Some(entry) = k.as_ref() {
        if let Ok(hit) = entry.clone().downcast::<Shared<DeviceProvingKey<'static, G1, G2>>>() {
            return Ok(CachedKey(hit));
        }
    }
    let key = Arc::new(Shared(upload()?));
    *k = Some(key.clone() as Arc<dyn Any + Send + Sync>);
    Ok(CachedKey(key))
}

/// Drop the resident key of `params` (releases its device memory when the last proof that still holds it returns).
pub fn evict<P>(params: &P) -> bool {
    relock(table()).remove(&(params as *const P as usize)).is_some()
}

/// Number of resident keys (diagnostics / tests).
pub fn len() -> usize {
    relock(table()).len()
}
