//! Device-resident Groth16 proving key + `create_proof` after synthesis (groth16/src/prover.rs:148-210).
//!
//! `zkp-groth16` keeps circuit synthesis (`ProvingAssignment`, prover.rs:16-95) and calls in here with what synthesis
//! produced: the three constraint matrices (fixed per circuit -> uploaded once with the key) and the assignment.
//! See rust/patches/groth16-accel.diff for the 30-line seam inside `zkp-groth16`.
use std::ptr;

use ark_ec::models::short_weierstrass_jacobian::GroupAffine;
use zkp_r1cs::Index;

use crate::ffi;
use crate::{check, marshal_points, unmarshal_affine, AbiField, AccelGroup, Ctx, Error};

/// CSR over constraints, columns index z = input_assignment ++ aux_assignment (prover.rs:17-19 flattened).
pub struct Csr {
    pub row_ptr: Vec<u32>,
    pub col: Vec<u32>,
    pub coeff: Vec<u64>,
}

/// `at` / `bt` / `ct` of `ProvingAssignment` (one `Vec<(Fr, Index)>` per constraint) -> CSR.  `F` is the scalar field as
/// it sits in memory (Fp256, Montgomery): coefficients are copied limb for limb.
pub fn csr_from_rows<F: AbiField>(rows: &[Vec<(F, Index)>], num_inputs: usize) -> Csr {
    let nnz: usize = rows.iter().map(|r| r.len()).sum();
    let mut out = Csr { row_ptr: Vec::with_capacity(rows.len() + 1), col: Vec::with_capacity(nnz), coeff: vec![0u64; nnz * 4] };
    out.row_ptr.push(0);
    let mut k = 0usize;
    for row in rows {
        for (coeff, index) in row {
            let c = match index {
                Index::Input(i) => *i,
                Index::Aux(i) => num_inputs + *i,
            };
            out.col.push(c as u32);
            coeff.write_limbs(&mut out.coeff[4 * k..4 * k + 4]);
            k += 1;
        }
        out.row_ptr.push(k as u32);
    }
    out
}

/// The `Parameters<E>` fields the prover reads (groth16/src/lib.rs:81-91), borrowed.
pub struct KeyRef<'a, G1: AccelGroup, G2: AccelGroup>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    pub alpha_g1: &'a GroupAffine<G1>,
    pub beta_g1: &'a GroupAffine<G1>,
    pub delta_g1: &'a GroupAffine<G1>,
    pub beta_g2: &'a GroupAffine<G2>,
    pub delta_g2: &'a GroupAffine<G2>,
    pub a_query: &'a [GroupAffine<G1>],
    pub b_g1_query: &'a [GroupAffine<G1>],
    pub b_g2_query: &'a [GroupAffine<G2>],
    pub h_query: &'a [GroupAffine<G1>],
    pub l_query: &'a [GroupAffine<G1>],
}

/// Marshals a borrowed key + matrices into a `zkp_groth16_pk_desc` (valid for the duration of `f`).
fn with_desc<G1: AccelGroup, G2: AccelGroup, T>(key: &KeyRef<G1, G2>, at: &Csr, bt: &Csr, ct: &Csr, num_inputs: usize, num_aux: usize,
                                                 f: impl FnOnce(&ffi::zkp_groth16_pk_desc) -> T) -> T
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    let single = |p: &GroupAffine<G1>| marshal_points(std::slice::from_ref(p)).0;
    let single2 = |p: &GroupAffine<G2>| marshal_points(std::slice::from_ref(p)).0;
    let (alpha, beta1, delta1) = (single(key.alpha_g1), single(key.beta_g1), single(key.delta_g1));
    let (beta2, delta2) = (single2(key.beta_g2), single2(key.delta_g2));
    let (a, a_inf) = marshal_points(key.a_query);
    let (b1, b1_inf) = marshal_points(key.b_g1_query);
    let (b2, b2_inf) = marshal_points(key.b_g2_query);
    let (h, h_inf) = marshal_points(key.h_query);
    let (l, l_inf) = marshal_points(key.l_query);
    let csr = |m: &Csr| ffi::zkp_csr { row_ptr: m.row_ptr.as_ptr(), col: m.col.as_ptr(), coeff: m.coeff.as_ptr() };
    let desc = ffi::zkp_groth16_pk_desc {
        curve: G1::CURVE,
        num_inputs: num_inputs as u32,
        num_aux: num_aux as u32,
        num_constraints: (at.row_ptr.len() - 1) as u32,
        at: csr(at),
        bt: csr(bt),
        ct: csr(ct),
        alpha_g1: alpha.as_ptr(),
        beta_g1: beta1.as_ptr(),
        delta_g1: delta1.as_ptr(),
        beta_g2: beta2.as_ptr(),
        delta_g2: delta2.as_ptr(),
        a_query: a.as_ptr(),
        a_inf: a_inf.as_ptr(),
        a_len: key.a_query.len(),
        b_g1_query: b1.as_ptr(),
        b_g1_inf: b1_inf.as_ptr(),
        b_g1_len: key.b_g1_query.len(),
        b_g2_query: b2.as_ptr(),
        b_g2_inf: b2_inf.as_ptr(),
        b_g2_len: key.b_g2_query.len(),
        h_query: h.as_ptr(),
        h_inf: h_inf.as_ptr(),
        h_len: key.h_query.len(),
        l_query: l.as_ptr(),
        l_inf: l_inf.as_ptr(),
        l_len: key.l_query.len(),
    };
    f(&desc)
}

/// `zkp_groth16_pk`: queries as window tables in HBM, matrices as CSR.  Upload once per (key, circuit), prove many.
/// Since ABI 0.4 the upload also puts the key into *evaluation form* on the device (h_query transformed over the coset of
/// `r1cs_to_qap.rs:164-169`, the C matrix folded into l_query): a proof then runs 4 of the 7 transforms of `witness_map`
/// and no `C z`, and returns the same `Proof` for every assignment.  The borrowed `Parameters` are not modified; the
/// upload takes ~0.7 s longer per 2^20 constraints.  `ZKP_H_LAGRANGE=0` / `ZKP_C_FOLD=0` in the environment keep the key as given.
pub struct DeviceProvingKey<'c, G1: AccelGroup, G2: AccelGroup>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    ctx: &'c Ctx,
    pk: *mut ffi::zkp_groth16_pk,
    nz: usize,
    _p: std::marker::PhantomData<(G1, G2)>,
}

impl<'c, G1: AccelGroup, G2: AccelGroup> DeviceProvingKey<'c, G1, G2>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    /// `shard = Some((rank, world))`: keep only this rank's 1/world of every query resident (multi-GPU, one process per
    /// GPU; see `prove_partials` / `fold_assemble`).
    pub fn upload(ctx: &'c Ctx, key: &KeyRef<G1, G2>, at: &Csr, bt: &Csr, ct: &Csr, num_inputs: usize, num_aux: usize,
                  shard: Option<(i32, i32)>) -> Result<Self, Error> {
        let mut pk: *mut ffi::zkp_groth16_pk = ptr::null_mut();
        let st = with_desc(key, at, bt, ct, num_inputs, num_aux, |desc| unsafe {
            match shard {
                None => ffi::zkp_groth16_pk_upload(ctx.0, desc, &mut pk),
                Some((rank, world)) => ffi::zkp_groth16_pk_upload_shard(ctx.0, desc, rank, world, &mut pk),
            }
        });
        check(st)?;
        Ok(DeviceProvingKey { ctx, pk, nz: num_inputs + num_aux, _p: std::marker::PhantomData })
    }

    /// `zkp_groth16_pk_upload_ex(ZKP_PK_KEEP_FORM)`: the key stays as `Parameters` holds it — no evaluation-form transforms at upload
    /// (~0.65 s less per 2^20 constraints), all seven transforms of `witness_map` per proof.  For callers that prove once or a few
    /// times per key (the reference CLI, cli/src/zkp_prove.rs); the transforms of `upload` pay back after ~2000 proofs.
    pub fn upload_as_given(ctx: &'c Ctx, key: &KeyRef<G1, G2>, at: &Csr, bt: &Csr, ct: &Csr, num_inputs: usize, num_aux: usize) -> Result<Self, Error> {
        let mut pk: *mut ffi::zkp_groth16_pk = ptr::null_mut();
        let st = with_desc(key, at, bt, ct, num_inputs, num_aux, |desc| unsafe {
            ffi::zkp_groth16_pk_upload_ex(ctx.0, desc, ffi::ZKP_PK_KEEP_FORM, &mut pk)
        });
        check(st)?;
        Ok(DeviceProvingKey { ctx, pk, nz: num_inputs + num_aux, _p: std::marker::PhantomData })
    }

    /// `create_proof(params, circuit, r, s)` minus synthesis: `input_assignment` (with the leading one, prover.rs:143)
    /// and `aux_assignment` as `Fr` elements, r / s as `Fr`.  Returns (A, B, C) affine.
    pub fn prove<F: AbiField>(&self, input_assignment: &[F], aux_assignment: &[F], r: &F, s: &F)
                              -> Result<(GroupAffine<G1>, GroupAffine<G2>, GroupAffine<G1>), Error> {
        assert_eq!(input_assignment.len() + aux_assignment.len(), self.nz);
        let mut z = vec![0u64; self.nz * 4];
        for (i, v) in input_assignment.iter().chain(aux_assignment.iter()).enumerate() {
            v.write_limbs(&mut z[4 * i..4 * i + 4]);
        }
        let (mut rl, mut sl) = ([0u64; 4], [0u64; 4]);
        r.write_limbs(&mut rl);
        s.write_limbs(&mut sl);
        let l1 = <G1::BaseField as AbiField>::LIMBS;
        let l2 = <G2::BaseField as AbiField>::LIMBS;
        let mut proof = vec![0u64; 4 * l1 + 2 * l2]; // A (G1 affine) | B (G2 affine) | C (G1 affine)
        let mut inf = [0u8; 3];
        check(unsafe { ffi::zkp_groth16_prove(self.ctx.0, self.pk, z.as_ptr(), rl.as_ptr(), sl.as_ptr(), proof.as_mut_ptr(), inf.as_mut_ptr()) })?;
        Ok((
            unmarshal_affine::<G1>(&proof[..2 * l1], inf[0] != 0),
            unmarshal_affine::<G2>(&proof[2 * l1..2 * l1 + 2 * l2], inf[1] != 0),
            unmarshal_affine::<G1>(&proof[2 * l1 + 2 * l2..], inf[2] != 0),
        ))
    }
}

impl<'c, G1: AccelGroup, G2: AccelGroup> Drop for DeviceProvingKey<'c, G1, G2>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    fn drop(&mut self) {
        unsafe { ffi::zkp_groth16_pk_free(self.ctx.0, self.pk) };
    }
}

/// How a multi-device key is laid out (`zkp_multi_mode`).
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum MultiMode {
    /// every query split by index over the devices: ONE proof uses all of them (BASELINE configs[4]; `prove`)
    Shard = ffi::ZKP_MULTI_SHARD as isize,
    /// the whole key on every device: independent proofs round-robin (`prove_batch`)
    Replicate = ffi::ZKP_MULTI_REPLICATE as isize,
}

/// `zkp_groth16_pk_multi` on a `MultiCtx`: the single-process multi-GPU prover.  The exchange step (partial sums gathered
/// over xGMI, folded and assembled on rank 0) and, from three devices on, the task-split witness map live inside the
/// library: the Rust caller of `create_proof` sees one call.
pub struct MultiDeviceProvingKey<'c, G1: AccelGroup, G2: AccelGroup>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    ctx: &'c crate::MultiCtx,
    pk: *mut ffi::zkp_groth16_pk_multi,
    nz: usize,
    mode: MultiMode,
    _p: std::marker::PhantomData<(G1, G2)>,
}

impl<'c, G1: AccelGroup, G2: AccelGroup> MultiDeviceProvingKey<'c, G1, G2>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    pub fn upload(ctx: &'c crate::MultiCtx, key: &KeyRef<G1, G2>, at: &Csr, bt: &Csr, ct: &Csr, num_inputs: usize, num_aux: usize,
                  mode: MultiMode) -> Result<Self, Error> {
        let mut pk: *mut ffi::zkp_groth16_pk_multi = ptr::null_mut();
        let st = with_desc(key, at, bt, ct, num_inputs, num_aux, |desc| unsafe {
            ffi::zkp_groth16_pk_upload_multi(ctx.root.0, desc, mode as i32, &mut pk)
        });
        check(st)?;
        Ok(MultiDeviceProvingKey { ctx, pk, nz: num_inputs + num_aux, mode, _p: std::marker::PhantomData })
    }

    fn flatten<F: AbiField>(&self, input_assignment: &[F], aux_assignment: &[F]) -> Vec<u64> {
        assert_eq!(input_assignment.len() + aux_assignment.len(), self.nz);
        let mut z = vec![0u64; self.nz * 4];
        for (i, v) in input_assignment.iter().chain(aux_assignment.iter()).enumerate() {
            v.write_limbs(&mut z[4 * i..4 * i + 4]);
        }
        z
    }
    fn decode(proof: &[u64], inf: &[u8]) -> (GroupAffine<G1>, GroupAffine<G2>, GroupAffine<G1>) {
        let l1 = <G1::BaseField as AbiField>::LIMBS;
        let l2 = <G2::BaseField as AbiField>::LIMBS;
        (
            unmarshal_affine::<G1>(&proof[..2 * l1], inf[0] != 0),
            unmarshal_affine::<G2>(&proof[2 * l1..2 * l1 + 2 * l2], inf[1] != 0),
            unmarshal_affine::<G1>(&proof[2 * l1 + 2 * l2..], inf[2] != 0),
        )
    }

    /// ONE proof over all devices of the context (`MultiMode::Shard`): `zkp_groth16_prove_multi`.
    pub fn prove<F: AbiField>(&self, input_assignment: &[F], aux_assignment: &[F], r: &F, s: &F)
                              -> Result<(GroupAffine<G1>, GroupAffine<G2>, GroupAffine<G1>), Error> {
        assert!(self.mode == MultiMode::Shard);
        let z = self.flatten(input_assignment, aux_assignment);
        let (mut rl, mut sl) = ([0u64; 4], [0u64; 4]);
        r.write_limbs(&mut rl);
        s.write_limbs(&mut sl);
        let words = 4 * <G1::BaseField as AbiField>::LIMBS + 2 * <G2::BaseField as AbiField>::LIMBS;
        let mut proof = vec![0u64; words];
        let mut inf = [0u8; 3];
        let zp = [z.as_ptr()];
        check(unsafe { ffi::zkp_groth16_prove_multi(self.ctx.root.0, self.pk, zp.as_ptr(), 0, rl.as_ptr(), sl.as_ptr(), proof.as_mut_ptr(), inf.as_mut_ptr()) })?;
        Ok(Self::decode(&proof, &inf))
    }

    /// `witnesses.len()` independent proofs, proof i on device i % n (`MultiMode::Replicate`): `zkp_groth16_prove_batch_multi`.
    /// `witnesses[i]` = (input_assignment, aux_assignment) of proof i; `rs[i]` = its (r, s).
    pub fn prove_batch<F: AbiField>(&self, witnesses: &[(&[F], &[F])], rs: &[(F, F)])
                                    -> Result<Vec<(GroupAffine<G1>, GroupAffine<G2>, GroupAffine<G1>)>, Error> {
        assert!(self.mode == MultiMode::Replicate && witnesses.len() == rs.len());
        let n = witnesses.len();
        let zs: Vec<Vec<u64>> = witnesses.iter().map(|(i, a)| self.flatten(i, a)).collect();
        let zp: Vec<*const u64> = zs.iter().map(|z| z.as_ptr()).collect();
        let (mut r, mut s) = (vec![0u64; 4 * n], vec![0u64; 4 * n]);
        for (i, (ri, si)) in rs.iter().enumerate() {
            ri.write_limbs(&mut r[4 * i..4 * i + 4]);
            si.write_limbs(&mut s[4 * i..4 * i + 4]);
        }
        let words = 4 * <G1::BaseField as AbiField>::LIMBS + 2 * <G2::BaseField as AbiField>::LIMBS;
        let mut proofs = vec![0u64; words * n];
        let mut inf = vec![0u8; 3 * n];
        check(unsafe {
            ffi::zkp_groth16_prove_batch_multi(self.ctx.root.0, self.pk, n, zp.as_ptr(), 0, r.as_ptr(), s.as_ptr(), proofs.as_mut_ptr(), inf.as_mut_ptr())
        })?;
        Ok((0..n).map(|i| Self::decode(&proofs[i * words..(i + 1) * words], &inf[3 * i..3 * i + 3])).collect())
    }
}

impl<'c, G1: AccelGroup, G2: AccelGroup> Drop for MultiDeviceProvingKey<'c, G1, G2>
where
    G1::BaseField: AbiField,
    G2::BaseField: AbiField,
{
    fn drop(&mut self) {
        unsafe { ffi::zkp_groth16_pk_multi_free(self.ctx.root.0, self.pk) };
    }
}
