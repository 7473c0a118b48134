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

/// `zkp_groth16_pk`: queries as window tables in HBM, matrices as CSR.  Upload once per (key, circuit), prove many.
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
        let mut pk: *mut ffi::zkp_groth16_pk = ptr::null_mut();
        let st = unsafe {
            match shard {
                None => ffi::zkp_groth16_pk_upload(ctx.0, &desc, &mut pk),
                Some((rank, world)) => ffi::zkp_groth16_pk_upload_shard(ctx.0, &desc, rank, world, &mut pk),
            }
        };
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
