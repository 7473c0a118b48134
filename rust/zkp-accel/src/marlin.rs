//! Marlin on the device: `zkp_marlin::index` (device half) and `zkp_marlin::create_random_proof`
//! (marlin/src/lib.rs:69-181) behind `zkp_marlin_index_upload` / `zkp_marlin_index_commit` / `zkp_marlin_prove`.
//!
//! What stays in `zkp-marlin` (Rust): circuit synthesis, the index-manipulation half of `AHP::index`
//! (`make_matrices_square`, `balance_matrices`, per-row column sort: ahp/constraint_systems.rs:9-31,100-133), the
//! serialisation of the `IndexVerifierKey` (`to_bytes![ivk]`) and the zk randomness drawn from the caller's `zk_rng`.
//! What the library does in ONE call: prover_init, the three AHP rounds, `PC::commit` after each round, the
//! merlin/ChaCha20 `FiatShamirRng` transcript (fs_rng.rs), the 21 evaluations and `PC::batch_open`.
//! See rust/patches/marlin-accel.diff for the seam.  SOURCE ONLY — never compiled (no Rust toolchain here).
use std::ptr;

use ark_ec::models::short_weierstrass_jacobian::GroupAffine;

use crate::ffi;
use crate::groth16::Csr;
use crate::{check, unmarshal_affine, AbiField, AccelGroup, Ctx, Error, ResidentBases};

/// The zk randomness `create_random_proof` draws from `zk_rng`, in the order the reference draws it
/// (ahp/prover.rs:190-203 and `Rand::rand(hiding_bound = 1)` inside KZG10::commit, kzg10.rs:112-116).
pub struct MarlinRand<F: AbiField> {
    pub w: F,
    pub z_a: F,
    pub z_b: F,
    /// `DensePolynomial::rand(3|H| - 1)` coefficients (prover.rs:202-203)
    pub mask: Vec<F>,
    pub blind_w: [F; 2],
    pub blind_z_a: [F; 2],
    pub blind_z_b: [F; 2],
    pub blind_g_1: [F; 2],
    pub blind_shifted_g_1: [F; 2],
}

/// Device-resident index (`zkp_marlin_index`): the arithmetization of the three square matrices, computed on the device.
pub struct MarlinIndex<'c> {
    ctx: &'c Ctx,
    ix: *mut ffi::zkp_marlin_index,
}

/// What `zkp_marlin_prove` returns, decoded (commitments in oracle order w, z_a, z_b, mask | t, g_1, h_1 | g_2, h_2).
pub struct MarlinProofParts<G1: AccelGroup, F>
where
    G1::BaseField: AbiField,
{
    pub commitments: Vec<GroupAffine<G1>>,
    pub shifted: [GroupAffine<G1>; 2],
    pub evaluations: Vec<F>,
    pub opening_w: Vec<GroupAffine<G1>>,
    pub opening_rand_v: Vec<Option<F>>,
}

fn flat<F: AbiField>(v: &[F]) -> Vec<u64> {
    let mut out = vec![0u64; 4 * v.len()];
    for (i, x) in v.iter().enumerate() {
        x.write_limbs(&mut out[4 * i..4 * i + 4]);
    }
    out
}

impl<'c> MarlinIndex<'c> {
    /// `a`, `b`, `c`: the square, balanced matrices with ascending columns per row (what `AHP::index` holds after
    /// constraint_systems.rs:100-133); `n` rows == columns, `pad_aux` dummy witness variables appended by
    /// `make_matrices_square`, `num_inputs` formatted public inputs incl. the leading one.
    pub fn upload(ctx: &'c Ctx, curve: std::os::raw::c_int, a: &Csr, b: &Csr, c: &Csr, n: usize, num_inputs: usize, pad_aux: usize) -> Result<Self, Error> {
        let csr = |m: &Csr| ffi::zkp_csr { row_ptr: m.row_ptr.as_ptr(), col: m.col.as_ptr(), coeff: m.coeff.as_ptr() };
        let desc = ffi::zkp_marlin_index_desc { curve, num_inputs: num_inputs as u32, n: n as u32, pad_aux: pad_aux as u32, a: csr(a), b: csr(b), c: csr(c) };
        let mut ix: *mut ffi::zkp_marlin_index = ptr::null_mut();
        check(unsafe { ffi::zkp_marlin_index_upload(ctx.0, &desc, &mut ix) })?;
        Ok(MarlinIndex { ctx, ix })
    }

    /// (|X|, |H|, |K|, |B|, max_degree, num_non_zeros): `index.max_degree()` sizes the trimmed SRS (lib.rs:71-76).
    pub fn info(&self) -> Result<[u64; 6], Error> {
        let mut i = [0u64; 6];
        check(unsafe { ffi::zkp_marlin_index_info(self.ix, i.as_mut_ptr()) })?;
        Ok(i)
    }

    /// The 12 index commitments of `index()` (lib.rs:77-83): a_row, a_col, a_val, a_row_col, b_..., c_...
    pub fn commit<G1: AccelGroup>(&self, powers_of_g: &ResidentBases<'c, G1>) -> Result<Vec<GroupAffine<G1>>, Error>
    where
        G1::BaseField: AbiField,
    {
        let mut xy = vec![0u64; 12 * 12];
        let mut inf = [0u8; 12];
        check(unsafe { ffi::zkp_marlin_index_commit(self.ctx.0, self.ix, powers_of_g.handle(), xy.as_mut_ptr(), inf.as_mut_ptr()) })?;
        Ok((0..12).map(|k| unmarshal_affine::<G1>(&xy[12 * k..12 * k + 12], inf[k] != 0)).collect())
    }

    /// `create_random_proof` after synthesis (lib.rs:97-181).  `ivk_bytes` = `to_bytes![ipk.index_verifier_key]`,
    /// `x` = formatted public input (leading one included), `w` = witness without the squaring padding.
    pub fn prove<G1: AccelGroup, F: AbiField + Copy>(&self, powers_of_g: &ResidentBases<'c, G1>, powers_of_gamma_g: &ResidentBases<'c, G1>,
                                                     ivk_bytes: &[u8], x: &[F], w: &[F], rnd: &MarlinRand<F>) -> Result<MarlinProofParts<G1, F>, Error>
    where
        G1::BaseField: AbiField,
    {
        let (xl, wl, mask) = (flat(x), flat(w), flat(&rnd.mask));
        let one = |v: &F| flat(std::slice::from_ref(v));
        let (rw, rza, rzb) = (one(&rnd.w), one(&rnd.z_a), one(&rnd.z_b));
        let (bw, bza, bzb, bg1, bsg1) = (flat(&rnd.blind_w), flat(&rnd.blind_z_a), flat(&rnd.blind_z_b), flat(&rnd.blind_g_1), flat(&rnd.blind_shifted_g_1));
        let r = ffi::zkp_marlin_rand {
            w: rw.as_ptr(), z_a: rza.as_ptr(), z_b: rzb.as_ptr(), mask: mask.as_ptr(), mask_on_device: 0,
            blind_w: bw.as_ptr(), blind_z_a: bza.as_ptr(), blind_z_b: bzb.as_ptr(), blind_g_1: bg1.as_ptr(), blind_shifted_g_1: bsg1.as_ptr(),
        };
        let mut out: ffi::zkp_marlin_proof = unsafe { std::mem::zeroed() };
        check(unsafe {
            ffi::zkp_marlin_prove(self.ctx.0, self.ix, powers_of_g.handle(), powers_of_gamma_g.handle(), ivk_bytes.as_ptr(), ivk_bytes.len(),
                                  xl.as_ptr(), wl.as_ptr(), w.len(), &r, ptr::null(), &mut out)
        })?;
        let l = <G1::BaseField as AbiField>::LIMBS;
        let pt = |buf: &[u64], k: usize, inf: u8| unmarshal_affine::<G1>(&buf[12 * k..12 * k + 2 * l], inf != 0);
        let nopen = out.num_opening_proofs as usize;
        Ok(MarlinProofParts {
            commitments: (0..9).map(|k| pt(&out.comm, k, out.comm_inf[k])).collect(),
            shifted: [pt(&out.shifted, 0, out.shifted_inf[0]), pt(&out.shifted, 1, out.shifted_inf[1])],
            evaluations: (0..ffi::ZKP_MARLIN_NUM_EVALS).map(|k| F::read_limbs(&out.evaluations[4 * k..4 * k + 4])).collect(),
            opening_w: (0..nopen).map(|k| pt(&out.opening_w, k, out.opening_w_inf[k])).collect(),
            opening_rand_v: (0..nopen).map(|k| if out.opening_has_rand[k] != 0 { Some(F::read_limbs(&out.opening_rand_v[4 * k..4 * k + 4])) } else { None }).collect(),
        })
    }
}

impl<'c> Drop for MarlinIndex<'c> {
    fn drop(&mut self) {
        unsafe { ffi::zkp_marlin_index_free(self.ctx.0, self.ix) };
    }
}
