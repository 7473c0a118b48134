"""GPU: the HIP path (through the C ABI) against the committed golden vectors (tests/golden/golden.json)."""
import numpy as np
import pytest

from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.params import get_curve
from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
from oracle.pyref import groth16 as og
from tests.golden_util import GOLDEN, I, TOXIC, abi_params_from_oracle, golden_circuits, unpt
from tests.util import OC, jac_limbs_to_affine_oracle, jac_to_affine, to_abi_points

pytestmark = pytest.mark.gpu
CURVES = ["bn254", "bls12_381"]


@pytest.mark.parametrize("curve", CURVES)
def test_ntt_golden(ctx, curve):
    c = get_curve(curve)
    for e in GOLDEN["curves"][curve]["ntt"]:
        x = codec.fr_to_mont([I(v) for v in e["input"]], c)
        for op, name in enumerate(["fft", "ifft", "coset_fft", "coset_ifft"]):
            assert codec.fr_from_mont(ctx.ntt(c, x, op), c) == [I(v) for v in e[name]], (curve, name)


@pytest.mark.parametrize("curve", CURVES)
def test_msm_golden(ctx, curve):
    c = get_curve(curve)
    for e in GOLDEN["curves"][curve]["msm"]:
        g = e["group"]
        pts = [unpt(p, g) for p in e["bases"]]
        xy, inf = to_abi_points(curve, g, pts)
        bases = ctx.upload_bases(c, g, xy, inf)
        try:
            out = bases.msm(codec.fr_canonical([I(k) for k in e["scalars"]], c))
            assert jac_limbs_to_affine_oracle(curve, g, out) == unpt(e["result"], g)
            assert jac_to_affine(ctx, curve, g, out) == unpt(e["result"], g)
        finally:
            bases.free()


@pytest.mark.parametrize("curve", CURVES)
def test_groth16_golden(ctx, curve):
    c = get_curve(curve)
    for e in GOLDEN["curves"][curve]["groth16"]:
        ocirc, ocirc_setup, pcirc, pcirc_setup = golden_circuits(curve, e)
        opk = og.generate_parameters(OC[curve], ocirc_setup, **TOXIC, g1_k=e["g1_k"], g2_k=e["g2_k"])
        cs = ConstraintSystem(curve, True)
        pcirc.generate_constraints(cs)
        inst = R1csInstance.from_cs(cs)
        params = abi_params_from_oracle(curve, opk, inst.num_inputs, inst.num_aux, inst.num_constraints())
        pk = groth16.ProvingKey(ctx, params, inst)
        try:
            z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
            assert codec.fr_from_mont(pk.witness_map(z), c) == [I(v) for v in e["h"]]
            proof = groth16.create_proof(pk, pcirc, I(e["r"]), I(e["s"]))
            assert (proof.a, proof.b, proof.c) == (unpt(e["a"], 1), unpt(e["b"], 2), unpt(e["c"], 1))
        finally:
            pk.free()
