"""Loader for tests/golden/golden.json + builders of ABI-layout keys from oracle (pyref) keys."""
import json
from pathlib import Path

import numpy as np

from ckb_zkp_amd import codec
from ckb_zkp_amd.groth16 import Parameters
from ckb_zkp_amd.params import get_curve
from oracle.pyref import groth16 as og
from tests.util import OC

GOLDEN = json.loads((Path(__file__).parent / "golden" / "golden.json").read_text())
TOXIC = {k: int(v, 16) for k, v in GOLDEN["toxic"].items()}
I = lambda h: int(h, 16)


def unpt(p, g):
    if p is None:
        return None
    if g == 1:
        return (I(p[0]), I(p[1]))
    return ((I(p[0][0]), I(p[0][1])), (I(p[1][0]), I(p[1][1])))


def abi_params_from_oracle(curve, opk: og.Parameters, num_inputs, num_aux, num_constraints) -> Parameters:
    """oracle/pyref Parameters (python ints) -> product Parameters (Montgomery limbs), no device involved."""
    c = get_curve(curve)
    g1 = lambda pts: codec.g1_to_mont(pts, c)
    g2 = lambda pts: codec.g2_to_mont(pts, c)
    return Parameters(curve=c, num_inputs=num_inputs, num_aux=num_aux, num_constraints=num_constraints,
                      alpha_g1=g1([opk.alpha_g1])[0][0], beta_g1=g1([opk.beta_g1])[0][0],
                      beta_g2=g2([opk.beta_g2])[0][0], gamma_g2=g2([opk.gamma_g2])[0][0],
                      delta_g1=g1([opk.delta_g1])[0][0], delta_g2=g2([opk.delta_g2])[0][0],
                      gamma_abc_g1=g1(opk.gamma_abc_g1), a_query=g1(opk.a_query), b_g1_query=g1(opk.b_g1_query),
                      b_g2_query=g2(opk.b_g2_query), h_query=g1(opk.h_query), l_query=g1(opk.l_query),
                      toxic=dict(opk.trapdoor))


def golden_circuits(curve, entry):
    """-> (oracle circuit w/ witness, oracle circuit w/o witness, product circuit w/ witness, product w/o)"""
    from ckb_zkp_amd.circuits import Mini, MimcChain
    if entry["circuit"] == "mini":
        a = (entry["x"], entry["y"], entry["z"], entry["num"])
        return og.MiniCircuit(*a), og.MiniCircuit(num=entry["num"]), Mini(*a), Mini(num=entry["num"])
    consts = [I(v) for v in entry["constants"]]
    pre = [(I(a), I(b)) for a, b in entry["preimages"]]
    nopre = [(None, None)] * len(pre)
    return (og.MimcChain(OC[curve], consts, pre), og.MimcChain(OC[curve], consts, pre),
            MimcChain(curve, consts, pre), MimcChain(curve, consts, nopre))
