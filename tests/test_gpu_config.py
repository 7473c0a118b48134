"""zkp_ctx_config (ABI 0.6): the prover's switches are per CONTEXT — `zkp_ctx_create_ex(ctx**, device, const zkp_ctx_config*)`,
environment variables only supply the defaults (SURVEY §5 "Config / flags": env vars / struct; §8(b) "no global state besides
zkp_ctx").  Two contexts of one process with different lane counts, key forms and window sizes prove the SAME proof, equal to the
oracle's; the resolved configuration can be read back; out-of-range fields are rejected."""
import ctypes as C

import numpy as np
import pytest

from ckb_zkp_amd import _lib, codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from oracle import cpu_oracle

pytestmark = pytest.mark.gpu
TOXIC = dict(alpha=0x51, beta=0x52, gamma=0x53, delta=0x54, tau=0x123456789)


def test_two_contexts_with_different_configurations_prove_the_same_proofs(ctx, monkeypatch):
    monkeypatch.setenv("ZKP_LANES", "7")                       # a default for contexts created from here on, not a global
    inst = mimc_chain_instance("bn254", samples_for_domain(13))
    params = groth16.generate_parameters(ctx, "bn254", inst, **TOXIC)
    c = params.curve
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    rng = np.random.default_rng(5)
    n = 6
    rs = codec.fr_to_mont([int.from_bytes(rng.bytes(40), "little") % c.r for _ in range(n)], c)
    ss = codec.fr_to_mont([int.from_bytes(rng.bytes(40), "little") % c.r for _ in range(n)], c)
    configs = [None,
               dict(lanes=2),
               dict(lanes=8, h_evaluation_form=False, host_affine=False),
               dict(lanes=3, c_fold=False, msm_window_bits=11, msm_window_bits_g2=9, table_budget_gb=0.25)]
    ctxs = [Context(ctx.device, cfg) for cfg in configs]
    try:
        got = [x.config() for x in ctxs]
        assert got[0]["lanes"] == 7 and got[1]["lanes"] == 2 and got[2]["lanes"] == 8 and got[3]["lanes"] == 3     # env = default only
        assert got[0]["h_evaluation_form"] == _lib.ZKP_ON and got[2]["h_evaluation_form"] == _lib.ZKP_OFF
        assert got[2]["host_affine"] == _lib.ZKP_OFF and got[3]["c_fold"] == _lib.ZKP_OFF
        assert got[3]["msm_window_bits"] == 11 and got[3]["msm_window_bits_g2"] == 9 and got[3]["table_budget_gb"] == 0.25
        assert all(g["struct_size"] == C.sizeof(_lib.CtxConfig) for g in got)
        pks = [groth16.ProvingKey(x, params, inst) for x in ctxs]
        plans = [pk.table_plan() for pk in pks]
        assert plans[0]["h_evaluation_form"] and not plans[2]["h_evaluation_form"]          # the key form followed the context
        assert plans[0]["c_folded_into_l"] and not plans[3]["c_folded_into_l"]
        outs = []
        for x, pk in zip(ctxs, pks):
            zd = x.to_device(z)
            outs.append(pk.prove_batch_raw([zd] * n, rs, ss))
            x.dev_free(zd)
        for o, i in outs[1:]:
            assert np.array_equal(o, outs[0][0]) and np.array_equal(i, outs[0][1])
        o_out, o_inf, _ = cpu_oracle.groth16_prove(params, inst, z, rs[0], ss[0], threads=4)
        assert np.array_equal(outs[0][0][0], o_out) and np.array_equal(outs[0][1][0], o_inf)
        for pk in pks:
            pk.free()
    finally:
        for x in ctxs:
            x.close()


def test_config_validation_and_forward_compatible_struct_size(ctx):
    lib = ctx.lib
    for bad in (dict(lanes=9), dict(lanes=-1), dict(msm_window_bits=1), dict(msm_window_bits_g2=23), dict(msm_chunk_points=100),
                dict(h_evaluation_form=3), dict(multi_exchange=5), dict(table_budget_gb=-1.0), dict(multi_exchange_timeout_ms=-5)):
        with pytest.raises(_lib.ZkpError) as e:
            Context(ctx.device, bad)
        assert e.value.status == -1, bad
    # struct_size 0 is rejected; a SHORTER struct (an older caller) is accepted and the missing tail is all defaults
    h = C.c_void_p()
    cfg = _lib.CtxConfig()
    assert lib.zkp_ctx_create_ex(C.byref(h), ctx.device, C.byref(cfg)) == -1
    cfg.struct_size, cfg.lanes, cfg.multi_witness_split = _lib.CtxConfig.msm_batch_lanes.offset, 4, 1      # only struct_size + lanes are "there"
    assert lib.zkp_ctx_create_ex(C.byref(h), ctx.device, C.byref(cfg)) == 0
    out = _lib.CtxConfig()
    assert lib.zkp_ctx_get_config(h, C.byref(out)) == 0
    assert out.lanes == 4 and out.multi_witness_split == 0                                    # the field beyond struct_size was not read
    assert lib.zkp_ctx_destroy(h) == 0
    with pytest.raises(KeyError):
        _lib.make_config(dict(no_such_field=1))
