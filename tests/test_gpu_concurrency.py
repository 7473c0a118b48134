"""GPU: the threading contract of the boundary (include/zkp_accel.h "Conventions"; SURVEY §8(b): the reference may prove from several
rayon threads, groth16/src/prover.rs:164-190).  Two host threads, each with its OWN context on the same device, interleave
zkp_groth16_prove / zkp_msm_g1 / zkp_ntt calls: every result equals the sequential one.  Two threads entering the SAME context are
serialised by the per-context lock (ABI 0.5) and also get the sequential results."""
import random
import threading

import numpy as np
import pytest

from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context, NTT_COSET_FFT
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.params import get_curve

pytestmark = pytest.mark.gpu
TOXIC = dict(alpha=0x51, beta=0x52, gamma=0x53, delta=0x54, tau=0x5555555555555)


def _workload(curve, k, seed):
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples_for_domain(k), seed=seed)
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    rnd = random.Random(seed)
    rs = [(codec.fr_to_mont([rnd.randrange(c.r)], c)[0], codec.fr_to_mont([rnd.randrange(c.r)], c)[0]) for _ in range(6)]
    sc = codec.fr_canonical([rnd.randrange(c.r) for _ in range(1 << k)], c)
    vec = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(1 << k)], c).reshape(-1, 4)
    return inst, z, rs, sc, vec


def _run(ctx, pk, hb, z, rs, sc, vec, curve, out, barrier=None):
    """the interleaved call sequence of one prover thread"""
    try:
        if barrier:
            barrier.wait()
        res = []
        for i, (r, s) in enumerate(rs):
            res.append(("proof", pk.prove_raw(z, r, s)))
            xy, inf = hb.msm_affine(sc[: min(len(sc), hb.n) - 17 * i])       # affine: a Jacobian triple is not a canonical form
            res.append(("msm", np.concatenate([np.asarray(xy, dtype=np.uint64).ravel(), np.asarray([int(inf)], dtype=np.uint64)])))
            res.append(("ntt", ctx.ntt(curve, vec, NTT_COSET_FFT)))
        out.append(res)
    except BaseException as e:                        # surfaced by the main thread
        out.append(e)


def _same(a, b):
    assert len(a) == len(b)
    for (ka, va), (kb, vb) in zip(a, b):
        assert ka == kb
        if ka == "proof":
            assert np.array_equal(va[0], vb[0]) and np.array_equal(va[1], vb[1])
        else:
            assert np.array_equal(np.asarray(va), np.asarray(vb)), ka


@pytest.mark.parametrize("curve,k", [("bn254", 13), ("bls12_381", 11)])
def test_two_threads_two_contexts_equal_sequential(ctx, curve, k):
    inst, z, rs, sc, vec = _workload(curve, k, seed=0xC0 + k)
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    ctxs = [Context(0), Context(0)]
    pks = [groth16.ProvingKey(cx, params, inst) for cx in ctxs]
    hbs = [cx.upload_bases(get_curve(curve), 1, *params.h_query) for cx in ctxs]
    try:
        want = []
        _run(ctxs[0], pks[0], hbs[0], z, rs, sc, vec, curve, want)
        assert not isinstance(want[0], BaseException), want[0]
        for rep in range(2):
            outs, bar = [[], []], threading.Barrier(2)
            th = [threading.Thread(target=_run, args=(ctxs[t], pks[t], hbs[t], z, rs[::-1] if t else rs, sc, vec, curve, outs[t], bar))
                  for t in range(2)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join(timeout=600)
                assert not t_.is_alive()
            for t in range(2):
                assert not isinstance(outs[t][0], BaseException), outs[t][0]
            _same(outs[0][0], want[0])
            # thread 1 proved the (r, s) list in reverse order: compare proof by proof
            got1 = [v for kind, v in outs[1][0] if kind == "proof"][::-1]
            want_p = [v for kind, v in want[0] if kind == "proof"]
            for a, b in zip(got1, want_p):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        for hb in hbs:
            hb.free()
        for pk in pks:
            pk.free()
        for cx in ctxs:
            cx.close()


def test_two_threads_one_context_are_serialised(ctx):
    curve, k = "bn254", 12
    inst, z, rs, sc, vec = _workload(curve, k, seed=0xD1)
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    cx = Context(0)
    pk = groth16.ProvingKey(cx, params, inst)
    hb = cx.upload_bases(get_curve(curve), 1, *params.h_query)
    try:
        want = []
        _run(cx, pk, hb, z, rs, sc, vec, curve, want)
        outs, bar = [[], []], threading.Barrier(2)
        th = [threading.Thread(target=_run, args=(cx, pk, hb, z, rs, sc, vec, curve, outs[t], bar)) for t in range(2)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join(timeout=600)
            assert not t_.is_alive()
        for t in range(2):
            assert not isinstance(outs[t][0], BaseException), outs[t][0]
            _same(outs[t][0], want[0])
    finally:
        hb.free()
        pk.free()
        cx.close()
