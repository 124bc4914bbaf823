"""The committed golden fixtures (tests/golden/golden_v1.json, generated from the oracle by
tests/golden/make_golden.py) still match what the oracle computes (CPU), and the HIP path
reproduces them at BASELINE.json's full sizes (GPU)."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.json")))["cases"]


def _check(orc, arr, ent):
    flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
    assert flat.size == ent["words"]
    for i, v in ent["samples"].items():
        assert int(flat[int(i)]) == int(v), f"word {i}"
    assert f"{orc.fnv1a64(flat):016x}" == ent["fnv1a64"]


def test_oracle_reproduces_config1(orc):
    g = GOLD["config1_ntt_n65536_q60"]
    q = np.array([int(g["modulus"])], dtype=np.uint64)
    o = orc.Oracle(16, q, 0)
    x = orc.fill_uniform(65536, int(q[0]), g["seed"])
    _check(orc, x, g["input"])
    _check(orc, o.ntt_fwd(x, 0), g["ntt"])


@pytest.mark.parametrize("size_p", [2, 1])
def test_oracle_reproduces_keyswitch_cases(orc, size_p):
    g = GOLD[f"keyswitch_n4096_alpha{size_p}"]
    mods = np.array([int(m) for m in g["moduli"]], dtype=np.uint64)
    assert (mods == orc.coeff_modulus_create(4096, [45] * 6 + [55] * size_p)).all()
    o = orc.Oracle(12, mods, size_p)
    evk = [np.ascontiguousarray(np.stack([o.uniform_poly(list(range(o.size_QP)), 70 + 10 * d),
                                           o.uniform_poly(list(range(o.size_QP)), 71 + 10 * d)]))
           for d in range(o.size_Q // size_p)]
    for size_ql in (6, 5):
        c = g[f"size_Ql_{size_ql}"]
        c2 = o.uniform_poly(list(range(size_ql)), 6)
        ct = np.stack([o.uniform_poly(list(range(size_ql)), 7), o.uniform_poly(list(range(size_ql)), 8)])
        t = o.modup(size_ql, c2)
        _check(orc, t, c["modup"])
        _check(orc, o.key_switch_inner_prod(size_ql, t, evk), c["inner_prod"])
        ks = ct.copy()
        o.keyswitch_inplace(size_ql, ks, c2, evk)
        _check(orc, ks, c["keyswitch"])
        _check(orc, o.divide_and_round_q_last_ntt(size_ql, ct, 2), c["rescale"])
        _check(orc, o.apply_galois_ntt(ct[0], 5), c["galois5"])
        _check(orc, o.tensor_prod_2x2(ct, ks), c["tensor"])


@pytest.mark.gpu
def test_gpu_config1_and_config2_match_golden(gpu, pha, orc):
    g = GOLD["config1_ntt_n65536_q60"]
    q = np.array([int(g["modulus"])], dtype=np.uint64)
    ctx = pha.PhantomContext(16, q, 0)
    x = orc.fill_uniform(65536, int(q[0]), g["seed"])
    d = pha.to_device(x[None])
    ctx.nwt_2d_radix8_forward_inplace(d, 1, 0)
    _check(orc, pha.to_host(d), g["ntt"])
    ctx.nwt_2d_radix8_backward_inplace(d, 1, 0)
    assert (pha.to_host(d)[0] == x).all()
    ctx.close()
    g = GOLD["config2_ntt_2x36x65536"]
    mods = np.array([int(m) for m in g["moduli"]], dtype=np.uint64)
    ctx, o = pha.PhantomContext(16, mods, 0), orc.Oracle(16, mods, 0)
    polys = np.stack([o.uniform_poly(list(range(36)), s) for s in g["seeds"]])
    _check(orc, polys, g["input"])
    d = pha.to_device(polys)
    ctx.ntt_forward_batched(d, 2, 36, 36, 0)
    _check(orc, pha.to_host(d), g["ntt"])
    ctx.ntt_backward_batched(d, 2, 36, 36, 0)
    assert (pha.to_host(d) == polys).all()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size_p", [2, 1])
def test_gpu_keyswitch_cases_match_golden(gpu, pha, orc, size_p):
    g = GOLD[f"keyswitch_n4096_alpha{size_p}"]
    mods = np.array([int(m) for m in g["moduli"]], dtype=np.uint64)
    ctx, o = pha.PhantomContext(12, mods, size_p), orc.Oracle(12, mods, size_p)
    evk_h = [np.ascontiguousarray(np.stack([o.uniform_poly(list(range(o.size_QP)), 70 + 10 * d),
                                             o.uniform_poly(list(range(o.size_QP)), 71 + 10 * d)]))
             for d in range(o.size_Q // size_p)]
    evk = pha.EvalKey([pha.to_device(h) for h in evk_h])
    for size_ql in (6, 5):
        chain = 6 - size_ql + 1
        c = g[f"size_Ql_{size_ql}"]
        c2 = o.uniform_poly(list(range(size_ql)), 6)
        ct = np.stack([o.uniform_poly(list(range(size_ql)), 7), o.uniform_poly(list(range(size_ql)), 8)])
        beta = ctx.beta(chain)
        d_t = pha.to_device(np.zeros((beta, size_ql + size_p, 4096), dtype=np.uint64))
        ctx.modup(chain, d_t, pha.to_device(c2))
        _check(orc, pha.to_host(d_t), c["modup"])
        d_cx = pha.to_device(np.zeros((2, size_ql + size_p, 4096), dtype=np.uint64))
        ctx.key_switch_inner_prod(chain, d_cx, d_t, evk)
        _check(orc, pha.to_host(d_cx), c["inner_prod"])
        d_ks = pha.to_device(ct)
        ctx.keyswitch_inplace(chain, d_ks, pha.to_device(c2), evk)
        _check(orc, pha.to_host(d_ks), c["keyswitch"])
        d_rs = pha.to_device(np.zeros((2, size_ql - 1, 4096), dtype=np.uint64))
        ctx.rescale_to_next(chain, pha.to_device(ct), 2, d_rs)
        _check(orc, pha.to_host(d_rs), c["rescale"])
        d_g = pha.to_device(np.zeros((size_ql, 4096), dtype=np.uint64))
        ctx.apply_galois_ntt(pha.to_device(ct[0]), size_ql, 5, d_g)
        _check(orc, pha.to_host(d_g), c["galois5"])
        d_tp = pha.to_device(np.zeros((3, size_ql, 4096), dtype=np.uint64))
        ctx.tensor_prod_2x2(pha.to_device(ct), d_ks, size_ql, d_tp)
        _check(orc, pha.to_host(d_tp), c["tensor"])
    ctx.close()
