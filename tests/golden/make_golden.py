#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.json from the CPU oracle (oracle/oracle.c).

The reference is CUDA-only and ships no integer-level vectors, so these are
not reference outputs: they freeze the oracle's outputs (whose correctness is pinned by
tests/test_oracle.py) so that later oracle edits cannot silently move the target.  Each entry
records the seeded inputs' parameters, an FNV-1a-64 digest of the full output and a few sampled
words.  Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

SAMPLES = [0, 1, 2, 3, 255, 256, 4095, 4096, 32767, 32768, 65534, 65535]


def entry(arr):
    flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
    idx = [i % flat.size for i in SAMPLES] + [flat.size - 1]
    return {"fnv1a64": f"{oracle.fnv1a64(flat):016x}", "words": int(flat.size),
            "samples": {str(i): str(int(flat[i])) for i in idx}}


def evk_uniform(o, seed):
    return [np.ascontiguousarray(np.stack([o.uniform_poly(list(range(o.size_QP)), seed + 10 * d),
                                           o.uniform_poly(list(range(o.size_QP)), seed + 10 * d + 1)]))
            for d in range(o.size_Q // o.size_P)]


def main():
    out = {"generator": "tests/golden/make_golden.py", "source": "oracle/oracle.c (CPU restatement; parity unpinned by reference fixtures)",
           "cases": {}}
    # config 1: N=2^16, one 60-bit limb, seed 1
    q0 = oracle.coeff_modulus_create(65536, [60, 60, 60, 50])[:1]
    o = oracle.Oracle(16, q0, 0)
    x = oracle.fill_uniform(65536, q0[0], 1)
    y = o.ntt_fwd(x, 0)
    out["cases"]["config1_ntt_n65536_q60"] = {"modulus": str(int(q0[0])), "seed": 1, "input": entry(x), "ntt": entry(y)}
    # config 2: uint64[2][36][65536], chain {60} + {50}x35, seeds 2 (poly 0) and 3 (poly 1) via uniform_poly
    mods = oracle.coeff_modulus_create(65536, [60] + [50] * 35)
    o = oracle.Oracle(16, mods, 0)
    polys = np.stack([o.uniform_poly(list(range(36)), 2), o.uniform_poly(list(range(36)), 3)])
    f = polys.copy()
    for p in range(2):
        o.nwt_forward_inplace(f[p], 36, 0)
    out["cases"]["config2_ntt_2x36x65536"] = {"moduli": [str(int(m)) for m in mods], "seeds": [2, 3], "input": entry(polys),
                                              "ntt": entry(f)}
    # small key-switch / rescale / galois / tensor cases at N=4096: alpha = 2 (beta = 3) and alpha = 1
    for size_p in (2, 1):
        mods = oracle.coeff_modulus_create(4096, [45] * 6 + [55] * size_p)
        o = oracle.Oracle(12, mods, size_p)
        evk = evk_uniform(o, 70)
        case = {"moduli": [str(int(m)) for m in mods], "size_P": size_p, "evk_seed": 70}
        for size_ql in (6, 5):
            c2 = o.uniform_poly(list(range(size_ql)), 6)
            ct = np.stack([o.uniform_poly(list(range(size_ql)), 7), o.uniform_poly(list(range(size_ql)), 8)])
            t = o.modup(size_ql, c2)
            cx = o.key_switch_inner_prod(size_ql, t, evk)
            ks = ct.copy()
            o.keyswitch_inplace(size_ql, ks, c2, evk)
            rs = o.divide_and_round_q_last_ntt(size_ql, ct, 2)
            case[f"size_Ql_{size_ql}"] = {"modup": entry(t), "inner_prod": entry(cx), "keyswitch": entry(ks),
                                          "rescale": entry(rs), "galois5": entry(o.apply_galois_ntt(ct[0], 5)),
                                          "tensor": entry(o.tensor_prod_2x2(ct, ks))}
        out["cases"][f"keyswitch_n4096_alpha{size_p}"] = case
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
