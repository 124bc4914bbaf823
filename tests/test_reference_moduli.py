"""The one piece of integer data the reference itself holds for the hot path's table generation: SEAL's default
moduli (`/root/reference/src/util/globals.cu:66-109`, copied as data into tests/golden/seal_default_moduli.json).  For
N = 4096, 8192, 16384 and 32768 those lists are exactly what the reference's `get_primes` walk
(`src/util/numth.cu:209-235`) finds, largest first, and `CoeffModulus::Create` (`src/modulus.cu:79-110`) hands out the
primes of one bit size smallest first.  The oracle's restatement and the product's generator must reproduce them.
This pins prime generation only: kernel-level parity with the reference stays unpinned."""
import ctypes
import json
import os

import numpy as np
import pytest

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seal_default_moduli.json")
SETS = json.load(open(FIXTURE))["sets"]


@pytest.mark.parametrize("entry", SETS, ids=[f"n{e['n']}" for e in SETS])
def test_oracle_get_primes_walk(orc, entry):
    for group in entry["walk"]:
        want = [int(p, 16) for p in group["primes"]]
        out = np.zeros(len(want), dtype=np.uint64)
        rc = orc.lib().orc_get_primes(entry["n"], group["bits"], len(want), orc.p(out))
        assert rc == 0
        assert [int(v) for v in out] == want, f"walk from 2^{group['bits']} at N = {entry['n']}"
        for q in want:
            assert q.bit_length() == group["bits"] and q % (2 * entry["n"]) == 1


def _create_order(entry):
    """CoeffModulus::Create on the bit sizes in ascending order: per bit size the smallest prime found first."""
    bits, want = [], []
    for group in entry["walk"]:
        primes = [int(p, 16) for p in group["primes"]]
        bits += [group["bits"]] * len(primes)
        want += primes[::-1]
    return bits, want


@pytest.mark.parametrize("entry", SETS, ids=[f"n{e['n']}" for e in SETS])
def test_coeff_modulus_create_reproduces_reference_moduli(pha, orc, entry):
    bits, want = _create_order(entry)
    assert [int(v) for v in orc.coeff_modulus_create(entry["n"], bits)] == want
    assert [int(v) for v in pha.coeff_modulus_create(entry["n"], bits)] == want
    # interleaved bit sizes: each request draws from its own bit size's list, in request order
    if len(entry["walk"]) == 2 and len(entry["walk"][0]["primes"]) >= 2:
        a = [int(p, 16) for p in entry["walk"][0]["primes"]][::-1]
        b = [int(p, 16) for p in entry["walk"][1]["primes"]][::-1]
        mixed_bits = [entry["walk"][0]["bits"], entry["walk"][1]["bits"], entry["walk"][0]["bits"]]
        # Create counts the requests per bit size first (two of the first size, one of the second here)
        na = len(a)
        got = [int(v) for v in pha.coeff_modulus_create(entry["n"], mixed_bits)]
        first = [int(p, 16) for p in entry["walk"][0]["primes"]][:2][::-1]
        second = [int(p, 16) for p in entry["walk"][1]["primes"]][:1]
        assert got == [first[0], second[0], first[1]]
        assert got == [int(v) for v in orc.coeff_modulus_create(entry["n"], mixed_bits)]
