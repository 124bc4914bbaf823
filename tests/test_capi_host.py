"""CPU-side checks of the product: the C-ABI library loads, exports every symbol that
include/phantom_amd.h declares, and its host number theory (table generation, modulus chain,
Galois tables, level constants) equals the oracle's.  No compute entry point is called."""
import ctypes
import re

import numpy as np
import pytest


def _declared_functions(header_path):
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pha_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pha):
    names = _declared_functions(pha.HEADER_PATH)
    assert len(names) >= 30
    handle = ctypes.CDLL(pha.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"declared in include/phantom_amd.h but not exported: {missing}"
    # and the Python binding covers all of them
    assert sorted(pha.SIGNATURES) == names


def test_version_string(pha):
    assert b"gfx950" in pha.lib().pha_version()


@pytest.mark.parametrize("n,bits", [(65536, [60, 60, 60, 50]), (65536, [60] + [50] * 35 + [60] * 12),
                                    (8192, [60, 40, 40, 60]), (4096, [36, 36, 37])])
def test_coeff_modulus_create_matches_oracle(pha, orc, n, bits):
    got = pha.coeff_modulus_create(n, bits)
    assert (got == orc.coeff_modulus_create(n, bits)).all()
    assert len(set(got.tolist())) == len(bits)
    for q, b in zip(got.tolist(), bits):
        assert q % (2 * n) == 1 and q.bit_length() == b


def test_coeff_modulus_create_rejects_bad_arguments(pha):
    with pytest.raises(pha.PhantomError) as e:
        pha.coeff_modulus_create(1000, [40])
    assert e.value.code == pha.PHA_ERR_INVALID_ARGUMENT
    with pytest.raises(pha.PhantomError):
        pha.coeff_modulus_create(4096, [61])
    with pytest.raises(pha.PhantomError) as e:
        pha.coeff_modulus_create(65536, [19] * 4)  # no such primes
    assert e.value.code == pha.PHA_ERR_LOGIC


@pytest.mark.parametrize("log_n,bits", [(12, 40), (14, 50), (16, 60)])
def test_host_ntt_tables_match_oracle(pha, orc, log_n, bits):
    q = pha.coeff_modulus_create(1 << log_n, [bits])
    o = orc.Oracle(log_n, q, 0)
    tw, tws, itw, itws, ni, nis = pha.host_ntt_tables(log_n, q[0])
    for which, arr in enumerate((tw, tws, itw, itws)):
        assert (arr == o.table(which, 0)).all()
    assert ni == int(o.table(4)[0]) and nis == int(o.table(5)[0])


def test_host_ntt_tables_reject_non_ntt_prime(pha):
    with pytest.raises(pha.PhantomError):
        pha.host_ntt_tables(12, 97)          # prime but 97 != 1 mod 8192
    with pytest.raises(pha.PhantomError):
        pha.host_ntt_tables(12, 8193 * 3)    # not prime


def test_barrett_ratio(pha):
    q = 1152921504597016577
    assert pha.host_barrett_ratio(q) == (2516582144, 16)
    lo, hi = pha.host_barrett_ratio(1125899903827969)
    assert (hi << 64) + lo == (1 << 128) // 1125899903827969


@pytest.mark.parametrize("log_n", [12, 16])
def test_galois_tables_match_oracle(pha, orc, log_n):
    for step in (1, -1, 2, 100, 0):
        elt = pha.galois_elt_from_step(log_n, step)
        assert elt == orc.galois_elt_from_step(log_n, step)
        assert (pha.host_galois_table(log_n, elt) == orc.galois_table(log_n, elt)).all()
    with pytest.raises(pha.PhantomError):
        pha.galois_elt_from_step(log_n, 1 << (log_n - 1))
    with pytest.raises(pha.PhantomError):
        pha.host_galois_table(log_n, 4)


@pytest.mark.parametrize("size_p,size_ql", [(1, 4), (2, 5), (3, 6), (12, 36)])
def test_level_constants_match_oracle(pha, orc, size_p, size_ql):
    n = 4096
    size_q = max(size_ql, 6) if size_p < 12 else 36
    mods = orc.coeff_modulus_create(n, [45] * size_q + [50] * size_p)
    o = orc.Oracle(12, mods, size_p)
    got = pha.host_level_constants(mods, size_p, size_ql)
    ref = o.level_constants(size_ql)
    for k in ref:
        assert (got[k] == ref[k]).all(), k
