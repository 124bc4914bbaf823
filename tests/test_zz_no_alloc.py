"""pha_rescale_to_next_noalloc (the entry point that allocates nothing) against pha_rescale_to_next, through the
C ABI, bit for bit, on two calls in a row, the second with a smaller input; and against the CPU oracle."""
import ctypes

import numpy as np
import pytest


@pytest.mark.gpu
def test_noalloc_rescale_matches_rescale_twice(gpu, pha, orc):
    log_n, n = 13, 1 << 13
    mods = orc.coeff_modulus_create(n, [55] + [45] * 5 + [60] * 2)
    ctx = pha.PhantomContext(log_n, mods, 2)
    o = orc.Oracle(log_n, mods, 2)
    lib, h = pha.lib(), ctx._h
    stream = ctypes.c_void_p(gpu.cuda.current_stream().cuda_stream)

    def run(chain_index, cipher_size, seed, poison):
        size_ql = 6 - (chain_index - 1)
        src_h = np.stack([o.uniform_poly(list(range(size_ql)), seed + p) for p in range(cipher_size)])
        src = pha.to_device(src_h)
        want = gpu.full((cipher_size, size_ql - 1, n), poison, dtype=gpu.int64, device="cuda")
        got = gpu.full((cipher_size, size_ql - 1, n), poison + 1, dtype=gpu.int64, device="cuda")
        assert lib.pha_rescale_to_next(h, chain_index, ctypes.c_void_p(src.data_ptr()), cipher_size,
                                       ctypes.c_void_p(want.data_ptr()), stream) == 0
        assert lib.pha_rescale_to_next_noalloc(h, chain_index, ctypes.c_void_p(src.data_ptr()), cipher_size,
                                               ctypes.c_void_p(got.data_ptr()), stream) == 0, pha.lib().pha_last_error().decode()
        gpu.cuda.synchronize()
        assert gpu.equal(src, pha.to_device(src_h)), "src must be preserved"
        assert gpu.equal(got, want), f"noalloc differs at chain_index {chain_index}, cipher_size {cipher_size}"
        assert (pha.to_host(got) == o.divide_and_round_q_last_ntt(size_ql, src_h, cipher_size)).all(), \
            "noalloc differs from the oracle"

    # a level that was never used: the entry point must refuse (uploading its constants would allocate)
    src = gpu.zeros((2, 6, n), dtype=gpu.int64, device="cuda")
    dst = gpu.zeros((2, 5, n), dtype=gpu.int64, device="cuda")
    rc = lib.pha_rescale_to_next_noalloc(h, 1, ctypes.c_void_p(src.data_ptr()), 2, ctypes.c_void_p(dst.data_ptr()), stream)
    assert rc != 0 and "not prepared" in pha.lib().pha_last_error().decode()
    ctx.prepare_level(1)
    ctx.prepare_level(3)
    run(1, 3, seed=10, poison=-1)      # 3 polys, 6 limbs
    run(3, 2, seed=20, poison=-7)      # then a smaller input: 2 polys, 4 limbs
    run(1, 2, seed=30, poison=-3)
    # beyond the scratch: refused with a clear error, nothing launched
    too_many = (3 << 17) // n + 1
    rc = lib.pha_rescale_to_next_noalloc(h, 1, ctypes.c_void_p(src.data_ptr()), too_many, ctypes.c_void_p(dst.data_ptr()), stream)
    assert rc != 0 and "exceeds" in pha.lib().pha_last_error().decode()
    ctx.close()


@pytest.mark.gpu
def test_noalloc_rescale_under_graph_capture(gpu, pha, orc):
    """The defining property: after prepare_level the call can be captured into a HIP graph on a context whose
    workspace was never grown (an allocation or a synchronisation inside it would invalidate the capture); the replays
    reproduce pha_rescale_to_next of a second context bit for bit, also after the input changed."""
    log_n, n = 13, 1 << 13
    mods = orc.coeff_modulus_create(n, [55] + [45] * 5 + [60] * 2)
    ctx = pha.PhantomContext(log_n, mods, 2)        # captured: nothing but prepare_level ever runs on it eagerly
    ref = pha.PhantomContext(log_n, mods, 2)        # the allocating entry point runs here
    o = orc.Oracle(log_n, mods, 2)
    lib = pha.lib()
    chain_index, cipher_size, size_ql = 2, 2, 5
    ctx.prepare_level(chain_index)
    src = gpu.zeros((cipher_size, size_ql, n), dtype=gpu.int64, device="cuda")
    got = gpu.full((cipher_size, size_ql - 1, n), -5, dtype=gpu.int64, device="cuda")
    want = gpu.full((cipher_size, size_ql - 1, n), -6, dtype=gpu.int64, device="cuda")
    gpu.cuda.synchronize()
    graph = gpu.cuda.CUDAGraph()
    with gpu.cuda.graph(graph):
        cap_stream = ctypes.c_void_p(gpu.cuda.current_stream().cuda_stream)
        rc = lib.pha_rescale_to_next_noalloc(ctx._h, chain_index, ctypes.c_void_p(src.data_ptr()), cipher_size,
                                             ctypes.c_void_p(got.data_ptr()), cap_stream)
        assert rc == 0, lib.pha_last_error().decode()
    stream = ctypes.c_void_p(gpu.cuda.current_stream().cuda_stream)
    for seed in (40, 50):
        src_h = np.stack([o.uniform_poly(list(range(size_ql)), seed + p) for p in range(cipher_size)])
        src.copy_(pha.to_device(src_h))
        graph.replay()
        assert lib.pha_rescale_to_next(ref._h, chain_index, ctypes.c_void_p(src.data_ptr()), cipher_size,
                                       ctypes.c_void_p(want.data_ptr()), stream) == 0
        gpu.cuda.synchronize()
        assert gpu.equal(got, want), f"replay differs from pha_rescale_to_next (seed {seed})"
        assert (pha.to_host(got) == o.divide_and_round_q_last_ntt(size_ql, src_h, cipher_size)).all()
    del graph
    ctx.close()
    ref.close()
