"""The C++ host layer (phantom-fhe_amd/host/phantom_amd.hpp: PhantomContext, PhantomCiphertext,
multiply_inplace / relinearize_inplace / rescale_to_next / rotate_inplace with the reference's
names and error behaviour) driven by a C++ program, checked against the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

BIN = os.path.join(os.path.dirname(__file__), "cpp", "host_mirror_test")


def test_header_only_layer_is_self_contained():
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "phantom-fhe_amd", "host", "phantom_amd.hpp")).read()
    for name in ("multiply_inplace", "relinearize_inplace", "rescale_to_next", "rotate_inplace", "apply_galois_inplace",
                 "class PhantomContext", "class PhantomCiphertext", "class PhantomRelinKey", "class PhantomGaloisKey",
                 "class PhantomSecretKey", "class PhantomPlaintext", "encrypt_symmetric", "gen_relinkey",
                 "class PhantomPublicKey", "gen_publickey", "encrypt_asymmetric"):
        assert name in hdr
    assert "oracle" not in hdr


def test_bootstrap_header_mirrors_the_reference_api():
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "phantom-fhe_amd", "host", "phantom_boot.hpp")).read()
    for name in ("class CKKSEvaluator", "class Bootstrapper", "prepare_mod_polynomial", "addLeftRotKeys_Linear_to_vector_3",
                 "generate_LT_coefficient_3", "bootstrap_3", "bootstrap_sparse_3", "bootstrap_full_3", "bootstrap_inplace_3",
                 "modraise_inplace", "slot_vec", "create_galois_keys_from_steps"):
        assert name in hdr
    assert "oracle" not in hdr


def test_cnn_header_mirrors_the_reference_api():
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "phantom-fhe_amd", "host", "cnn_phantom.hpp")).read()
    for name in ("class TensorCipher", "multiplexed_parallel_convolution_phantom", "multiplexed_parallel_batch_norm_phantom",
                 "ReLU_phantom", "cnn_add_phantom", "multiplexed_parallel_downsampling_phantom", "averagepooling_phantom_scale",
                 "matrix_multiplication_phantom", "import_parameters_cifar10"):
        assert name in hdr
    assert "oracle" not in hdr


@pytest.mark.gpu
def test_cpp_resnet_with_bootstrapping(gpu):
    """The flow of the reference's examples/cnn.cu on the C++ host layer (cnn_phantom.hpp): three residual blocks on
    8 x 8 images at N = 2^13, six bootstrappings, logits against the same network on plain numbers."""
    exe = os.path.join(os.path.dirname(__file__), "cpp", "cnn_test")
    assert os.path.exists(exe), "tests/cpp/cnn_test not built (run __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = {}
    for l in out.stdout.splitlines():
        if l.startswith("{"):
            lines.update(json.loads(l))
    for key, special in (("resnet8_2^13", 8), ("resnet8_2^13_precise", 4)):     # dnum 4 with 50-bit special primes, dnum 8 with 60-bit ones
        res = lines[key]
        assert res["bootstrappings"] == 6 and res["relu_levels"] == 14 and res["relu_alpha"] > 13.0 and res["special_primes"] == special
        assert res["max_logit_error_vs_plain_same_relu"] < 1e-3 and res["argmax"] == res["argmax_plain"]


@pytest.mark.gpu
def test_cpp_bootstrapper(gpu):
    """The flow of the reference's examples/bootstrapping.cu on the C++ host layer: full slots (2048 of N = 2^12)
    and a sparse packing.  tests/bootstrapping_test.cu asks for a mean error around 1e-3."""
    exe = os.path.join(os.path.dirname(__file__), "cpp", "bootstrap_test")
    assert os.path.exists(exe), "tests/cpp/bootstrap_test not built (run __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    res = {}
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            res.update(json.loads(line))
    full, sparse = res["full_slots"], res["sparse_128"]
    assert full["slots"] == 2048 and full["return_limbs"] == 3 and full["return_level"] == full["levels_needed"] == 16
    assert full["input_err"] < 1e-9 and full["max_err"] < 2e-5 and full["square_err"] < 4e-5      # delta = 2^50
    assert sparse["slots"] == 128 and sparse["return_limbs"] == 3
    assert sparse["max_err"] < 1e-3 and sparse["mean_err"] < 2e-4 and sparse["square_err"] < 2e-3  # delta = 2^40
    # slim order: SlotToCoeff first (3 levels at the end of the chain), the result comes back 3 levels earlier
    fs, ss = res["full_slots_slim"], res["sparse_128_slim"]
    assert fs["return_level"] == fs["levels_needed"] == 13 and fs["return_limbs"] == 6
    assert fs["max_err"] < 3e-5 and fs["square_err"] < 1e-4            # floor: the sine's cubic term at |z| ~ 1.4
    assert ss["return_level"] == 13 and ss["max_err"] < 1e-4 and ss["mean_err"] < 2e-5


@pytest.mark.gpu
def test_cpp_host_layer_against_oracle(gpu, orc):
    assert os.path.exists(BIN), "tests/cpp/host_mirror_test not built (run __graft_entry__.build())"
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    res = {}
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            res.update(json.loads(line))
    size_p, n = 2, 4096
    mods = orc.coeff_modulus_create(n, [45, 45, 45, 45, 55, 55])
    o = orc.Oracle(12, mods, size_p)

    def key(seed):
        return [np.ascontiguousarray(np.stack([o.uniform_poly(list(range(6)), seed + 10 * d),
                                               o.uniform_poly(list(range(6)), seed + 10 * d + 1)])) for d in range(2)]

    def ct(limbs, s0, s1):
        return np.stack([o.uniform_poly(list(range(limbs)), s0), o.uniform_poly(list(range(limbs)), s1)])

    def digest(a):
        return f"{orc.fnv1a64(np.ascontiguousarray(a)):016x}"

    rlk = key(70)
    e1, e4, em1 = (orc.galois_elt_from_step(12, s) for s in (1, 4, -1))
    assert res["galois_elts"] == [e1, e4, em1]
    gk = {e1: key(170), e4: key(270), em1: key(370)}
    # multiply + relin + rescale
    x, y = ct(4, 7, 8), ct(4, 17, 18)
    tp = o.tensor_prod_2x2(x, y)
    mr = tp[:2].copy()
    o.keyswitch_inplace(4, mr, tp[2], rlk)
    assert res["multiply_relin"] == digest(mr) and res["size"] == 2
    assert res["scale"] == 1099511627776.0 ** 2
    rs = o.divide_and_round_q_last_ntt(4, mr, 2)
    assert res["rescale"] == digest(rs) and res["chain_index"] == 2 and res["limbs"] == 3
    # square at level 2
    sq = ct(3, 27, 28)
    tp = o.tensor_prod_2x2(sq, sq)
    s2 = tp[:2].copy()
    o.keyswitch_inplace(3, s2, tp[2], rlk)
    assert res["square_relin_level2"] == digest(s2)

    def rotate(c, elt):
        out = np.stack([o.apply_galois_ntt(c[0], elt), np.zeros_like(c[1])])
        o.keyswitch_inplace(c.shape[1], out, o.apply_galois_ntt(c[1], elt), gk[elt])
        return out

    assert res["rotate1"] == digest(rotate(ct(4, 37, 38), e1))
    # step 3 = NAF (-1, +4): applied in that order (least significant digit first)
    assert res["rotate3_naf"] == digest(rotate(rotate(ct(4, 47, 48), em1), e4))
    a, b = ct(4, 57, 58), ct(4, 67, 68)
    q = mods[:4][None, :, None]
    add = (a + b) % q
    assert res["add"] == digest(add)
    ns = (2 * q - add - b) % q          # -(a+b) - b
    assert res["neg_sub"] == digest(ns)
    assert res["mod_switch_to_3"] == digest(ns[:, :2]) and res["ms_limbs"] == 2
    assert res["errors_caught"] == 5
    assert res["out_of_place_ok"] == 7       # add / sub(negate) / add_many / negate + mod_switch_to / rotate / multiply_and_relin / plaintext
    # keygen -> encrypt -> HMult+relin -> decrypt with real keys: product of the constants, small noise
    assert res["real_keys_rel_err"] < 1e-3 and res["real_keys_rest"] < 1e-3
    assert res["gen_twice_throws"] == 1
    assert res["encoder_pipeline_err"] < 1e-4
    # public-key encryption: decrypts to the message, two encryptions of one plaintext differ.  Scale 2^30; the
    # division by P is the reference's uncentred moddown, whose +1/2 mean per coefficient lands on slot 0 (9e-5)
    assert res["asymmetric_err"] < 1e-3 and res["asymmetric_fresh_randomness"] > 0.99 and res["asymmetric_limbs"] == 4


@pytest.mark.gpu
def test_cpp_limb_parallel_host_layer(gpu):
    """phantom-fhe_amd/host/limb_parallel.hpp (the RCCL twin of limb_parallel.py): HMult + relinearise, rotation and rescale
    through LocalComm and through a one-rank RCCL communicator equal the single-GPU entry points bit for bit."""
    binary = os.path.join(os.path.dirname(__file__), "cpp", "lp_test")
    assert os.path.exists(binary), "tests/cpp/lp_test not built (run __graft_entry__.build())"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([binary], capture_output=True, text=True, timeout=170, env=env)
    assert out.returncode == 0 and "lp_test ok" in out.stdout, out.stdout + out.stderr[-2000:]
    assert "local: rank 0 of 1" in out.stdout and "rccl: rank 0 of 1" in out.stdout
