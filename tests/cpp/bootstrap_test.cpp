// The flow of the reference's examples/bootstrapping.cu and tests/bootstrapping_test.cu on the C++ host layer
// (phantom-fhe_amd/host/phantom_boot.hpp), at N = 2^12 so that it runs in a second: a ciphertext at the last level
// comes back at a high level and decrypts to the same message.  Prints one JSON line per case.
#include <chrono>
#include <cstdio>
#include <random>

#include "../../phantom-fhe_amd/host/phantom_boot.hpp"

using namespace phantom;

static int run_case(const char *name, long logN, long logn, int logp, int logq0, int log_evalmod, int log_c2s, int remaining_level,
                    size_t special, int reps, long boundary_K = 12, long scale_factor = 3, size_t hamming_weight = 32,
                    int lt_levels = 3, bool slim = false) {
    const long deg = 31, inverse_deg = 1, loge = 10;
    const int lt = std::min<long>(lt_levels, logn), boot_level = lt + (5 + 1) + static_cast<int>(scale_factor) + lt;
    std::vector<int> coeff_bit_vec;
    coeff_bit_vec.push_back(logq0);
    for (int i = 0; i < remaining_level; i++) coeff_bit_vec.push_back(logp);
    for (int i = 0; i < lt; i++) coeff_bit_vec.push_back(logp);                        // SlotToCoeff
    for (int i = 0; i < boot_level - 2 * lt; i++) coeff_bit_vec.push_back(log_evalmod);  // EvalMod
    for (int i = 0; i < lt; i++) coeff_bit_vec.push_back(log_c2s);                     // CoeffToSlot
    for (size_t i = 0; i < special; i++) coeff_bit_vec.push_back(60);
    const int total_level = remaining_level + boot_level;

    EncryptionParameters parms;
    const size_t poly_modulus_degree = size_t(1) << logN, sparse_slots = size_t(1) << logn;
    const double scale = std::pow(2.0, logp);
    parms.set_poly_modulus_degree(poly_modulus_degree);
    parms.set_coeff_modulus(arith::CoeffModulus::Create(poly_modulus_degree, coeff_bit_vec));
    parms.set_special_modulus_size(special);
    parms.set_secret_key_hamming_weight(hamming_weight);
    parms.set_sparse_slots(sparse_slots);
    PhantomContext context(parms);
    PhantomSecretKey secret_key(context, 21);
    hip_stream_wrapper sw;
    PhantomRelinKey relin_keys = secret_key.gen_relinkey(context, sw);
    PhantomGaloisKey galois_keys;
    PhantomCKKSEncoder encoder(context);
    check_hip(hipDeviceSynchronize(), "sync");
    CKKSEvaluator ckks_evaluator(&context, &secret_key, &encoder, &relin_keys, &galois_keys, scale);

    Bootstrapper bootstrapper(loge, logn, logN - 1, total_level, scale, boundary_K, deg, scale_factor, inverse_deg, &ckks_evaluator);
    bootstrapper.lt_levels = lt;
    bootstrapper.slim = slim;
    bootstrapper.evalmod_scale = std::pow(2.0, log_evalmod);
    bootstrapper.prepare_mod_polynomial();
    std::vector<int> gal_steps_vector;
    gal_steps_vector.push_back(0);
    for (int i = 0; i < logN - 1; i++) gal_steps_vector.push_back(1 << i);
    bootstrapper.addLeftRotKeys_Linear_to_vector_3(gal_steps_vector);
    ckks_evaluator.create_galois_keys_from_steps(gal_steps_vector);
    bootstrapper.slot_vec.push_back(logn);
    bootstrapper.generate_LT_coefficient_3();

    std::mt19937_64 rnd(7);
    std::uniform_real_distribution<double> dist(-1, 1);
    cvec msg(sparse_slots);
    for (auto &v : msg) v = {dist(rnd), dist(rnd)};
    // encode at the top, walk down to the last level like the example does
    PhantomCiphertext cipher = ckks_evaluator.encrypt(ckks_evaluator.encode(msg, scale, 1, sparse_slots));
    for (int i = 0; i < total_level - static_cast<int>(bootstrapper.pre_levels()); i++) mod_switch_to_next_inplace(context, cipher, ckks_evaluator.stream);
    cvec before = ckks_evaluator.decrypt_decode(cipher, sparse_slots);

    PhantomCiphertext rtn;
    bootstrapper.bootstrap_3(rtn, cipher);
    ckks_evaluator.sync();
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        bootstrapper.bootstrap_3(rtn, cipher);
        ckks_evaluator.sync();
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    cvec after = ckks_evaluator.decrypt_decode(rtn, sparse_slots);
    double mean_err = 0, max_err = 0, in_err = 0;
    for (size_t i = 0; i < sparse_slots; i++) {
        mean_err += std::abs(before[i] - after[i]);
        max_err = std::max(max_err, std::abs(msg[i] - after[i]));
        in_err = std::max(in_err, std::abs(msg[i] - before[i]));
    }
    mean_err /= static_cast<double>(sparse_slots);
    // the refreshed ciphertext computes: square it, rescale
    PhantomCiphertext sq = ckks_evaluator.rescale(ckks_evaluator.mul(rtn, rtn));
    cvec sqv = ckks_evaluator.decrypt_decode(sq, sparse_slots);
    double sq_err = 0;
    for (size_t i = 0; i < sparse_slots; i++) sq_err = std::max(sq_err, std::abs(sqv[i] - msg[i] * msg[i]));
    printf("{\"%s\": {\"log_n\": %ld, \"slots\": %zu, \"input_err\": %.3e, \"mean_err\": %.3e, \"max_err\": %.3e, \"square_err\": %.3e, "
           "\"return_limbs\": %zu, \"return_level\": %zu, \"levels_needed\": %zu, \"rotation_keys\": %zu, \"ms\": %.3f}}\n",
           name, logN, sparse_slots, in_err, mean_err, max_err, sq_err, rtn.coeff_modulus_size(), rtn.chain_index(),
           bootstrapper.levels_needed(), gal_steps_vector.size(), best);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc > 1 && std::string(argv[1]) == "bench") {
            // N = 2^16, full slots, the shape of tools/bench_bootstrap.py --slots 32768 --transform fft
            // (bench 4: four levels per transform, 13 remaining levels)
            if (argc > 2 && std::string(argv[2]) == "4") return run_case("bench_full_2^16_lt4", 16, 15, 46, 56, 58, 60, 13, 4, 3, 25, 4, 192, 4);
            if (argc > 2 && std::string(argv[2]) == "slim") return run_case("bench_full_2^16_lt4_slim", 16, 15, 46, 56, 58, 60, 9, 4, 3, 25, 4, 192, 4, true);
            return run_case("bench_full_2^16", 16, 15, 46, 56, 58, 60, 15, 4, 3, 25, 4, 192);
        }
        run_case("full_slots", 12, 11, 50, 60, 50, 60, 2, 1, 1);
        run_case("sparse_128", 12, 7, 40, 50, 40, 58, 2, 1, 1);
        run_case("full_slots_slim", 12, 11, 50, 60, 50, 60, 2, 1, 1, 12, 3, 32, 3, true);
        run_case("sparse_128_slim", 12, 7, 40, 50, 40, 58, 2, 1, 1, 12, 3, 32, 3, true);
        // a wrong scale is refused like the reference's argument checks
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "FAILED: %s\n", e.what());
        return 1;
    }
}
