// The flow of the reference's examples/cnn.cu (ResNet_cifar10_seal_sparse) on the C++ host layer
// (phantom-fhe_amd/host/cnn_phantom.hpp): a three-block ResNet on 8 x 8 images at N = 2^13 with a bootstrapping before
// every ReLU after the first, logits against the same network on plain numbers.  `cnn_test bench <dir>` runs
// ResNet-20 at N = 2^16 on the parameter directory <dir> (the reference's pretrained_parameters/resnet20_new format).
// Prints one JSON line.
#include <chrono>
#include <cstdio>
#include <random>

#include "../../phantom-fhe_amd/host/cnn_phantom.hpp"

using namespace phantom;
using Tensor = std::vector<double>;  // [c][h][w]

// ---- the network on plain numbers (the checker) ---------------------------------------------------------------------
static Tensor conv3x3(const Tensor &x, int ci, int h, int w, const std::vector<double> &wt, int co, int st) {
    const int ho = h / st, wo = w / st;
    Tensor out(static_cast<size_t>(co) * ho * wo, 0.0);
    for (int o = 0; o < co; o++)
        for (int r = 0; r < ho; r++)
            for (int c = 0; c < wo; c++) {
                double acc = 0;
                for (int i = 0; i < ci; i++)
                    for (int i1 = 0; i1 < 3; i1++)
                        for (int i2 = 0; i2 < 3; i2++) {
                            const int rr = r * st + i1 - 1, cc = c * st + i2 - 1;
                            if (rr < 0 || rr >= h || cc < 0 || cc >= w) continue;
                            acc += wt[((o * ci + i) * 3 + i1) * 3 + i2] * x[(i * h + rr) * w + cc];
                        }
                out[(o * ho + r) * wo + c] = acc;
            }
    return out;
}
static void batch_norm(Tensor &x, int c, int hw, const CnnParameters &p, const std::string &bn) {
    for (int i = 0; i < c; i++) {
        const double f = p.at(bn + "_weight")[i] / std::sqrt(p.at(bn + "_running_var")[i] + 1e-5);
        for (int j = 0; j < hw; j++) x[i * hw + j] = (x[i * hw + j] - p.at(bn + "_running_mean")[i]) * f + p.at(bn + "_bias")[i];
    }
}
static std::vector<double> plain_resnet(const CnnParameters &p, Tensor x, int h, size_t end_num, const std::function<double(double)> &relu) {
    const int widths[3] = {16, 32, 64};
    int c = 3, w = h;
    auto act = [&](Tensor &t) {
        for (double &v : t) v = relu(v);
    };
    x = conv3x3(x, c, h, w, p.at("conv1_weight"), 16, 1);
    c = 16;
    batch_norm(x, c, h * w, p, "bn1");
    act(x);
    for (int j = 0; j < 3; j++)
        for (size_t k = 0; k <= end_num; k++) {
            const std::string stem = "layer" + std::to_string(j + 1) + "_" + std::to_string(k);
            const int st = (j >= 1 && k == 0) ? 2 : 1, co = widths[j], ho = h / st, wo = w / st;
            Tensor y = conv3x3(x, c, h, w, p.at(stem + "_conv1_weight"), co, st);
            batch_norm(y, co, ho * wo, p, stem + "_bn1");
            act(y);
            y = conv3x3(y, co, ho, wo, p.at(stem + "_conv2_weight"), co, 1);
            batch_norm(y, co, ho * wo, p, stem + "_bn2");
            if (st == 2) {  // every other pixel, c/2 zero channels on both sides
                for (int i = 0; i < c; i++)
                    for (int r = 0; r < ho; r++)
                        for (int q = 0; q < wo; q++) y[((i + c / 2) * ho + r) * wo + q] += x[(i * h + 2 * r) * w + 2 * q];
            } else
                for (size_t i = 0; i < y.size(); i++) y[i] += x[i];
            act(y);
            x = std::move(y), c = co, h = ho, w = wo;
        }
    const size_t classes = p.at("linear_bias").size();
    std::vector<double> out(classes);
    for (size_t o = 0; o < classes; o++) {
        double acc = p.at("linear_bias")[o];
        for (int i = 0; i < c; i++) {
            double mean = 0;
            for (int j = 0; j < h * w; j++) mean += x[i * h * w + j];
            acc += p.at("linear_weight")[o * c + i] * mean / (h * w);
        }
        out[o] = acc;
    }
    return out;
}
static CnnParameters synthetic_parameters(size_t end_num, uint64_t seed) {
    std::mt19937_64 rnd(seed);
    std::normal_distribution<double> normal(0.0, 1.0);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    CnnParameters p;
    auto fill = [&](const std::string &name, size_t count, double sd) {
        auto &v = p[name];
        for (size_t i = 0; i < count; i++) v.push_back(sd * normal(rnd));
    };
    auto bn = [&](const std::string &stem, int c) {
        fill(stem + "_bias", c, 0.1);
        fill(stem + "_running_mean", c, 0.1);
        for (int i = 0; i < c; i++) p[stem + "_running_var"].push_back(0.8 + 0.4 * uni(rnd));
        for (int i = 0; i < c; i++) p[stem + "_weight"].push_back(0.5 + 0.5 * uni(rnd));
    };
    fill("conv1_weight", 16 * 3 * 9, std::sqrt(2.0 / 27));
    bn("bn1", 16);
    int ci = 16;
    const int widths[3] = {16, 32, 64};
    for (int j = 0; j < 3; j++)
        for (size_t k = 0; k <= end_num; k++) {
            const std::string stem = "layer" + std::to_string(j + 1) + "_" + std::to_string(k);
            const int co = widths[j];
            fill(stem + "_conv1_weight", static_cast<size_t>(co) * ci * 9, std::sqrt(2.0 / (9 * ci)));
            bn(stem + "_bn1", co);
            fill(stem + "_conv2_weight", static_cast<size_t>(co) * co * 9, std::sqrt(1.0 / (9 * co)));
            bn(stem + "_bn2", co);
            ci = co;
        }
    fill("linear_weight", 10 * 64, 0.3);
    fill("linear_bias", 10, 0.1);
    return p;
}

// fast chain (default): 38-bit messages, 48-bit EvalMod primes, EIGHT 50-bit special primes (dnum 4, every prime on the FP64
// transform path, log2 QP = 1750 at N = 2^16); precise: 42 / 56 bits, four 60-bit special primes (dnum 8, log2 QP = 1758)
static int run(const char *name, long logN, int image_hw, size_t end_num, CnnParameters params, double B, size_t hamming_weight, long boundary_K,
               long scale_factor, int reps, bool precise = false) {
    const int logp = precise ? 42 : 38, logq0 = logp + 10, log_evalmod = precise ? 56 : 48, log_c2s = logp + 8, c2s_levels = 3, s2c_levels = 2;
    const size_t special = precise ? 4 : 8;
    const int special_bits = precise ? 60 : 50;
    ApproxReLU relu;
    const int cnn_levels = relu.depth + 2, evalmod_levels = (5 + 1) + static_cast<int>(scale_factor);
    std::vector<int> bits{logq0};
    for (int i = 0; i < cnn_levels + s2c_levels; i++) bits.push_back(logp);
    for (int i = 0; i < evalmod_levels; i++) bits.push_back(log_evalmod);
    for (int i = 0; i < c2s_levels; i++) bits.push_back(log_c2s);
    while (bits.size() % special) bits.insert(bits.begin() + 1, logp);
    const size_t q_limbs = bits.size();
    for (size_t i = 0; i < special; i++) bits.push_back(special_bits);

    EncryptionParameters parms;
    const size_t N = size_t(1) << logN;
    const double scale = std::pow(2.0, logp);
    parms.set_poly_modulus_degree(N);
    parms.set_coeff_modulus(arith::CoeffModulus::Create(N, bits));
    parms.set_special_modulus_size(special);
    parms.set_secret_key_hamming_weight(hamming_weight);
    PhantomContext context(parms);
    PhantomSecretKey secret_key(context, 21);
    hip_stream_wrapper sw;
    PhantomRelinKey relin_keys = secret_key.gen_relinkey(context, sw);
    PhantomGaloisKey galois_keys;
    PhantomCKKSEncoder encoder(context);
    check_hip(hipDeviceSynchronize(), "sync");
    CKKSEvaluator ev(&context, &secret_key, &encoder, &relin_keys, &galois_keys, scale);
    ev.auto_keys = true;

    // one bootstrapper per packing: n/2, n/4, n/8 slots of the n = N/2 (examples/cnn.cu:144-186)
    const long logn = logN - 1, first = logn - (image_hw == 32 ? 1 : 2);
    std::vector<std::unique_ptr<Bootstrapper>> boots;
    ResNetCifar10 net(&ev, std::move(params), end_num, B, relu, 3, image_hw, image_hw);
    for (long ls = first; ls > first - 3; ls--) {
        boots.push_back(std::make_unique<Bootstrapper>(10, ls, logn, static_cast<long>(q_limbs) - 1, scale, boundary_K, 31, scale_factor, 1, &ev));
        Bootstrapper &b = *boots.back();
        b.lt_levels = c2s_levels, b.s2c_lt_levels = s2c_levels;
        b.slim = true;
        b.evalmod_scale = std::pow(2.0, log_evalmod);
        b.prepare_mod_polynomial();
        std::vector<int> steps{0};
        b.addLeftRotKeys_Linear_to_vector_3(steps);
        ev.create_galois_keys_from_steps(steps);
        b.generate_LT_coefficient_3();
        net.bootstrappers[1L << ls] = &b;
    }
    std::mt19937_64 rnd(3);
    std::normal_distribution<double> normal(0.0, 1.0);
    Tensor image(3 * image_hw * image_hw);
    for (double &v : image) v = std::max(-2.5, std::min(2.5, normal(rnd)));
    const size_t level = net.input_level();
    TensorCipher x = net.encrypt_image(image, scale, level);
    TensorCipher out = net(x);  // first run: encodes the constants, generates rotation keys
    ev.sync();
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        out = net(x);
        ev.sync();
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::vector<double> logits = net.decrypt_logits(out);
    std::vector<double> want = plain_resnet(net.params, image, image_hw, end_num, [&](double v) { return B * relu(v / B); });
    std::vector<double> exact = plain_resnet(net.params, image, image_hw, end_num, [](double v) { return std::max(v, 0.0); });
    double err = 0, err_exact = 0;
    size_t arg = 0, arg_exact = 0;
    for (size_t i = 0; i < logits.size(); i++) {
        err = std::max(err, std::abs(logits[i] - want[i]));
        err_exact = std::max(err_exact, std::abs(logits[i] - exact[i]));
        if (logits[i] > logits[arg]) arg = i;
        if (exact[i] > exact[arg_exact]) arg_exact = i;
    }
    printf("{\"%s\": {\"log_n\": %ld, \"q_limbs\": %zu, \"special_primes\": %zu, \"relu_alpha\": %.2f, \"relu_levels\": %d, \"bootstrappings\": %zu, "
           "\"max_logit_error_vs_plain_same_relu\": %.3e, \"max_logit_error_vs_exact_relu\": %.3e, \"argmax\": %zu, \"argmax_plain\": %zu, "
           "\"out_level\": %zu, \"encoded_constants\": %zu, \"ms\": %.2f}}\n",
           name, logN, q_limbs, special, -std::log2(relu.precision), relu.depth, 6 * (end_num + 1), err, err_exact, arg, arg_exact,
           out.cipher().chain_index(), net.cache.size(), best);
    fflush(stdout);
    return err < 1e-2 && arg == arg_exact ? 0 : 2;
}

int main(int argc, char **argv) {
    try {
        if (argc > 2 && std::string(argv[1]) == "bench")
            return run("resnet20_2^16", 16, 32, 2, import_parameters_cifar10(argv[2], 2), 40.0, 192, 25, 4, 2, argc > 3 && std::string(argv[3]) == "precise");
        const int rc = run("resnet8_2^13", 13, 8, 0, synthetic_parameters(0, 3), 16.0, 32, 12, 3, 1);
        return rc ? rc : run("resnet8_2^13_precise", 13, 8, 0, synthetic_parameters(0, 3), 16.0, 32, 12, 3, 1, true);
    } catch (const std::exception &e) {
        fprintf(stderr, "FAILED: %s\n", e.what());
        return 1;
    }
}
