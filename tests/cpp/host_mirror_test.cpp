// Drives the C++ host layer (phantom-fhe_amd/host/phantom_amd.hpp) the way the reference's tests
// drive evaluate.cuh (tests/relin.cu, tests/rotation.cu, tests/rescale_to_next.cu): build
// parameters, context and keys, run an evaluator entry point, report the result.  Inputs are the
// seeded uniform residues of the oracle (SplitMix64 + rejection), the results are printed as
// FNV-1a digests and compared with the oracle by tests/test_cpp_host.py.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>

#include "../../phantom-fhe_amd/host/phantom_amd.hpp"

using namespace phantom;

static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void fill_uniform(uint64_t *out, size_t count, uint64_t q, uint64_t seed) {
    uint64_t s = seed;
    const int bits = 64 - __builtin_clzll(q);
    const uint64_t mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
    for (size_t i = 0; i < count; i++) {
        uint64_t v;
        do v = splitmix64(s) & mask; while (v >= q);
        out[i] = v;
    }
}
// [limbs][n], limb i modulo moduli[idx0 + i], seeded like oracle.Oracle.uniform_poly
static std::vector<uint64_t> uniform_poly(const std::vector<uint64_t> &moduli, size_t idx0, size_t limbs, size_t n, uint64_t seed) {
    std::vector<uint64_t> v(limbs * n);
    for (size_t i = 0; i < limbs; i++) fill_uniform(v.data() + i * n, n, moduli[idx0 + i], seed * 1000003ULL + i);
    return v;
}
static uint64_t fnv(const std::vector<uint64_t> &v) {
    uint64_t h = 0xcbf29ce484222325ULL;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(v.data());
    for (size_t i = 0; i < v.size() * 8; i++) h = (h ^ p[i]) * 0x100000001b3ULL;
    return h;
}
static std::vector<uint64_t> download(const PhantomCiphertext &ct) {
    std::vector<uint64_t> h(ct.size() * ct.coeff_modulus_size() * ct.poly_modulus_degree());
    check_hip(hipDeviceSynchronize(), "sync");
    check_hip(hipMemcpy(h.data(), ct.data(), h.size() * 8, hipMemcpyDeviceToHost), "download");
    return h;
}
static PhantomCiphertext make_ct(const PhantomContext &ctx, const std::vector<uint64_t> &moduli, size_t chain, uint64_t seed0,
                                 uint64_t seed1, const hipStream_t &s) {
    PhantomCiphertext ct;
    ct.resize(ctx, chain, 2, s);
    const size_t limbs = ctx.coeff_mod_size(chain), n = ctx.poly_degree();
    auto a = uniform_poly(moduli, 0, limbs, n, seed0), b = uniform_poly(moduli, 0, limbs, n, seed1);
    check_hip(hipMemcpy(ct.data(), a.data(), a.size() * 8, hipMemcpyHostToDevice), "upload");
    check_hip(hipMemcpy(ct.data() + limbs * n, b.data(), b.size() * 8, hipMemcpyHostToDevice), "upload");
    ct.set_scale(1099511627776.0);
    return ct;
}
static PhantomRelinKey make_key(const std::vector<uint64_t> &moduli, size_t size_P, size_t n, uint64_t seed) {
    const size_t size_QP = moduli.size(), dnum = (size_QP - size_P) / size_P;
    std::vector<device_words> digits;
    for (size_t d = 0; d < dnum; d++) {
        auto b = uniform_poly(moduli, 0, size_QP, n, seed + 10 * d), a = uniform_poly(moduli, 0, size_QP, n, seed + 10 * d + 1);
        device_words w(2 * size_QP * n);
        check_hip(hipMemcpy(w.get(), b.data(), b.size() * 8, hipMemcpyHostToDevice), "upload");
        check_hip(hipMemcpy(w.get() + size_QP * n, a.data(), a.size() * 8, hipMemcpyHostToDevice), "upload");
        digits.push_back(std::move(w));
    }
    return PhantomRelinKey(std::move(digits));
}

int main() {
    try {
        const size_t n = 4096, size_P = 2;
        EncryptionParameters parms;
        parms.set_poly_modulus_degree(n);
        parms.set_coeff_modulus(arith::CoeffModulus::Create(n, {45, 45, 45, 45, 55, 55}));
        parms.set_special_modulus_size(size_P);
        PhantomContext context(parms);
        const auto &moduli = parms.coeff_modulus();
        hip_stream_wrapper sw;
        const hipStream_t &s = sw.get_stream();

        PhantomRelinKey rlk = make_key(moduli, size_P, n, 70);
        PhantomGaloisKey gk;
        uint32_t e1, e4, em1;
        check(pha_galois_elt_from_step(12, 1, &e1));
        check(pha_galois_elt_from_step(12, 4, &e4));
        check(pha_galois_elt_from_step(12, -1, &em1));
        gk.insert(e1, make_key(moduli, size_P, n, 170));
        gk.insert(e4, make_key(moduli, size_P, n, 270));
        gk.insert(em1, make_key(moduli, size_P, n, 370));
        printf("{\"galois_elts\": [%u, %u, %u]}\n", e1, e4, em1);

        // HMult + relinearize + rescale (tests/relin.cu, tests/rescale_to_next.cu)
        PhantomCiphertext x = make_ct(context, moduli, 1, 7, 8, s), y = make_ct(context, moduli, 1, 17, 18, s);
        multiply_and_relin_inplace(context, x, y, rlk, sw);
        printf("{\"multiply_relin\": \"%016llx\", \"size\": %zu, \"scale\": %.1f}\n", (unsigned long long)fnv(download(x)), x.size(),
               x.scale());
        PhantomCiphertext r = rescale_to_next(context, x, sw);
        printf("{\"rescale\": \"%016llx\", \"chain_index\": %zu, \"limbs\": %zu}\n", (unsigned long long)fnv(download(r)),
               r.chain_index(), r.coeff_modulus_size());
        // square of a ciphertext at the next level
        PhantomCiphertext sq = make_ct(context, moduli, 2, 27, 28, s);
        multiply_inplace(context, sq, sq, sw);
        relinearize_inplace(context, sq, rlk, sw);
        printf("{\"square_relin_level2\": \"%016llx\"}\n", (unsigned long long)fnv(download(sq)));
        // rotate by 1 (key present) and by 3 (NAF: 4 - 1) (tests/rotation.cu)
        PhantomCiphertext rot = make_ct(context, moduli, 1, 37, 38, s);
        rotate_inplace(context, rot, 1, gk, sw);
        printf("{\"rotate1\": \"%016llx\"}\n", (unsigned long long)fnv(download(rot)));
        PhantomCiphertext rot3 = make_ct(context, moduli, 1, 47, 48, s);
        rotate_inplace(context, rot3, 3, gk, sw);
        printf("{\"rotate3_naf\": \"%016llx\"}\n", (unsigned long long)fnv(download(rot3)));

        // element-wise entry points (tests/add.cu-style): (a + b) - b', -a, mod_switch_to_next
        {
            PhantomCiphertext a = make_ct(context, moduli, 1, 57, 58, s), b = make_ct(context, moduli, 1, 67, 68, s);
            add_inplace(context, a, b, sw);
            printf("{\"add\": \"%016llx\"}\n", (unsigned long long)fnv(download(a)));
            negate_inplace(context, a, sw);
            sub_inplace(context, a, b, sw);
            printf("{\"neg_sub\": \"%016llx\"}\n", (unsigned long long)fnv(download(a)));
            mod_switch_to_inplace(context, a, 3, sw);
            printf("{\"mod_switch_to_3\": \"%016llx\", \"ms_limbs\": %zu}\n", (unsigned long long)fnv(download(a)),
                   a.coeff_modulus_size());
        }

        // out-of-place forms with the default stream argument, like the reference's tests call them (tests/add.cu,
        // add_many.cu, negation.cu, mod_switch_to_inplace.cu): each must equal its in-place twin
        {
            check_hip(hipDeviceSynchronize(), "sync");
            PhantomCiphertext a = make_ct(context, moduli, 1, 57, 58, s), b = make_ct(context, moduli, 1, 67, 68, s);
            check_hip(hipDeviceSynchronize(), "sync");
            PhantomCiphertext sum = add(context, a, b);                 // default stream
            PhantomCiphertext a2 = a;                                   // deep copy
            add_inplace(context, a2, b);
            check_hip(hipDeviceSynchronize(), "sync");
            int ok = fnv(download(sum)) == fnv(download(a2));
            PhantomCiphertext d1 = sub(context, a, b, true);            // b - a
            PhantomCiphertext d2 = sub(context, b, a);
            check_hip(hipDeviceSynchronize(), "sync");
            ok += fnv(download(d1)) == fnv(download(d2));
            std::vector<PhantomCiphertext> many;
            many.push_back(a);
            many.push_back(b);
            many.push_back(a);
            PhantomCiphertext total, twice = add(context, sum, a);
            add_many(context, many, total);
            check_hip(hipDeviceSynchronize(), "sync");
            ok += fnv(download(total)) == fnv(download(twice));
            PhantomCiphertext n1 = negate(context, a), lower = mod_switch_to(context, a, 3);
            PhantomCiphertext a3 = a;
            negate_inplace(context, a3);
            check_hip(hipDeviceSynchronize(), "sync");
            ok += fnv(download(n1)) == fnv(download(a3)) && lower.coeff_modulus_size() == 2 && a.coeff_modulus_size() == 4;
            PhantomCiphertext r1 = rotate(context, a, 1, gk), r2 = a;
            rotate_inplace(context, r2, 1, gk);
            PhantomCiphertext m1 = multiply_and_relin(context, a, b, rlk), m2 = a;
            multiply_and_relin_inplace(context, m2, b, rlk);
            check_hip(hipDeviceSynchronize(), "sync");
            ok += fnv(download(r1)) == fnv(download(r2));
            ok += fnv(download(m1)) == fnv(download(m2)) && m1.size() == 2;
            PhantomPlaintext pl;
            pl.resize(4, n, s);
            pl.set_chain_index(1);
            PhantomPlaintext pl2 = pl;
            mod_switch_to_inplace(context, pl2, 3);
            ok += pl2.coeff_modulus_size() == 2 && pl2.chain_index() == 3 && pl.coeff_modulus_size() == 4;
            printf("{\"out_of_place_ok\": %d}\n", ok);
        }

        // real keys: keygen -> encrypt -> HMult+relin -> decrypt (tests/relin.cu), all on the GPU.
        // Plaintexts are the constant polynomials 3*2^20 and 5*2^20 (in NTT form a constant is the
        // same residue in every slot), so the product must decrypt to 15*2^40 + small noise in
        // coefficient 0 and to noise elsewhere.
        {
            PhantomSecretKey sk(context, 12345);
            PhantomRelinKey rk = sk.gen_relinkey(context, sw);
            auto constant_plain = [&](uint64_t v) {
                PhantomPlaintext p;
                p.resize(context.coeff_mod_size(1), n, s);
                std::vector<uint64_t> h(context.coeff_mod_size(1) * n, v);
                check_hip(hipMemcpy(p.data(), h.data(), h.size() * 8, hipMemcpyHostToDevice), "upload");
                p.set_chain_index(1);
                p.set_scale(1048576.0);
                return p;
            };
            PhantomPlaintext pa = constant_plain(3ULL << 20), pb = constant_plain(5ULL << 20), out;
            PhantomCiphertext ca, cb;
            sk.encrypt_symmetric(context, pa, ca, sw);
            sk.encrypt_symmetric(context, pb, cb, sw);
            multiply_and_relin_inplace(context, ca, cb, rk, sw);
            sk.decrypt(context, ca, out, sw);
            nwt_2d_radix8_backward_inplace(out.data(), context, 1, 0, s);   // limb 0 to coefficient form
            std::vector<uint64_t> h(n);
            check_hip(hipStreamSynchronize(s), "sync");
            check_hip(hipMemcpy(h.data(), out.data(), n * 8, hipMemcpyDeviceToHost), "download");
            const uint64_t q0 = moduli[0];
            auto centred = [&](uint64_t v) { return v > q0 / 2 ? -(double)(q0 - v) : (double)v; };
            double err0 = std::fabs(centred(h[0]) - 15.0 * 1099511627776.0), rest = 0;
            for (size_t i = 1; i < n; i++) rest = std::max(rest, std::fabs(centred(h[i])));
            printf("{\"real_keys_rel_err\": %.3e, \"real_keys_rest\": %.3e}\n", err0 / 1099511627776.0, rest / 1099511627776.0);
            int twice = 0;
            try { sk.gen_secretkey(context, s); } catch (const std::logic_error &) { twice = 1; }
            printf("{\"gen_twice_throws\": %d}\n", twice);
        }

        // encoder round trip through ciphertext arithmetic: (z1 + z2) * z1 slot-wise (tests/multi.cu)
        {
            PhantomSecretKey sk(context, 777);
            PhantomRelinKey rk = sk.gen_relinkey(context, sw);
            PhantomCKKSEncoder enc(context);
            std::vector<std::complex<double>> z1(n / 2), z2(n / 2), out;
            for (size_t i = 0; i < n / 2; i++) {
                z1[i] = {std::cos(0.001 * i), 0.3 * std::sin(0.002 * i)};
                z2[i] = {0.5 - 0.0001 * i, 0.25};
            }
            PhantomPlaintext p1, p2, dec;
            enc.encode(context, z1, 1073741824.0, p1, 1, sw);
            enc.encode(context, z2, 1073741824.0, p2, 1, sw);
            PhantomCiphertext c1, c2;
            sk.encrypt_symmetric(context, p1, c1, sw);
            sk.encrypt_symmetric(context, p2, c2, sw);
            add_inplace(context, c2, c1, sw);
            multiply_and_relin_inplace(context, c2, c1, rk, sw);
            sk.decrypt(context, c2, dec, sw);
            enc.decode(context, dec, out, sw);
            double err = 0;
            for (size_t i = 0; i < n / 2; i++) err = std::max(err, std::abs(out[i] - (z1[i] + z2[i]) * z1[i]));
            printf("{\"encoder_pipeline_err\": %.3e}\n", err);
            // public-key encryption (examples/*.cu: secret_key.gen_publickey + public_key.encrypt_asymmetric)
            PhantomPublicKey pk = sk.gen_publickey(context, sw);
            PhantomCiphertext ca, cb;
            pk.encrypt_asymmetric(context, p1, ca, sw);
            pk.encrypt_asymmetric(context, p1, cb, sw);
            sk.decrypt(context, ca, dec, sw);
            enc.decode(context, dec, out, sw);
            double aerr = 0;
            for (size_t i = 0; i < n / 2; i++) aerr = std::max(aerr, std::abs(out[i] - z1[i]));
            std::vector<uint64_t> ha = download(ca), hb = download(cb);
            size_t differ = 0;
            for (size_t i = 0; i < ha.size(); i++) differ += ha[i] != hb[i];
            printf("{\"asymmetric_err\": %.3e, \"asymmetric_fresh_randomness\": %.3f, \"asymmetric_limbs\": %zu}\n", aerr,
                   (double)differ / (double)ha.size(), ca.coeff_modulus_size());
        }

        // error behaviour of the reference
        int caught = 0;
        try { relinearize_inplace(context, rot, rlk, sw); } catch (const std::invalid_argument &) { caught++; }
        try { rotate_inplace(context, rot, 2, gk, sw); } catch (const std::invalid_argument &) { caught++; }
        try {
            PhantomCiphertext low = make_ct(context, moduli, 4, 1, 2, s);
            (void)rescale_to_next(context, low, sw);
        } catch (const std::invalid_argument &) { caught++; }
        try { multiply_inplace(context, r, y, sw); } catch (const std::invalid_argument &) { caught++; }
        try {
            EncryptionParameters bad = parms;
            bad.set_poly_modulus_degree(1024);       // below the SAMPLE_SIZE list (include/ntt.cuh:16-24)
            PhantomContext c2(bad);
        } catch (const std::invalid_argument &) { caught++; }
        printf("{\"errors_caught\": %d}\n", caught);
    } catch (const std::exception &e) {
        printf("{\"fatal\": \"%s\"}\n", e.what());
        return 1;
    }
    return 0;
}
