// CKKS bootstrapping on the C++ host layer (phantom_amd.hpp): the reference's CKKSEvaluator
// (include/boot/ckks_evaluator.cuh) and Bootstrapper (include/boot/Bootstrapper.cuh, src/boot/Bootstrapper.cu)
// with the constructor and the calls of examples/bootstrapping.cu —
//
//     CKKSEvaluator ckks(&context, &secret_key, &encoder, &relin_keys, &galois_keys, scale);
//     Bootstrapper boot(loge, logn, logN - 1, total_level, scale, boundary_K, deg, scale_factor, inverse_deg, &ckks);
//     boot.prepare_mod_polynomial();
//     boot.addLeftRotKeys_Linear_to_vector_3(gal_steps_vector);
//     ckks.create_galois_keys_from_steps(gal_steps_vector);
//     boot.slot_vec.push_back(logn);
//     boot.generate_LT_coefficient_3();
//     boot.bootstrap_3(rtn, cipher);
//
// The circuit is this repository's (the Python twin is phantom-fhe_amd/bootstrap.py):
// ModRaise, SubSum, CoeffToSlot and SlotToCoeff as three merged groups of special-FFT stages each
// (the "_3" = three levels per linear transform of the reference), EvalMod as a Chebyshev interpolant of
// the scaled cosine evaluated by Paterson-Stockmeyer with exact scales plus `scale_factor` double
// angles.  No NTL / Remez files: the polynomial is interpolated in double precision.  Every
// polynomial operation is a C-ABI call; this header only sequences them and keeps (level, scale).
//
// Parameter guidance (differs from the reference, whose EvalMod is tuned for q0 / scale = 2^5..2^10 with an
// arcsine stage): q0 about 2^10 x scale, EvalMod primes wide (58 bit) for full slots; chain from the top:
// 3 CoeffToSlot primes (58-60 bit), depth + scale_factor EvalMod primes, 3 SlotToCoeff primes, then the
// remaining levels; `loge`, `L` and `inverse_deg` are accepted for signature compatibility and not used.
#pragma once

#include <functional>
#include <numeric>
#include <set>
#include <tuple>

#include "phantom_amd.hpp"

namespace phantom {

using cvec = std::vector<std::complex<double>>;
using DiagMap = std::map<int, cvec>;  // rotation offset d -> diagonal v:  x -> sum_d v_d * rot(x, d)

// =====================================================================================================
// CKKSEvaluator: out-of-place operations with (chain_index, scale) bookkeeping
// =====================================================================================================
class CKKSEvaluator {
public:
    PhantomContext *context;
    PhantomSecretKey *secret_key;
    PhantomCKKSEncoder *encoder;
    PhantomRelinKey *relin_keys;
    PhantomGaloisKey *galois_keys;
    double scale;
    hip_stream_wrapper stream;
    bool auto_keys = false;  // generate a missing rotation key on first use (cnn_phantom.hpp: hundreds of steps)

private:
    std::map<std::tuple<double, double, double, size_t>, PhantomPlaintext> consts_;
    std::map<size_t, device_words> mono_;
    struct Sibling {
        std::unique_ptr<PhantomContext> context;
        std::unique_ptr<CKKSEvaluator> ev;
        hipEvent_t fork = nullptr, join = nullptr;
    };
    std::map<int, Sibling> siblings_;
    int nest_ = 0;
    const hipStream_t &s() const { return stream.get_stream(); }
    pha_context *h() const { return context->handle(); }
    static void same(const PhantomCiphertext &a, const PhantomCiphertext &b) {
        if (a.chain_index() != b.chain_index()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (std::abs(a.scale() / b.scale() - 1.0) > 1e-9) throw std::invalid_argument("scale mismatch");
    }
    PhantomCiphertext make(size_t level, double sc) {
        PhantomCiphertext c;
        c.allocate(*context, level, 2, s());
        c.set_scale(sc);
        return c;
    }

public:
    CKKSEvaluator(PhantomContext *context_, PhantomSecretKey *secret_key_, PhantomCKKSEncoder *encoder_,
                  PhantomRelinKey *relin_keys_, PhantomGaloisKey *galois_keys_, double scale_)
        : context(context_), secret_key(secret_key_), encoder(encoder_), relin_keys(relin_keys_), galois_keys(galois_keys_),
          scale(scale_) {}

    ~CKKSEvaluator() {
        for (auto &sb : siblings_) {
            if (sb.second.fork) (void)hipEventDestroy(sb.second.fork);
            if (sb.second.join) (void)hipEventDestroy(sb.second.join);
        }
    }
    // An evaluator with the same keys on its own library context (own workspace) and HIP stream.  Two independent
    // sub-circuits at 15-30 limbs each fill less than half of the GPU: side by side they take little more than one.
    // One sibling per nesting depth of run_pair, so that nested pairs form a tree of streams.
    CKKSEvaluator &sibling(int nest = 0) {
        auto it = siblings_.find(nest);
        if (it == siblings_.end()) {
            Sibling sb;
            sb.context = std::make_unique<PhantomContext>(context->key_parms());
            sb.ev = std::make_unique<CKKSEvaluator>(sb.context.get(), secret_key, encoder, relin_keys, galois_keys, scale);
            check_hip(hipEventCreateWithFlags(&sb.fork, hipEventDisableTiming), "hipEventCreate");
            check_hip(hipEventCreateWithFlags(&sb.join, hipEventDisableTiming), "hipEventCreate");
            it = siblings_.emplace(nest, std::move(sb)).first;
        }
        return *it->second.ev;
    }
    static void adopt(PhantomCiphertext &c, const hipStream_t &st) { c.touch(st); }
    static void adopt(std::vector<PhantomCiphertext> &v, const hipStream_t &st) {
        for (auto &c : v) c.touch(st);
    }
    template <class T>
    static auto adopt(T &v, const hipStream_t &st) -> decltype(v.adopt(st)) {
        return v.adopt(st);
    }
    static constexpr int kMaxNest = 2;  // run_pair inside run_pair: up to four streams
    // (fa(*this), fb(sibling)) concurrently; both results are usable on this evaluator's stream afterwards
    template <class FA, class FB>
    auto run_pair(FA &&fa, FB &&fb) -> std::pair<decltype(fa(*this)), decltype(fb(*this))> {
        if (nest_ >= kMaxNest) {
            auto a0 = fa(*this);
            auto b0 = fb(*this);
            return {std::move(a0), std::move(b0)};
        }
        const int nest = nest_;
        CKKSEvaluator &sib = sibling(nest);
        Sibling &sb = siblings_.at(nest);
        check_hip(hipEventRecord(sb.fork, s()), "hipEventRecord");
        check_hip(hipStreamWaitEvent(sib.s(), sb.fork, 0), "hipStreamWaitEvent");
        sib.nest_ = nest + 1;
        sib.auto_keys = false;
        auto b = fb(sib);
        check_hip(hipEventRecord(sb.join, sib.s()), "hipEventRecord");
        nest_ = nest + 1;
        auto a = fa(*this);
        nest_ = nest;
        check_hip(hipStreamWaitEvent(s(), sb.join, 0), "hipStreamWaitEvent");
        adopt(b, s());  // allocated on the sibling's stream, consumed (and released) on this one
        return {std::move(a), std::move(b)};
    }

    size_t n() const { return context->poly_degree(); }
    size_t limbs(size_t level) const { return context->coeff_mod_size(level); }
    double q(size_t i) const { return static_cast<double>(context->modulus(i)); }
    double q_last(size_t level) const { return q(limbs(level) - 1); }  // the prime a rescale at `level` divides by
    size_t bottom() const { return context->size_Q(); }
    void sync() const { check_hip(hipStreamSynchronize(s()), "hipStreamSynchronize"); }

    // ---- keys (ckks_evaluator.decryptor.create_galois_keys_from_steps, examples/bootstrapping.cu:145) --------
    void create_galois_keys_from_steps(const std::vector<int> &steps, bool conjugate = true) {
        std::vector<uint32_t> elts;
        const int half = static_cast<int>(n() >> 1);
        for (int st : steps) {
            if (st % half == 0) continue;
            uint32_t e;
            check(pha_galois_elt_from_step(context->log_n(), st, &e));
            elts.push_back(e);
        }
        if (conjugate) elts.push_back(static_cast<uint32_t>(2 * n() - 1));
        secret_key->add_galois_keys_from_elts(*context, elts, *galois_keys, stream);
    }

    void ensure_keys(const std::vector<int> &steps) {
        if (auto_keys) create_galois_keys_from_steps(steps, false);
    }

    // ---- encode / encrypt / decrypt --------------------------------------------------------------------
    PhantomPlaintext encode(const cvec &values, double sc, size_t level, size_t slots) {
        PhantomPlaintext p;
        encoder->encode_slots(*context, values, slots, sc, p, level, stream);
        return p;
    }
    // plaintext over Q_l u P: the operand format of the double-hoisted linear transforms
    PhantomPlaintext encode_qp(const cvec &values, double sc, size_t level, size_t slots) {
        PhantomPlaintext p;
        p.resize(context->coeff_mod_size(level) + context->size_P(), n(), s());
        check(pha_ckks_encode_qp(h(), reinterpret_cast<const double *>(values.data()), values.size(), slots, level, sc, p.data(), s()));
        p.set_chain_index(level);
        p.set_scale(sc);
        return p;
    }
    // sum_g rot_G( sum_b plains[g * n_baby + b] * rot_B(a) ), double-hoisted: ONE C-ABI call (one modup for the
    // babies, one moddown per giant step).  Steps that are multiples of N/2 do not rotate; plains entries may be null.
    PhantomCiphertext bsgs(const PhantomCiphertext &a, const std::vector<int> &baby_steps, const std::vector<int> &giant_steps,
                           const std::vector<const PhantomPlaintext *> &plains) {
        const int half = static_cast<int>(n() >> 1);
        ensure_keys(baby_steps);
        ensure_keys(giant_steps);
        std::vector<uint32_t> be, ge;
        std::vector<const uint64_t *const *> bk, gk;
        auto fill = [&](const std::vector<int> &steps, std::vector<uint32_t> &elts, std::vector<const uint64_t *const *> &keys) {
            for (int st : steps) {
                const uint32_t e = st % half ? elt(st) : 1u;
                elts.push_back(e);
                keys.push_back(e == 1 ? nullptr : galois_keys->get_relin_keys(e).public_keys_ptr());
            }
        };
        fill(baby_steps, be, bk);
        fill(giant_steps, ge, gk);
        std::vector<const uint64_t *> pp;
        double pscale = 0;
        for (const PhantomPlaintext *p : plains) {
            pp.push_back(p ? p->data() : nullptr);
            if (p) pscale = p->scale();
        }
        PhantomCiphertext d = make(a.chain_index(), a.scale() * pscale);
        check(pha_linear_transform_double_hoisted(h(), a.chain_index(), a.data(), be.size(), be.data(), bk.data(), ge.size(), ge.data(),
                                                  gk.data(), pp.data(), d.data(), s()));
        return d;
    }
    PhantomCiphertext encrypt(const PhantomPlaintext &p) {
        PhantomCiphertext c;
        secret_key->encrypt_symmetric(*context, p, c, stream);
        c.set_chain_index(p.chain_index());
        return c;
    }
    cvec decrypt_decode(const PhantomCiphertext &c, size_t slots) {
        PhantomPlaintext p;
        secret_key->decrypt(*context, c, p, stream);
        cvec out(slots);
        check(pha_ckks_decode(h(), p.data(), p.chain_index(), p.scale(), slots, reinterpret_cast<double *>(out.data()), s()));
        return out;
    }

    // ---- arithmetic -----------------------------------------------------------------------------------
    PhantomCiphertext add(const PhantomCiphertext &a, const PhantomCiphertext &b) {
        same(a, b);
        PhantomCiphertext d = make(a.chain_index(), a.scale());
        check(pha_add_rns_poly(h(), d.data(), a.data(), b.data(), a.coeff_modulus_size(), 2, 2, s()));
        return d;
    }
    void add_inplace(PhantomCiphertext &a, const PhantomCiphertext &b) {
        same(a, b);
        check(pha_add_rns_poly(h(), a.data(), a.data(), b.data(), a.coeff_modulus_size(), 2, 2, s()));
    }
    PhantomCiphertext sub(const PhantomCiphertext &a, const PhantomCiphertext &b) {
        same(a, b);
        PhantomCiphertext d = make(a.chain_index(), a.scale());
        check(pha_sub_rns_poly(h(), d.data(), a.data(), b.data(), a.coeff_modulus_size(), 2, 2, s()));
        return d;
    }
    PhantomCiphertext neg(const PhantomCiphertext &a) {
        PhantomCiphertext d = make(a.chain_index(), a.scale());
        check(pha_negate_rns_poly(h(), d.data(), a.data(), a.coeff_modulus_size(), 2, s()));
        return d;
    }
    PhantomCiphertext add_plain(const PhantomCiphertext &a, const PhantomPlaintext &p) {
        if (a.chain_index() != p.chain_index() || std::abs(a.scale() / p.scale() - 1.0) > 1e-9)
            throw std::invalid_argument("encrypted and plain parameter mismatch");
        PhantomCiphertext d = a.clone(s());
        check(pha_add_rns_poly(h(), d.data(), d.data(), p.data(), d.coeff_modulus_size(), 1, 1, s()));
        return d;
    }
    PhantomCiphertext mul_plain(const PhantomCiphertext &a, const PhantomPlaintext &p) {
        if (a.chain_index() != p.chain_index()) throw std::invalid_argument("encrypted and plain parameter mismatch");
        PhantomCiphertext d = make(a.chain_index(), a.scale() * p.scale());
        check(pha_multiply_rns_poly(h(), d.data(), a.data(), p.data(), a.coeff_modulus_size(), 2, 1, s()));
        return d;
    }
    // acc += a * p in one pass (multiply_and_add_rns_poly); an empty acc is initialised with a * p
    void mul_plain_add(PhantomCiphertext &acc, const PhantomCiphertext &a, const PhantomPlaintext &p) {
        if (acc.data() == nullptr) {
            acc = mul_plain(a, p);
            return;
        }
        if (a.chain_index() != p.chain_index() || a.chain_index() != acc.chain_index() ||
            std::abs(acc.scale() / (a.scale() * p.scale()) - 1.0) > 1e-9)
            throw std::invalid_argument("multiply-accumulate parameter mismatch");
        check(pha_multiply_add_rns_poly(h(), acc.data(), a.data(), p.data(), acc.data(), a.coeff_modulus_size(), 2, 1, s()));
    }
    // HMult + relinearise (multiply_inplace + relinearize_inplace of the reference's evaluator)
    PhantomCiphertext mul(const PhantomCiphertext &a, const PhantomCiphertext &b) {
        if (a.chain_index() != b.chain_index()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        PhantomCiphertext d;
        d.allocate(*context, a.chain_index(), 3, s());
        const size_t words = 2 * a.coeff_modulus_size() * n();
        check_hip(hipMemcpyAsync(d.data(), a.data(), words * sizeof(uint64_t), hipMemcpyDeviceToDevice, s()), "hipMemcpyAsync");
        check(pha_multiply_and_relin_inplace(h(), a.chain_index(), d.data(), b.data(), relin_keys->public_keys_ptr(), s()));
        d.truncate(2);
        d.set_scale(a.scale() * b.scale());
        return d;
    }
    // rescale(relinearize(a * b)) (x 2) as ONE composite: pha_multiply_relin_rescale divides the key-switched part and
    // the lifted tensor polys by P * q_last in a single, rounded-to-nearest moddown
    PhantomCiphertext mul_rescale(const PhantomCiphertext &a, const PhantomCiphertext &b, bool times2 = false) {
        if (a.chain_index() != b.chain_index()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        PhantomCiphertext d = make(a.chain_index() + 1, a.scale() * b.scale() / q_last(a.chain_index()));
        check(pha_multiply_relin_rescale(h(), a.chain_index(), a.data(), b.data(), relin_keys->public_keys_ptr(), d.data(),
                                         times2 ? 1 : 0, s()));
        return d;
    }
    PhantomCiphertext rescale(const PhantomCiphertext &a) {
        PhantomCiphertext d = make(a.chain_index() + 1, a.scale() / q_last(a.chain_index()));
        check(pha_rescale_to_next(h(), a.chain_index(), a.data(), 2, d.data(), s()));
        return d;
    }
    // CKKS mod_switch_to: drop limbs, no scaling (src/evaluate.cu:618-700) — one strided copy per poly
    PhantomCiphertext drop(const PhantomCiphertext &a, size_t level) {
        if (level < a.chain_index()) throw std::invalid_argument("cannot switch to higher level modulus");
        if (level == a.chain_index()) return a.clone(s());
        PhantomCiphertext d = make(level, a.scale());
        const size_t src = a.coeff_modulus_size() * n(), dst = d.coeff_modulus_size() * n();
        check_hip(hipMemcpy2DAsync(d.data(), dst * sizeof(uint64_t), a.data(), src * sizeof(uint64_t), dst * sizeof(uint64_t), 2,
                                   hipMemcpyDeviceToDevice, s()), "hipMemcpy2DAsync");
        return d;
    }
    // a at `level` without a copy when it is already there
    const PhantomCiphertext &at_level(const PhantomCiphertext &a, size_t level, PhantomCiphertext &tmp) {
        if (level == a.chain_index()) return a;
        tmp = drop(a, level);
        return tmp;
    }

    // ---- constants (a constant vector is the constant polynomial: two slots are enough; cached) --------------
    const PhantomPlaintext &const_pt(std::complex<double> c, double sc, size_t level) {
        const auto key = std::make_tuple(c.real(), c.imag(), sc, level);
        auto it = consts_.find(key);
        if (it == consts_.end()) it = consts_.emplace(key, encode(cvec{c, c}, sc, level, 2)).first;
        return it->second;
    }
    PhantomCiphertext add_const(const PhantomCiphertext &a, double c) { return add_plain(a, const_pt(c, a.scale(), a.chain_index())); }
    PhantomCiphertext mul_const(const PhantomCiphertext &a, double c, double const_scale) {
        return mul_plain(a, const_pt(c, const_scale, a.chain_index()));
    }
    PhantomCiphertext dbl(const PhantomCiphertext &a) { return add(a, a); }
    // exact (level, scale): multiply by the constant 1 encoded at the scale that makes the rescaled product land on it
    PhantomCiphertext to_level_scale(const PhantomCiphertext &a, size_t level, double sc) {
        if (a.chain_index() >= level) throw std::invalid_argument("to_level_scale needs a level to spend");
        PhantomCiphertext tmp;
        const PhantomCiphertext &x = at_level(a, level - 1, tmp);
        return rescale(mul_const(x, 1.0, sc * q_last(level - 1) / x.scale()));
    }
    // 2a^2 - 1: a Chebyshev doubling / a cosine double angle, one level
    PhantomCiphertext square_double_minus_one(const PhantomCiphertext &a) { return add_const(mul_rescale(a, a, true), -1.0); }

    // ---- rotations ---------------------------------------------------------------------------------------
    uint32_t elt(int step) const {
        uint32_t e;
        check(pha_galois_elt_from_step(context->log_n(), step, &e));
        return e;
    }
    // all rotations of one ciphertext share its modup (pha_rotate_hoisted); steps that are multiples of N/2 copy
    std::vector<PhantomCiphertext> rotate_many(const PhantomCiphertext &a, const std::vector<int> &steps) {
        const int half = static_cast<int>(n() >> 1);
        ensure_keys(steps);
        std::vector<PhantomCiphertext> out(steps.size());
        std::vector<uint32_t> elts;
        std::vector<const uint64_t *const *> evks;
        std::vector<uint64_t *> dst;
        for (size_t i = 0; i < steps.size(); i++) {
            if (steps[i] % half == 0) {
                out[i] = a.clone(s());
                continue;
            }
            out[i] = make(a.chain_index(), a.scale());
            elts.push_back(elt(steps[i]));
            evks.push_back(galois_keys->get_relin_keys(elts.back()).public_keys_ptr());
            dst.push_back(out[i].data());
        }
        if (!elts.empty())
            check(pha_rotate_hoisted(h(), a.chain_index(), a.data(), elts.size(), elts.data(), evks.data(), dst.data(), s()));
        return out;
    }
    PhantomCiphertext rotate(const PhantomCiphertext &a, int step) {
        PhantomCiphertext d = a.clone(s());
        if (step % static_cast<int>(n() >> 1) == 0) return d;
        ensure_keys({step});
        const uint32_t e = elt(step);
        check(pha_apply_galois_inplace(h(), d.chain_index(), d.data(), e, galois_keys->get_relin_keys(e).public_keys_ptr(), s()));
        return d;
    }
    PhantomCiphertext conj(const PhantomCiphertext &a) {
        PhantomCiphertext d = a.clone(s());
        const uint32_t e = static_cast<uint32_t>(2 * n() - 1);
        check(pha_apply_galois_inplace(h(), d.chain_index(), d.data(), e, galois_keys->get_relin_keys(e).public_keys_ptr(), s()));
        return d;
    }
    // multiplication of every slot by i = multiplication of the polynomial by X^(N/2): exact
    PhantomCiphertext mul_i(const PhantomCiphertext &a) {
        const size_t l = a.coeff_modulus_size();
        auto it = mono_.find(l);
        if (it == mono_.end()) {
            std::vector<uint64_t> host(l * n(), 0);
            for (size_t j = 0; j < l; j++) host[j * n() + n() / 2] = 1;
            device_words m(l * n());
            check_hip(hipMemcpy(m.get(), host.data(), host.size() * sizeof(uint64_t), hipMemcpyHostToDevice), "hipMemcpy");
            check(pha_nwt_2d_radix8_forward_inplace(h(), m.get(), l, 0, s()));
            it = mono_.emplace(l, std::move(m)).first;
        }
        PhantomCiphertext d = make(a.chain_index(), a.scale());
        check(pha_multiply_rns_poly(h(), d.data(), a.data(), it->second.get(), l, 2, 1, s()));
        return d;
    }
    // modraise_inplace (src/boot/Bootstrapper.cu:2814-2868): one-limb ciphertext -> top level, t = m + q0 * I
    PhantomCiphertext mod_raise(const PhantomCiphertext &a) {
        if (a.chain_index() != bottom()) throw std::invalid_argument("mod_raise expects a ciphertext at the last level");
        PhantomCiphertext d = make(1, a.scale());
        check(pha_mod_raise(h(), a.data(), d.data(), s()));
        return d;
    }
};

// =====================================================================================================
// Linear maps as sparse sets of generalised diagonals; the special FFT as a product of such maps
// =====================================================================================================
inline DiagMap compose_diagonals(const DiagMap &outer, const DiagMap &inner, int S) {
    // v1 * rot(v2 * rot(x, d2), d1) = v1 * rot(v2, d1) * rot(x, d1 + d2)
    DiagMap out;
    for (const auto &o : outer)
        for (const auto &i : inner) {
            const int d = (o.first + i.first) % S;
            cvec &acc = out[d];
            if (acc.empty()) acc.assign(S, 0.0);
            for (int j = 0; j < S; j++) acc[j] += o.second[j] * i.second[(j + o.first) % S];
        }
    for (auto it = out.begin(); it != out.end();) {
        double m = 0;
        for (const auto &v : it->second) m = std::max(m, std::abs(v));
        it = m > 1e-300 ? std::next(it) : out.erase(it);
    }
    return out;
}

// one radix-2 stage of the special FFT over the rotation group (the butterflies of src/fft.cu, twiddle
// exp(2 pi i (5^j mod 4 len) / (4 len))): forward (u, v) -> (u + w v, u - w v); inverse (a, b) -> ((a + b)/2, (a - b) conj(w)/2)
inline DiagMap special_fft_stage(int S, int length, bool inverse) {
    const int lenh = length / 2, lenq = length * 4;
    std::vector<std::complex<double>> tw(lenh);
    long long p = 1;
    const double kPi = 3.14159265358979323846;
    for (int t = 0; t < lenh; t++) {
        tw[t] = std::polar(1.0, 2.0 * kPi * static_cast<double>(p) / lenq);
        p = p * 5 % lenq;
    }
    DiagMap out;
    auto put = [&](int d, int j, std::complex<double> v) {
        cvec &a = out[((d % S) + S) % S];
        if (a.empty()) a.assign(S, 0.0);
        a[j] += v;
    };
    for (int j = 0; j < S; j++) {
        const int r = j % length;
        const bool first = r < lenh;
        const std::complex<double> w = tw[r % lenh];
        if (!inverse) {
            put(0, j, first ? 1.0 : -w);
            if (first) put(lenh, j, w);
            else put(S - lenh, j, 1.0);
        } else {
            put(0, j, first ? std::complex<double>(0.5) : -0.5 * std::conj(w));
            if (first) put(lenh, j, 0.5);
            else put(S - lenh, j, 0.5 * std::conj(w));
        }
    }
    return out;
}

// the log2(S) stages merged into n_groups maps, in application order: forward len = 2, 4, .., S on bit-reversed input;
// inverse len = S, .., 2 leaving bit-reversed output
inline std::vector<DiagMap> special_fft_groups(int S, int n_groups, bool inverse) {
    std::vector<int> lengths;
    for (int l = 2; l <= S; l <<= 1) lengths.push_back(l);
    if (inverse) std::reverse(lengths.begin(), lengths.end());
    const int total = static_cast<int>(lengths.size());
    n_groups = std::max(1, std::min(n_groups, total));
    std::vector<DiagMap> groups;
    int pos = 0;
    for (int i = 0; i < n_groups; i++) {
        const int cnt = total / n_groups + (i < total % n_groups ? 1 : 0);
        DiagMap m;
        m[0] = cvec(S, 1.0);
        for (int k = pos; k < pos + cnt; k++) m = compose_diagonals(special_fft_stage(S, lengths[k], inverse), m, S);
        pos += cnt;
        if (!m.empty()) groups.push_back(std::move(m));
    }
    return groups;
}

// x -> sum_d v_d * rot(x, d) on S slots: baby-step / giant-step over the offsets that occur (multiples of one
// stride), evaluated double-hoisted by one library call; one level
class DiagonalTransform {
    CKKSEvaluator *ev_;
    int S_, stride_ = 1, n1_ = 1;
    size_t level_;
    DiagMap diagonals_;
    std::vector<int> babies_, giants_;          // baby / giant indices that occur (units of stride, of n1 * stride)
    std::vector<PhantomPlaintext> plains_;      // encoded diagonals over Q_l u P
    std::vector<const PhantomPlaintext *> table_;  // [giant][baby] -> plaintext or null

public:
    DiagonalTransform(CKKSEvaluator *ev, DiagMap diagonals, int S, size_t level) : ev_(ev), S_(S), level_(level), diagonals_(std::move(diagonals)) {
        int stride = S;
        for (const auto &d : diagonals_)
            if (d.first) stride = std::gcd(stride, d.first);
        stride_ = stride;
        const int m = S / stride;
        const double cnt = std::max<double>(2.0, static_cast<double>(diagonals_.size()));
        // a baby step (one inner product) is several times cheaper than a giant step (moddown + modup + inner
        // product): one more bit of baby steps than the square-root split (measured: 31.3 -> 30.5 ms at N = 2^16)
        n1_ = std::max(1, 1 << static_cast<int>(std::ceil((std::log2(cnt) + 1.0) / 2.0)));
        n1_ = std::min(n1_, m);
        std::set<int> babies, giants;
        for (const auto &d : diagonals_) {
            babies.insert((d.first / stride_) % n1_);
            giants.insert((d.first / stride_) / n1_);
        }
        babies_.assign(babies.begin(), babies.end());
        giants_.assign(giants.begin(), giants.end());
    }
    size_t level() const { return level_; }
    std::vector<int> steps() const {
        std::vector<int> out;
        for (int b : babies_)
            if (b) out.push_back(b * stride_);
        for (int g : giants_)
            if (g) out.push_back(g * n1_ * stride_);
        return out;
    }
    // encode the diagonals (rotated back by their giant step) at diag_scale, over Q_l u P
    void encode(double diag_scale) {
        plains_.clear();
        plains_.reserve(diagonals_.size());
        table_.assign(giants_.size() * babies_.size(), nullptr);
        for (const auto &d : diagonals_) {
            const int k = d.first / stride_, g = k / n1_, b = k % n1_, shift = g * n1_ * stride_;
            cvec rolled(S_);
            for (int j = 0; j < S_; j++) rolled[j] = d.second[((j - shift) % S_ + S_) % S_];
            plains_.push_back(ev_->encode_qp(rolled, diag_scale, level_, static_cast<size_t>(S_)));
            const size_t gi = std::lower_bound(giants_.begin(), giants_.end(), g) - giants_.begin();
            const size_t bi = std::lower_bound(babies_.begin(), babies_.end(), b) - babies_.begin();
            table_[gi * babies_.size() + bi] = &plains_.back();
        }
    }
    PhantomCiphertext apply(const PhantomCiphertext &in) {
        PhantomCiphertext tmp;
        const PhantomCiphertext &ct = ev_->at_level(in, level_, tmp);
        std::vector<int> bs, gs;
        for (int b : babies_) bs.push_back(b * stride_);
        for (int g : giants_) gs.push_back(g * n1_ * stride_);
        return ev_->rescale(ev_->bsgs(ct, bs, gs, table_));
    }
};

// =====================================================================================================
// Chebyshev series on [-1, 1] by baby-step / giant-step (Paterson-Stockmeyer) with exact target scales
// =====================================================================================================
class ChebyshevEvaluator {
    CKKSEvaluator *ev_;
    std::vector<double> c_;
    int m_, l_;
    bool optimal_ = false;  // depth ceil(log2(d+1)): a leaf c1 T1 + c2 T2 + c3 T3 is (c1 - c3) T1 + c2 T2 + (2 c3 T1) T2
    std::map<int, PhantomCiphertext> T_;
    const std::map<int, PhantomCiphertext> *Tp_ = nullptr;  // a twin on a sibling evaluator reads its parent's powers
    std::map<std::pair<int, size_t>, PhantomCiphertext> dropped_;
    const std::map<int, PhantomCiphertext> &powers_ref() const { return Tp_ ? *Tp_ : T_; }
    // T_j at `level` (dropping limbs copies the ciphertext: once per level)
    const PhantomCiphertext &at(int j, size_t level) {
        const PhantomCiphertext &t = powers_ref().at(j);
        if (t.chain_index() == level) return t;
        auto key = std::make_pair(j, level);
        auto it = dropped_.find(key);
        if (it == dropped_.end()) it = dropped_.emplace(key, ev_->drop(t, level)).first;
        return it->second;
    }

    struct Val {
        enum Kind { NONE, CONST, CT } kind = NONE;
        double c = 0;
        PhantomCiphertext ct;
        void adopt(const hipStream_t &st) {
            if (kind == CT) ct.touch(st);
        }
    };
    static void trim(std::vector<double> &c) {
        while (!c.empty() && c.back() == 0.0) c.pop_back();
    }
    // T_{a+b} = 2 T_a T_b - T_{|a-b|}
    PhantomCiphertext product(int a, int b) {
        const PhantomCiphertext &x = T_.at(a), &y = T_.at(b);
        const size_t lvl = std::max(x.chain_index(), y.chain_index());
        PhantomCiphertext tx, ty;
        PhantomCiphertext p = ev_->mul_rescale(ev_->at_level(x, lvl, tx), ev_->at_level(y, lvl, ty), true);
        if (a == b) return ev_->add_const(p, -1.0);
        return ev_->sub(p, ev_->to_level_scale(T_.at(std::abs(a - b)), p.chain_index(), p.scale()));
    }
    void powers(const PhantomCiphertext &w) {
        T_.clear();
        dropped_.clear();
        T_.emplace(1, w.clone(ev_->stream.get_stream()));
        for (int j = 2; j < (optimal_ ? 3 : (1 << l_)); j++) {  // babies T_2 .. T_{2^l - 1} (T_3 is never formed at optimal depth)
            int a = 1;
            while (2 * a <= j) a <<= 1;
            T_.emplace(j, j == a ? product(j / 2, j / 2) : product(a, j - a));
        }
        for (int k = l_; k < m_; k++) {  // giants T_{2^l} .. T_{2^(m-1)}
            const int hh = 1 << k;
            if (!T_.count(hh)) T_.emplace(hh, product(hh / 2, hh / 2));
        }
    }
    // sum_j c_j T_j, j < 2^l, landing exactly on (level, scale)
    Val leaf(const std::vector<double> &c, size_t level, double scale) {
        Val out;
        const size_t lvl = level - 1;
        const double ql = ev_->q_last(lvl);
        PhantomCiphertext acc;
        std::vector<double> cc = c;
        if (optimal_ && cc.size() > 3 && std::abs(cc[3]) >= 1e-300) {
            // (2 c3 T1) T2; T3 = 2 T1 T2 - T1 puts -c3 on T1
            const PhantomCiphertext &t1 = at(1, lvl - 1), &t2 = at(2, lvl);
            PhantomCiphertext a = ev_->rescale(ev_->mul_const(t1, 2.0 * cc[3], (scale * ql / t2.scale()) * ev_->q_last(lvl - 1) / t1.scale()));
            const double kappa = (cc[1] - cc[3]) / (2.0 * cc[3]);
            if (std::abs(cc[2]) < 1e-300 && std::abs(kappa) <= 16.0) {
                // an odd leaf: (2 c3 T1)(T2 + kappa) is the whole of it, one merged product
                out.kind = Val::CT;
                out.ct = ev_->mul_rescale(a, ev_->add_const(t2, kappa));
                if (std::abs(cc[0]) >= 1e-300) out.ct = ev_->add_const(out.ct, cc[0]);
                return out;
            }
            acc = ev_->mul(a, t2);  // unrescaled at (lvl, scale * ql)
            cc = {cc[0], cc[1] - cc[3], cc[2]};
        }
        for (size_t j = 1; j < cc.size(); j++) {
            if (std::abs(cc[j]) < 1e-300) continue;
            const PhantomCiphertext &t = at(static_cast<int>(j), lvl);
            ev_->mul_plain_add(acc, t, ev_->const_pt(cc[j], scale * ql / t.scale(), lvl));
        }
        if (acc.data() == nullptr) {
            if (std::abs(cc[0]) >= 1e-300) out.kind = Val::CONST, out.c = cc[0];
            return out;
        }
        if (std::abs(cc[0]) >= 1e-300) acc = ev_->add_const(acc, cc[0]);
        out.kind = Val::CT;
        out.ct = ev_->rescale(acc);
        return out;
    }
    Val eval(std::vector<double> c, int k, size_t level, double scale) {
        trim(c);
        if (c.empty()) return Val{};
        const size_t nb = size_t(1) << l_;
        if (k <= l_ || c.size() <= nb) {
            if (c.size() > nb) c.resize(nb);
            return leaf(c, level, scale);
        }
        const size_t hh = size_t(1) << (k - 1);
        if (c.size() <= hh) return eval(std::move(c), k - 1, level, scale);
        // p = q * T_h + r
        std::vector<double> qc(c.size() - hh, 0.0), r(c.begin(), c.begin() + hh);
        qc[0] = c[hh];
        for (size_t j = hh + 1; j < c.size(); j++) {
            qc[j - hh] = 2.0 * c[j];
            r[2 * hh - j] -= c[j];
        }
        const PhantomCiphertext &Th = at(static_cast<int>(hh), level - 1);
        const double qscale = scale * ev_->q_last(level - 1) / Th.scale();
        Val qv, rv;
        if (parallel && k >= m_ - 1 && k > l_) {
            // two independent halves at 5-20 limbs each: side by side on the sibling (context, stream) pair
            auto pr = ev_->run_pair([&](CKKSEvaluator &) { return eval(r, k - 1, level, scale); },
                                    [&](CKKSEvaluator &e2) {
                                        ChebyshevEvaluator twin(&e2, *this);
                                        return twin.eval(qc, k - 1, level - 1, qscale);
                                    });
            rv = std::move(pr.first), qv = std::move(pr.second);
        } else {
            qv = eval(std::move(qc), k - 1, level - 1, qscale);
            rv = eval(std::move(r), k - 1, level, scale);
        }
        Val prod;
        if (qv.kind == Val::CONST) {
            prod.kind = Val::CT;
            prod.ct = ev_->rescale(ev_->mul_const(Th, qv.c, qscale));
        } else if (qv.kind == Val::CT) {
            prod.kind = Val::CT;
            prod.ct = ev_->mul_rescale(qv.ct, Th);
        }
        if (prod.kind == Val::NONE) return rv;
        if (rv.kind == Val::NONE) return prod;
        if (rv.kind == Val::CONST)
            prod.ct = ev_->add_const(prod.ct, rv.c);
        else
            ev_->add_inplace(prod.ct, rv.ct);
        return prod;
    }

public:
    bool parallel = false;  // the two halves of the top splits (p = q T_h + r) through ev.run_pair
    ChebyshevEvaluator() : ev_(nullptr), m_(1), l_(1) {}
    // the same polynomial on another evaluator, reading `parent`'s powers
    ChebyshevEvaluator(CKKSEvaluator *ev, const ChebyshevEvaluator &parent)
        : ev_(ev), c_(parent.c_), m_(parent.m_), l_(parent.l_), optimal_(parent.optimal_), Tp_(&parent.powers_ref()), parallel(parent.parallel) {}
    ChebyshevEvaluator(CKKSEvaluator *ev, std::vector<double> coeffs, int baby_log = 2, bool optimal_depth = false)
        : ev_(ev), c_(std::move(coeffs)), optimal_(optimal_depth) {
        const double d = static_cast<double>(c_.size()) - 1.0;
        m_ = std::max(optimal_ ? 2 : 1, static_cast<int>(std::ceil(std::log2(d + 1.0))));
        l_ = std::min(baby_log, m_);
        if (optimal_ && l_ != 2) throw std::invalid_argument("the optimal-depth leaves need baby_log = 2");
    }
    int depth() const { return optimal_ ? m_ : m_ + 1; }
    // Chebyshev interpolant of f at the degree + 1 Chebyshev points of the first kind
    static std::vector<double> interpolate(const std::function<double(double)> &f, int degree) {
        const int nn = degree + 1;
        const double kPi = 3.14159265358979323846;
        std::vector<double> fx(nn), c(nn);
        for (int k = 0; k < nn; k++) fx[k] = f(std::cos(kPi * (k + 0.5) / nn));
        for (int j = 0; j < nn; j++) {
            double acc = 0;
            for (int k = 0; k < nn; k++) acc += fx[k] * std::cos(kPi * j * (k + 0.5) / nn);
            c[j] = acc * (j == 0 ? 1.0 : 2.0) / nn;
        }
        return c;
    }
    // p(w) at level w.level + depth(), scale exactly target_scale
    PhantomCiphertext operator()(const PhantomCiphertext &w, double target_scale) {
        powers(w);
        Val v = eval(c_, m_, w.chain_index() + depth(), target_scale);
        T_.clear();
        dropped_.clear();
        if (v.kind != Val::CT) throw std::logic_error("the polynomial is constant");
        return std::move(v.ct);
    }
};

// =====================================================================================================
// Bootstrapper
// =====================================================================================================
class Bootstrapper {
public:
    long loge, logn, n, logNh, Nh, L;
    double initial_scale, final_scale;
    long boundary_K, sin_cos_deg, scale_factor, inverse_deg;
    CKKSEvaluator *ckks = nullptr;
    std::vector<long> slot_vec;
    int lt_levels = 3;        // groups per linear transform (the "_3" of the reference's API)
    int s2c_lt_levels = 0;    // groups of SlotToCoeff when they differ (0: lt_levels)
    // slim order (Chen, Chillotti, Song 2019): SlotToCoeff FIRST, on the last levels of the chain (a few limbs), then
    // modulus raise, CoeffToSlot, EvalMod, whose result already holds the message in the slots.  The factor
    // q0 / (2 pi scale) that the usual order folds into the SlotToCoeff diagonals is a re-declaration of the scale, and
    // the scale on which the Chebyshev polynomial lands is chosen so that the output scale equals the input scale.
    bool slim = false;
    bool concurrent_evalmod = true;  // real and imaginary EvalMod on two (context, stream) pairs
    double evalmod_scale = 0;  // 0: the power of two nearest to the first EvalMod prime

private:
    ChebyshevEvaluator cheb_, cheb_sibling_;
    std::vector<double> cheb_coeffs_;
    std::vector<DiagonalTransform> c2s_, s2c_;
    size_t evalmod_level_ = 0, s2c_level_ = 0, out_level_ = 0;
    double delta_ = 0, delta_out_ = 0, Kb_ = 0;
    bool prepared_ = false, generated_ = false;

    int groups(bool s2c = false) const {
        return std::max(1, std::min<int>(s2c && s2c_lt_levels > 0 ? s2c_lt_levels : lt_levels, static_cast<int>(logn)));
    }
    size_t slots() const { return size_t(1) << logn; }
    void plan_levels() {
        evalmod_level_ = 1 + groups();
        const size_t after_evalmod = evalmod_level_ + cheb_.depth() + scale_factor;
        if (slim) {
            out_level_ = after_evalmod;
            if (out_level_ + groups(true) > ckks->bottom()) throw std::invalid_argument("the modulus chain is too short for bootstrapping");
            s2c_level_ = ckks->bottom() - groups(true);
        } else {
            s2c_level_ = after_evalmod;
            out_level_ = s2c_level_ + groups(true);
            if (out_level_ > ckks->bottom()) throw std::invalid_argument("the modulus chain is too short for bootstrapping");
        }
        delta_ = evalmod_scale > 0 ? evalmod_scale : std::exp2(std::round(std::log2(ckks->q_last(evalmod_level_))));
        delta_out_ = delta_;
        if (slim) {  // delta_out^(2^r) / prod q_i^(2^(r-i)) = q0 / (2 pi) over the primes of the r double-angle levels
            const double kPi = 3.14159265358979323846;
            double lg = std::log2(ckks->q(0) / (2.0 * kPi));
            const size_t lvl0 = evalmod_level_ + cheb_.depth();
            for (long i = 0; i < scale_factor; i++) lg += std::exp2(static_cast<double>(scale_factor - 1 - i)) * std::log2(ckks->q_last(lvl0 + i));
            delta_out_ = std::exp2(lg / std::exp2(static_cast<double>(scale_factor)));
        }
    }
    size_t pre_levels_() const { return slim ? static_cast<size_t>(groups(true)) : 0; }
    void build_transforms() {
        if (!c2s_.empty()) return;
        const int S = static_cast<int>(slots());
        size_t lvl = 1;
        for (auto &g : special_fft_groups(S, groups(), true)) c2s_.emplace_back(ckks, std::move(g), S, lvl++);
        lvl = s2c_level_;
        for (auto &g : special_fft_groups(S, groups(true), false)) s2c_.emplace_back(ckks, std::move(g), S, lvl++);
    }
    PhantomCiphertext eval_mod(CKKSEvaluator &ev, ChebyshevEvaluator &cheb, const PhantomCiphertext &w) {
        PhantomCiphertext c = cheb(w, delta_out_);
        for (long i = 0; i < scale_factor; i++) c = ev.square_double_minus_one(c);
        return c;
    }

public:
    Bootstrapper(long _loge, long _logn, long _logNh, long _L, double _final_scale, long _boundary_K, long _sin_cos_deg,
                 long _scale_factor, long _inverse_deg, CKKSEvaluator *_ckks)
        : loge(_loge), logn(_logn), n(1L << _logn), logNh(_logNh), Nh(1L << _logNh), L(_L), initial_scale(_final_scale),
          final_scale(_final_scale), boundary_K(_boundary_K), sin_cos_deg(_sin_cos_deg), scale_factor(_scale_factor),
          inverse_deg(_inverse_deg), ckks(_ckks) {
        if (logn < 1 || logn > logNh) throw std::invalid_argument("logn out of range");
    }
    void set_final_scale(double _final_scale) { final_scale = _final_scale; }
    size_t levels_needed() const { return out_level_; }
    size_t pre_levels() const { return pre_levels_(); }  // levels spent before the modulus raise (slim order)

    // the approximate polynomial: Chebyshev interpolant of cos(2 pi (Kb w - 1/4) / 2^r) on [-1, 1], Kb = K + 1/2
    void prepare_mod_polynomial() {
        Kb_ = static_cast<double>(boundary_K) + 0.5;
        const double Kb = Kb_, rr = std::exp2(static_cast<double>(scale_factor)), kPi = 3.14159265358979323846;
        cheb_coeffs_ = ChebyshevEvaluator::interpolate([=](double w) { return std::cos(2.0 * kPi * (Kb * w - 0.25) / rr); },
                                                       static_cast<int>(sin_cos_deg));
        cheb_ = ChebyshevEvaluator(ckks, cheb_coeffs_);
        plan_levels();
        prepared_ = true;
    }
    // rotation steps of SubSum and of the two linear transforms
    void addLeftRotKeys_Linear_to_vector_3(std::vector<int> &gal_steps_vector) {
        if (!prepared_) prepare_mod_polynomial();
        build_transforms();
        std::set<int> have(gal_steps_vector.begin(), gal_steps_vector.end());
        auto add = [&](int st) {
            if (have.insert(st).second) gal_steps_vector.push_back(st);
        };
        for (auto &t : c2s_)
            for (int st : t.steps()) add(st);
        for (auto &t : s2c_)
            for (int st : t.steps()) add(st);
        for (long k = n; k < Nh; k <<= 1) add(static_cast<int>(k));
    }
    // encode the diagonals of both transforms at the scales that keep the bookkeeping exact
    void generate_LT_coefficient_3() {
        if (!prepared_) prepare_mod_polynomial();
        build_transforms();
        const double q0 = ckks->q(0), kPi = 3.14159265358979323846;
        const double g = static_cast<double>(Nh) / static_cast<double>(n);
        const double norm = 1.0 / (g * Kb_ * 2.0);
        // CoeffToSlot: intermediate results keep the re-declared scale q0, the last group lands on the EvalMod scale
        const size_t n_c2s = c2s_.size(), n_s2c = s2c_.size();
        const double c2s_cst = std::pow(norm, 1.0 / n_c2s);
        rebuild(c2s_, true, c2s_cst, [&](size_t i, size_t lvl) { return ckks->q_last(lvl) * (i + 1 == n_c2s ? delta_ / q0 : 1.0); });
        // EvalMod output scale: delta after the polynomial, squared and rescaled by every double angle
        double sv = delta_out_;
        size_t lvl = evalmod_level_ + cheb_.depth();
        for (long i = 0; i < scale_factor; i++, lvl++) sv = sv * sv / ckks->q_last(lvl);
        if (slim) {  // the forward transform as it is, scale-preserving
            rebuild(s2c_, false, 1.0, [&](size_t, size_t l2) { return ckks->q_last(l2); });
        } else {
            const double s2c_cst = std::pow((q0 / final_scale) / (2.0 * kPi), 1.0 / n_s2c);
            rebuild(s2c_, false, s2c_cst, [&](size_t i, size_t l2) { return ckks->q_last(l2) * (i == 0 ? final_scale / sv : 1.0); });
        }
        generated_ = true;
    }

private:
    // re-create the transforms with every diagonal multiplied by `cst`, then encode
    void rebuild(std::vector<DiagonalTransform> &ts, bool inverse, double cst, const std::function<double(size_t, size_t)> &scale_of) {
        const int S = static_cast<int>(slots());
        std::vector<DiagMap> maps = special_fft_groups(S, groups(!inverse), inverse);
        const size_t first = ts.front().level();
        ts.clear();
        for (size_t i = 0; i < maps.size(); i++) {
            for (auto &d : maps[i])
                for (auto &v : d.second) v *= cst;
            ts.emplace_back(ckks, std::move(maps[i]), S, first + i);
            ts.back().encode(scale_of(i, first + i));
        }
    }

public:
    void modraise_inplace(PhantomCiphertext &cipher) {
        PhantomCiphertext t = ckks->mod_raise(cipher);
        cipher = std::move(t);
    }
    // sparse (n < Nh: SubSum first) and full (n == Nh) packings take the same path
    void bootstrap_3(PhantomCiphertext &rtncipher, PhantomCiphertext &cipher) {
        if (!generated_) throw std::logic_error("generate_LT_coefficient_3() has not been called");
        const double in_scale = cipher.scale(), q0 = ckks->q(0);
        PhantomCiphertext tmp, pre;
        const PhantomCiphertext *src = &cipher;
        if (slim) {
            if (cipher.chain_index() > s2c_level_) throw std::invalid_argument("the slim order spends its SlotToCoeff levels before the modulus raise");
            pre = ckks->drop(cipher, s2c_level_);
            for (auto &lt : s2c_) pre = lt.apply(pre);
            src = &pre;
        } else if (std::abs(cipher.scale() / final_scale - 1.0) > 0.01)
            throw std::invalid_argument("the bootstrapper was set up for ciphertexts of scale final_scale");
        PhantomCiphertext t = ckks->mod_raise(ckks->at_level(*src, ckks->bottom(), tmp));
        t.set_scale(q0);  // src/boot/Bootstrapper.cu:2867-2868
        for (long k = n; k < Nh; k <<= 1) {  // SubSum onto the n-slot subring (:2870-2876)
            PhantomCiphertext r = ckks->rotate(t, static_cast<int>(k));
            ckks->add_inplace(t, r);
        }
        for (auto &lt : c2s_) t = lt.apply(t);  // (x_k + i x_{k+n}) / (2 Kb), bit-reversed slot order
        PhantomCiphertext cj = ckks->conj(t);
        PhantomCiphertext re = ckks->add(t, cj);
        PhantomCiphertext im = ckks->neg(ckks->mul_i(ckks->sub(t, cj)));
        // 2 pi m / q0 on both parts: two independent EvalMods, side by side on the sibling (context, stream) pair
        PhantomCiphertext v, vi;
        if (concurrent_evalmod) {
            cheb_sibling_ = ChebyshevEvaluator(&ckks->sibling(), cheb_coeffs_);
            auto pair = ckks->run_pair([&](CKKSEvaluator &ev) { return eval_mod(ev, cheb_, re); },
                                       [&](CKKSEvaluator &ev) { return eval_mod(ev, cheb_sibling_, im); });
            v = std::move(pair.first);
            vi = ckks->mul_i(pair.second);
        } else {
            v = eval_mod(*ckks, cheb_, re);
            vi = ckks->mul_i(eval_mod(*ckks, cheb_, im));
        }
        ckks->add_inplace(v, vi);
        if (slim) {
            v.set_scale(v.scale() * 2.0 * 3.14159265358979323846 * in_scale / q0);  // = in_scale by the choice of delta_out
            rtncipher = std::move(v);
            return;
        }
        for (auto &lt : s2c_) v = lt.apply(v);
        v.set_scale(v.scale() * in_scale / final_scale);  // src/boot/Bootstrapper.cu:2933 (a ratio of ~1)
        rtncipher = std::move(v);
    }
    void bootstrap_sparse_3(PhantomCiphertext &rtncipher, PhantomCiphertext &cipher) { bootstrap_3(rtncipher, cipher); }
    void bootstrap_full_3(PhantomCiphertext &rtncipher, PhantomCiphertext &cipher) { bootstrap_3(rtncipher, cipher); }
    void bootstrap_inplace_3(PhantomCiphertext &cipher) {
        PhantomCiphertext rtn;
        bootstrap_3(rtn, cipher);
        cipher = std::move(rtn);
    }
};

}  // namespace phantom
