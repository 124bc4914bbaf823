// ResNet inference on CKKS ciphertexts — the reference's CNN driver (include/cnn_phantom.h, src/cnn_phantom.cu,
// examples/cnn.cu: the multiplexed parallel convolutions of Lee et al., ICML 2022) on this repository's C++ host
// layer.  Header-only, on top of phantom_boot.hpp; every ciphertext operation is a CKKSEvaluator primitive, i.e. one
// or a few C-ABI calls.  The packing, the index formulas and the argument checks are the reference's (file:line cited
// per function); how the work is issued is this repository's (phantom-fhe_amd/cnn.py is the same circuit in Python):
//  * model constants are encoded once and kept on the device (PlainCache);
//  * a rotation by any step is one key switch; rotations of one ciphertext share their modup; the q partial sums of
//    a convolution rotate as one batch (pha_apply_galois_batched); the masked collecting rotations are one
//    double-hoisted sum (pha_linear_transform_double_hoisted); p output copies by log2 p doublings;
//  * batch norm adds a plaintext;
//  * the ReLU's sign polynomials are generated here (Remez exchange, double precision) and evaluated in the
//    Chebyshev basis at the optimal depth.
#pragma once

#include <algorithm>
#include <cmath>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "phantom_boot.hpp"

namespace phantom {

// ---- packing ----------------------------------------------------------------------------------------------------
// A tensor of c channels of h x w pixels with gap k: slot k^2hw*u + kw*v1 + v2 of a block holds channel
// k^2 u + k (v1 mod k) + (v2 mod k) at pixel (v1 div k, v2 div k); t = ceil(c / k^2) blocks; the n/p-slot vector is
// repeated p times (TensorCipher, src/cnn_phantom.cu:3-60).
class TensorCipher {
    int k_ = 0, h_ = 0, w_ = 0, c_ = 0, t_ = 0, p_ = 0, logn_ = 0;
    PhantomCiphertext cipher_;

public:
    TensorCipher() = default;
    TensorCipher(int logn, int k, int h, int w, int c, int t, int p, PhantomCiphertext cipher)
        : k_(k), h_(h), w_(w), c_(c), t_(t), p_(p), logn_(logn), cipher_(std::move(cipher)) {}
    TensorCipher(const TensorCipher &) = default;
    TensorCipher &operator=(const TensorCipher &) = default;
    TensorCipher(TensorCipher &&) = default;
    TensorCipher &operator=(TensorCipher &&) = default;
    int k() const { return k_; }
    int h() const { return h_; }
    int w() const { return w_; }
    int c() const { return c_; }
    int t() const { return t_; }
    int p() const { return p_; }
    int logn() const { return logn_; }
    const PhantomCiphertext &cipher() const { return cipher_; }
    void set_ciphertext(PhantomCiphertext cipher) { cipher_ = std::move(cipher); }
    // a copy whose buffer is filled in `stream`'s order (the copy constructor synchronises the device)
    TensorCipher clone(const hipStream_t &stream) const { return TensorCipher(logn_, k_, h_, w_, c_, t_, p_, cipher_.clone(stream)); }
    bool same_parms(const TensorCipher &o) const {
        return k_ == o.k_ && h_ == o.h_ && w_ == o.w_ && c_ == o.c_ && t_ == o.t_ && p_ == o.p_ && logn_ == o.logn_;
    }
};

namespace cnn_detail {
inline int log2_long(long v) {  // log2 of a power of two, -1 otherwise
    if (v <= 0 || (v & (v - 1))) return -1;
    int r = 0;
    while ((1L << r) < v) r++;
    return r;
}
inline long floor_pow2(double v) { return 1L << static_cast<int>(std::floor(std::log2(v))); }
inline int mod_steps(long step, long n) { return static_cast<int>(((step % n) + n) % n); }
inline cvec to_cvec(const std::vector<double> &v) {
    cvec out(v.size());
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    return out;
}
}  // namespace cnn_detail

inline std::vector<double> pack_tensor(const std::vector<double> &a, int c, int h, int w, int logn, int k, int p) {
    const long n = 1L << logn, t = (c + k * k - 1) / (k * k), block = static_cast<long>(k) * k * h * w;
    if (block * t * p > n || n % p) throw std::out_of_range("k^2 h w t p is larger than n");
    std::vector<double> v(n, 0.0);
    for (long j = 0; j < n; j++) {
        const long pos = j % (n / p);
        if (pos >= block * t) continue;
        const long blk = pos / block, v1 = (pos % block) / (k * w), v2 = pos % (k * w);
        const long ch = k * k * blk + k * (v1 % k) + v2 % k;
        if (ch < c) v[j] = a[(ch * h + v1 / k) * w + v2 / k];
    }
    return v;
}

// ---- encoded model constants, by name ------------------------------------------------------------------------------
class PlainCache {
    std::map<std::string, std::unique_ptr<PhantomPlaintext>> store_;

public:
    // nullptr: an all-zero constant.  `make` runs on the first request only.
    const PhantomPlaintext *get(CKKSEvaluator &ev, const std::string &key, const std::function<bool(std::vector<double> &)> &make,
                                double scale, size_t level, size_t slots, bool over_qp = false) {
        auto it = store_.find(key);
        if (it != store_.end()) return it->second.get();
        std::vector<double> v(slots, 0.0);
        std::unique_ptr<PhantomPlaintext> pt;
        if (make(v)) {
            pt = std::make_unique<PhantomPlaintext>(over_qp ? ev.encode_qp(cnn_detail::to_cvec(v), scale, level, slots)
                                                            : ev.encode(cnn_detail::to_cvec(v), scale, level, slots));
        }
        return store_.emplace(key, std::move(pt)).first->second.get();
    }
    size_t size() const { return store_.size(); }
    // a layer whose constants are all encoded may run its independent parts side by side
    std::set<std::string> warm;
};

namespace cnn_detail {
// fn over [first, last) split over up to 2^depth evaluators (nested run_pair)
template <class R, class Fn, class Join>
R fan_out(CKKSEvaluator &e, Fn &fn, Join &join, long first, long last, int depth) {
    if (depth == 0 || last - first < 2) return fn(e, first, last);
    const long mid = (first + last) / 2;
    auto pr = e.run_pair([&](CKKSEvaluator &e1) { return fan_out<R>(e1, fn, join, first, mid, depth - 1); },
                         [&](CKKSEvaluator &e2) { return fan_out<R>(e2, fn, join, mid, last, depth - 1); });
    return join(e, pr.first, pr.second);
}
}  // namespace cnn_detail

// ---- several ciphertexts of one (level, scale) in one buffer [batch][2][l][N] ----------------------------------------
struct CipherBatch {
    device_words data;
    size_t batch = 0, level = 0, limbs = 0, n = 0;
    double scale = 0;
    size_t ct_words() const { return 2 * limbs * n; }
};

class CnnOps {
public:
    static const hipStream_t &st(CKKSEvaluator &ev) { return ev.stream.get_stream(); }
    static CipherBatch gather(CKKSEvaluator &ev, const std::vector<PhantomCiphertext> &cts) {
        CipherBatch b;
        b.batch = cts.size(), b.level = cts[0].chain_index(), b.limbs = cts[0].coeff_modulus_size(), b.n = ev.n(), b.scale = cts[0].scale();
        b.data = device_words(b.batch * b.ct_words(), st(ev));
        for (size_t i = 0; i < cts.size(); i++)
            check_hip(hipMemcpyAsync(b.data.get() + i * b.ct_words(), cts[i].data(), b.ct_words() * sizeof(uint64_t), hipMemcpyDeviceToDevice, st(ev)),
                      "hipMemcpyAsync");
        return b;
    }
    static CipherBatch like(CKKSEvaluator &ev, const CipherBatch &a) {
        CipherBatch b;
        b.batch = a.batch, b.level = a.level, b.limbs = a.limbs, b.n = a.n, b.scale = a.scale;
        b.data = device_words(b.batch * b.ct_words(), st(ev));
        return b;
    }
    // one rotation of every ciphertext of the batch (pha_apply_galois_batched)
    static CipherBatch rotate(CKKSEvaluator &ev, const CipherBatch &a, int step) {
        CipherBatch b = like(ev, a);
        ev.ensure_keys({step});
        const uint32_t e = ev.elt(step);
        check(pha_apply_galois_batched(ev.context->handle(), a.level, a.data.get(), b.data.get(), a.batch, e,
                                       ev.galois_keys->get_relin_keys(e).public_keys_ptr(), st(ev)));
        return b;
    }
    static void add_inplace(CKKSEvaluator &ev, CipherBatch &a, const CipherBatch &b) {
        check(pha_add_rns_poly(ev.context->handle(), a.data.get(), a.data.get(), b.data.get(), a.limbs, 2 * a.batch, 2 * a.batch, st(ev)));
    }
    static PhantomCiphertext take(CKKSEvaluator &ev, const CipherBatch &a, size_t i) {
        PhantomCiphertext c;
        c.allocate(*ev.context, a.level, 2, st(ev));
        c.set_scale(a.scale);
        check_hip(hipMemcpyAsync(c.data(), a.data.get() + i * a.ct_words(), a.ct_words() * sizeof(uint64_t), hipMemcpyDeviceToDevice, st(ev)),
                  "hipMemcpyAsync");
        return c;
    }
    // po copies at distance n/po (the reference adds po rotations by -u n/po, src/cnn_phantom.cu:560-571)
    static PhantomCiphertext double_copies(CKKSEvaluator &ev, PhantomCiphertext ct, long n, long po) {
        for (long x = 1; x < po; x <<= 1) {
            PhantomCiphertext r = ev.rotate(ct, cnn_detail::mod_steps(-(n / po) * x, n));
            ev.add_inplace(ct, r);
        }
        return ct;
    }
};

// =====================================================================================================================
// layers
// =====================================================================================================================
// multiplexed_parallel_convolution_phantom, src/cnn_phantom.cu:306-574.  data: [co][ci][fh][fw] flat.  Two levels.
inline void multiplexed_parallel_convolution_phantom(CKKSEvaluator &ev, PlainCache &cache, const std::string &name, const TensorCipher &cnn_in,
                                                     TensorCipher &cnn_out, int co, int st, int fh, int fw, const std::vector<double> &data,
                                                     const std::vector<double> &running_var, const std::vector<double> &constant_weight,
                                                     double epsilon, bool end = false) {
    using namespace cnn_detail;
    const int ki = cnn_in.k(), hi = cnn_in.h(), wi = cnn_in.w(), ci = cnn_in.c(), ti = cnn_in.t(), pi = cnn_in.p(), logn = cnn_in.logn();
    int ko = 0, ho = 0, wo = 0;
    if (st != 1 && st != 2) throw std::invalid_argument("supported st is only 1 or 2");
    if (static_cast<int>(data.size()) != fh * fw * ci * co) throw std::invalid_argument("the size of data vector is not ker x ker x h x h");
    if (log2_long(ki) == -1) throw std::invalid_argument("ki is not power of two");
    if (static_cast<int>(running_var.size()) != co || static_cast<int>(constant_weight.size()) != co)
        throw std::invalid_argument("the size of running_var or weight is not correct");
    for (double num : running_var)
        if (std::abs(num) < 1e-16) throw std::invalid_argument("the size of running_var is too small. nearly zero.");
    if (fh % 2 == 0 || fw % 2 == 0) throw std::invalid_argument("fh and fw should be odd");
    if (st == 1)
        ho = hi, wo = wi, ko = ki;
    else {
        if (hi % 2 == 1 || wi % 2 == 1) throw std::invalid_argument("hi or wi is not even");
        ho = hi / 2, wo = wi / 2, ko = 2 * ki;
    }
    const long n = 1L << logn;
    const int to = (co + ko * ko - 1) / (ko * ko);
    const long po = floor_pow2(static_cast<double>(n) / static_cast<double>(ko * ko * ho * wo * to));
    const long q = (co + pi - 1) / pi;
    if (n % pi != 0) throw std::out_of_range("n is not divisible by pi");
    if (n % po != 0) throw std::out_of_range("n is not divisible by po");
    if (static_cast<long>(ki) * ki * hi * wi * ti * pi > n) throw std::out_of_range("ki^2 hi wi ti pi is larger than n");
    if (static_cast<long>(ko) * ko * ho * wo * to * po > n) throw std::out_of_range("ko^2 ho wo to po is larger than n");

    const PhantomCiphertext &ct = cnn_in.cipher();
    const size_t level = ct.chain_index();
    const long iblock = static_cast<long>(ki) * ki * hi * wi, oblock = static_cast<long>(ko) * ko * ho * wo;
    std::vector<int> tap_steps;
    for (int i1 = 0; i1 < fh; i1++)
        for (int i2 = 0; i2 < fw; i2++) tap_steps.push_back(mod_steps(static_cast<long>(ki) * ki * wi * (i1 - (fh - 1) / 2) + ki * (i2 - (fw - 1) / 2), n));
    std::vector<PhantomCiphertext> rots = ev.rotate_many(ct, tap_steps);
    auto weighted_sums = [&](CKKSEvaluator &e, long first, long last) {
        std::vector<PhantomCiphertext> out;
        for (long i9 = first; i9 < last; i9++) {
            PhantomCiphertext acc;
            for (int i1 = 0; i1 < fh; i1++)
                for (int i2 = 0; i2 < fw; i2++) {
                    // shifted weight vector (cnn_phantom.cu:382-401)
                    auto make = [&](std::vector<double> &v) {
                        bool any = false;
                        for (long j8 = 0; j8 < n; j8++) {
                            const long pos = j8 % (n / pi), i8 = j8 / (n / pi);
                            const long j5 = (pos % iblock) / (ki * wi), j6 = pos % (ki * wi), i7 = pos / iblock;
                            const long in_ch = ki * ki * i7 + ki * (j5 % ki) + j6 % ki, out_ch = i8 + pi * i9;
                            const long rr = j5 / ki - (fh - 1) / 2 + i1, cc = j6 / ki - (fw - 1) / 2 + i2;
                            if (pos >= iblock * ti || out_ch >= co || in_ch >= ci || cc < 0 || cc > wi - 1 || rr < 0 || rr > hi - 1) continue;
                            v[j8] = data[fh * fw * ci * out_ch + fh * fw * in_ch + fw * i1 + i2];
                            any = true;
                        }
                        return any;
                    };
                    const PhantomPlaintext *pt = cache.get(ev, name + "/w/" + std::to_string(i9) + "/" + std::to_string(i1) + "/" + std::to_string(i2),
                                                           make, ev.q_last(level), level, n);
                    if (pt) e.mul_plain_add(acc, rots[i1 * fw + i2], *pt);
                }
            out.push_back(e.rescale(acc));
        }
        return out;
    };
    // once the constants are encoded, the halves of the partial sums (and of the collection below) run side by side
    // on the evaluator's sibling (context, stream) pairs: kernels on 2-5 limbs leave most of the GPU idle
    const int depth = cache.warm.count(name) ? 2 : 0;
    auto concat = [](CKKSEvaluator &, std::vector<PhantomCiphertext> &a, std::vector<PhantomCiphertext> &b) {
        for (auto &c : b) a.push_back(std::move(c));
        return std::move(a);
    };
    std::vector<PhantomCiphertext> parts = fan_out<std::vector<PhantomCiphertext>>(ev, weighted_sums, concat, 0, q, depth);
    // summation over the input channels (cnn_phantom.cu:499-537): the q partial sums move as one batch
    CipherBatch var = CnnOps::gather(ev, parts);
    parts.clear();
    const int d = log2_long(ki), c = log2_long(ti);
    std::vector<long> sum_steps;
    for (int x = 0; x < d; x++) sum_steps.push_back(1L << x);
    for (int x = 0; x < d; x++) sum_steps.push_back((1L << x) * ki * wi);
    if (c != -1)
        for (int x = 0; x < c; x++) sum_steps.push_back((1L << x) * iblock);
    for (long step : sum_steps) {
        CipherBatch r = CnnOps::rotate(ev, var, mod_steps(step, n));
        CnnOps::add_inplace(ev, var, r);
    }
    if (c == -1) {
        CipherBatch acc = CnnOps::like(ev, var);
        check_hip(hipMemcpyAsync(acc.data.get(), var.data.get(), var.batch * var.ct_words() * sizeof(uint64_t), hipMemcpyDeviceToDevice, CnnOps::st(ev)),
                  "hipMemcpyAsync");
        for (int x = 1; x < ti; x++) {
            CipherBatch r = CnnOps::rotate(ev, var, mod_steps(iblock * x, n));
            CnnOps::add_inplace(ev, acc, r);
        }
        var = std::move(acc);
    }
    // collect the valid values into the output packing (cnn_phantom.cu:539-554); the masks carry the batch-norm factor
    auto collect = [&](CKKSEvaluator &e, long first, long last) {
        PhantomCiphertext total;
        for (long i9 = first; i9 < last; i9++) {
            std::vector<int> steps;
            std::vector<const PhantomPlaintext *> plains;
            for (long i8 = 0; i8 < pi && pi * i9 + i8 < co; i8++) {
                const long j4 = pi * i9 + i8;
                steps.push_back(mod_steps((n / pi) * (j4 % pi) - j4 % ko - (j4 / (ko * ko)) * oblock - ((j4 % (ko * ko)) / ko) * ko * wo, n));
                auto make = [&](std::vector<double> &v) {
                    const double f = constant_weight[j4] / std::sqrt(running_var[j4] + epsilon);
                    for (long u3 = 0; u3 < to; u3++)
                        for (long v1 = 0; v1 < static_cast<long>(ko) * ho; v1++)
                            for (long v2 = 0; v2 < static_cast<long>(ko) * wo; v2++)
                                if (ko * ko * u3 + ko * (v1 % ko) + v2 % ko == j4) v[oblock * u3 + ko * wo * v1 + v2] = f;
                    return true;
                };
                plains.push_back(cache.get(ev, name + "/sel/" + std::to_string(j4), make, ev.q_last(level + 1), level + 1, n, true));
            }
            PhantomCiphertext v = CnnOps::take(e, var, i9);
            PhantomCiphertext part = e.bsgs(v, steps, {0}, plains);
            if (total.data() == nullptr)
                total = std::move(part);
            else
                e.add_inplace(total, part);
        }
        return total;
    };
    auto sum = [](CKKSEvaluator &e, PhantomCiphertext &a, PhantomCiphertext &b) {
        e.add_inplace(a, b);
        return std::move(a);
    };
    PhantomCiphertext total = fan_out<PhantomCiphertext>(ev, collect, sum, 0, q, depth);
    cache.warm.insert(name);
    PhantomCiphertext out = ev.rescale(total);
    if (!end) out = CnnOps::double_copies(ev, std::move(out), n, po);
    cnn_out = TensorCipher(logn, ko, ho, wo, co, to, end ? 1 : static_cast<int>(po), std::move(out));
}

// multiplexed_parallel_batch_norm_phantom, src/cnn_phantom.cu:576-626
inline void multiplexed_parallel_batch_norm_phantom(CKKSEvaluator &ev, PlainCache &cache, const std::string &name, const TensorCipher &cnn_in,
                                                    TensorCipher &cnn_out, const std::vector<double> &bias, const std::vector<double> &running_mean,
                                                    const std::vector<double> &running_var, const std::vector<double> &weight, double epsilon,
                                                    double B) {
    const int ki = cnn_in.k(), hi = cnn_in.h(), wi = cnn_in.w(), ci = cnn_in.c(), ti = cnn_in.t(), pi = cnn_in.p(), logn = cnn_in.logn();
    if (static_cast<int>(bias.size()) != ci || static_cast<int>(running_mean.size()) != ci || static_cast<int>(running_var.size()) != ci ||
        static_cast<int>(weight.size()) != ci)
        throw std::invalid_argument("the size of bias, running_mean, running_var, or weight are not correct");
    for (double num : running_var)
        if (std::abs(num) < 1e-16) throw std::invalid_argument("the size of running_var is too small. nearly zero.");
    const long n = 1L << logn, block = static_cast<long>(ki) * ki * hi * wi;
    if (static_cast<long>(hi) * wi * ci > n) throw std::invalid_argument("hi*wi*ci should not be larger than n");
    if (n % pi != 0) throw std::out_of_range("n is not divisible by pi");
    const PhantomCiphertext &ct = cnn_in.cipher();
    auto make = [&](std::vector<double> &g) {
        for (long v4 = 0; v4 < n; v4++) {
            const long pos = v4 % (n / pi), v1 = (pos % block) / (ki * wi), v2 = pos % (ki * wi), u3 = pos / block;
            const long idx = ki * ki * u3 + ki * (v1 % ki) + v2 % ki;
            if (idx >= ci || pos >= block * ti) continue;
            g[v4] = -(running_mean[idx] * weight[idx] / std::sqrt(running_var[idx] + epsilon) - bias[idx]) / B;
        }
        return true;
    };
    PhantomCiphertext out = ev.add_plain(ct, *cache.get(ev, name + "/bn", make, ct.scale(), ct.chain_index(), n));
    cnn_out = TensorCipher(logn, ki, hi, wi, ci, ti, pi, std::move(out));
}

// cnn_add_phantom, src/cnn_phantom.cu:663-680
inline void cnn_add_phantom(CKKSEvaluator &ev, const TensorCipher &cnn1, const TensorCipher &cnn2, TensorCipher &destination) {
    if (!cnn1.same_parms(cnn2)) throw std::invalid_argument("the parameters of cnn1 and cnn2 are not the same");
    const size_t lvl = std::max(cnn1.cipher().chain_index(), cnn2.cipher().chain_index());
    PhantomCiphertext t1, t2;
    PhantomCiphertext sum = ev.add(ev.at_level(cnn1.cipher(), lvl, t1), ev.at_level(cnn2.cipher(), lvl, t2));
    destination = TensorCipher(cnn1.logn(), cnn1.k(), cnn1.h(), cnn1.w(), cnn1.c(), cnn1.t(), cnn1.p(), std::move(sum));
}

// multiplexed_parallel_downsampling_phantom, src/cnn_phantom.cu:681-756.  One level; the masks are applied after the
// (hoisted) rotations, rotated accordingly.
inline void multiplexed_parallel_downsampling_phantom(CKKSEvaluator &ev, PlainCache &cache, const std::string &name, const TensorCipher &cnn_in,
                                                      TensorCipher &cnn_out) {
    using namespace cnn_detail;
    const int ki = cnn_in.k(), hi = cnn_in.h(), wi = cnn_in.w(), ci = cnn_in.c(), ti = cnn_in.t(), logn = cnn_in.logn();
    const long n = 1L << logn;
    const int ko = 2 * ki, ho = hi / 2, wo = wi / 2, to = ti / 2, co = 2 * ci;
    if (ti % 8 != 0) throw std::invalid_argument("ti is not multiple of 8");
    if (hi % 2 != 0) throw std::invalid_argument("hi is not even");
    if (wi % 2 != 0) throw std::invalid_argument("wi is not even");
    const long po = floor_pow2(static_cast<double>(n) / static_cast<double>(ko * ko * ho * wo * to));
    if (n % po != 0) throw std::out_of_range("n is not divisible by po");
    const long iblock = static_cast<long>(ki) * ki * hi * wi, oblock = static_cast<long>(ko) * ko * ho * wo;
    const PhantomCiphertext &ct = cnn_in.cipher();
    const size_t level = ct.chain_index();
    std::vector<int> steps;
    for (int w1 = 0; w1 < ki; w1++)
        for (int w2 = 0; w2 < ti; w2++) {
            const long m = static_cast<long>(ki) * w2 + w1, w3 = (m % (2 * ko)) / 2, w4 = m % 2, w5 = m / (2 * ko);
            steps.push_back(mod_steps(iblock * w2 + ki * wi * w1 - oblock * w5 - ko * wo * w3 - ki * w4 - oblock * (ti / 8), n));
        }
    std::vector<PhantomCiphertext> rots = ev.rotate_many(ct, steps);
    PhantomCiphertext total;
    size_t idx = 0;
    for (int w1 = 0; w1 < ki; w1++)
        for (int w2 = 0; w2 < ti; w2++, idx++) {
            const long stp = steps[idx];
            auto make = [&](std::vector<double> &v) {
                for (long v4 = 0; v4 < iblock * ti && v4 < n; v4++) {
                    const long j5 = (v4 % iblock) / (ki * wi), j6 = v4 % (ki * wi), i7 = v4 / iblock;
                    if ((j5 / ki) % 2 == 0 && (j6 / ki) % 2 == 0 && j5 % ki == w1 && i7 == w2) v[((v4 - stp) % n + n) % n] = 1.0;
                }
                return true;
            };
            ev.mul_plain_add(total, rots[idx],
                             *cache.get(ev, name + "/down/" + std::to_string(w1) + "/" + std::to_string(w2), make, ev.q_last(level), level, n));
        }
    PhantomCiphertext out = CnnOps::double_copies(ev, ev.rescale(total), n, po);
    cnn_out = TensorCipher(logn, ko, ho, wo, co, to, static_cast<int>(po), std::move(out));
}

// averagepooling_phantom_scale, src/cnn_phantom.cu:758-828.  One level; channel j lands in slot j, times B.
inline void averagepooling_phantom_scale(CKKSEvaluator &ev, PlainCache &cache, const std::string &name, const TensorCipher &cnn_in,
                                         TensorCipher &cnn_out, double B) {
    using namespace cnn_detail;
    const int ki = cnn_in.k(), hi = cnn_in.h(), wi = cnn_in.w(), ci = cnn_in.c(), ti = cnn_in.t(), logn = cnn_in.logn();
    const long n = 1L << logn;
    if (log2_long(hi) == -1) throw std::invalid_argument("hi is not power of two");
    if (log2_long(wi) == -1) throw std::invalid_argument("wi is not power of two");
    PhantomCiphertext ct = cnn_in.cipher().clone(CnnOps::st(ev));
    for (int x = 0; x < log2_long(wi); x++) {
        PhantomCiphertext r = ev.rotate(ct, mod_steps((1L << x) * ki, n));
        ev.add_inplace(ct, r);
    }
    for (int x = 0; x < log2_long(hi); x++) {
        PhantomCiphertext r = ev.rotate(ct, mod_steps((1L << x) * ki * ki * wi, n));
        ev.add_inplace(ct, r);
    }
    const size_t level = ct.chain_index();
    std::vector<int> steps;
    for (int s = 0; s < ki; s++)
        for (int u = 0; u < ti; u++) steps.push_back(mod_steps(-static_cast<long>(ki * u + s) * ki + static_cast<long>(ki) * ki * hi * wi * u + ki * wi * s, n));
    std::vector<PhantomCiphertext> rots = ev.rotate_many(ct, steps);
    PhantomCiphertext total;
    size_t idx = 0;
    for (int s = 0; s < ki; s++)
        for (int u = 0; u < ti; u++, idx++) {
            auto make = [&](std::vector<double> &v) {
                for (int i = 0; i < ki; i++) v[(ki * u + s) * ki + i] = B / static_cast<double>(hi * wi);
                return true;
            };
            ev.mul_plain_add(total, rots[idx], *cache.get(ev, name + "/avg/" + std::to_string(s) + "/" + std::to_string(u), make, ev.q_last(level), level, n));
        }
    cnn_out = TensorCipher(logn, 1, 1, 1, ci, ti, 1, ev.rescale(total));
}

// matrix_multiplication_phantom, src/cnn_phantom.cu:829-875: q x r matrix times the first r slots, diagonal method.
// One level.  The bias is added (the reference builds the vector, :842-845, and never adds it).
inline void matrix_multiplication_phantom(CKKSEvaluator &ev, PlainCache &cache, const std::string &name, const TensorCipher &cnn_in,
                                          TensorCipher &cnn_out, const std::vector<double> &matrix, const std::vector<double> &bias, int q, int r) {
    using namespace cnn_detail;
    const int logn = cnn_in.logn();
    const long n = 1L << logn;
    if (static_cast<int>(matrix.size()) != q * r) throw std::invalid_argument("the size of matrix is not q*r");
    if (static_cast<int>(bias.size()) != q) throw std::invalid_argument("the size of bias is not q");
    const PhantomCiphertext &ct = cnn_in.cipher();
    const size_t level = ct.chain_index();
    std::vector<int> steps;
    for (int s = 0; s < q + r - 1; s++) steps.push_back(mod_steps(r - 1 - s, n));
    std::vector<PhantomCiphertext> rots = ev.rotate_many(ct, steps);
    PhantomCiphertext total;
    for (int s = 0; s < q + r - 1; s++) {
        auto make = [&](std::vector<double> &v) {
            for (int i = 0; i < q; i++) {
                const int j = i + r - 1 - s;
                if (j >= 0 && j < r) v[i] = matrix[i * r + j];
            }
            return true;
        };
        ev.mul_plain_add(total, rots[s], *cache.get(ev, name + "/fc/" + std::to_string(s), make, ev.q_last(level), level, n));
    }
    PhantomCiphertext out = ev.rescale(total);
    auto make_bias = [&](std::vector<double> &v) {
        for (int i = 0; i < q; i++) v[i] = bias[i];
        return true;
    };
    out = ev.add_plain(out, *cache.get(ev, name + "/fc_bias", make_bias, out.scale(), out.chain_index(), n));
    cnn_out = TensorCipher(logn, cnn_in.k(), cnn_in.h(), cnn_in.w(), cnn_in.c(), cnn_in.t(), cnn_in.p(), std::move(out));
}

// =====================================================================================================================
// approximate ReLU (minimax_ReLU_phantom, src/cnn_phantom.cu:876-940; degrees {15, 15, 27}, examples/cnn.cu:54-56)
// =====================================================================================================================
class ApproxReLU {
public:
    std::vector<int> degrees;
    std::vector<std::vector<double>> components, scaled;  // Chebyshev coefficients in x / bound; scaled: what the circuit evaluates
    std::vector<double> bounds, errors;
    double eps = 0, precision = 0;
    int depth = 0;
    bool parallel = true;  // the halves of every component polynomial on sibling streams (ChebyshevEvaluator::parallel)

private:
    static double cheb_eval(const std::vector<double> &c, double x) {  // Clenshaw
        double b1 = 0, b2 = 0;
        for (size_t j = c.size(); j-- > 1;) {
            const double t = 2 * x * b1 - b2 + c[j];
            b2 = b1, b1 = t;
        }
        return x * b1 - b2 + c[0];
    }
    static std::vector<double> solve(std::vector<std::vector<double>> A, std::vector<double> b) {  // Gaussian elimination, partial pivoting
        const size_t m = b.size();
        for (size_t col = 0; col < m; col++) {
            size_t piv = col;
            for (size_t r = col + 1; r < m; r++)
                if (std::abs(A[r][col]) > std::abs(A[piv][col])) piv = r;
            std::swap(A[col], A[piv]);
            std::swap(b[col], b[piv]);
            for (size_t r = col + 1; r < m; r++) {
                const double f = A[r][col] / A[col][col];
                for (size_t k = col; k < m; k++) A[r][k] -= f * A[col][k];
                b[r] -= f * b[col];
            }
        }
        std::vector<double> x(m);
        for (size_t i = m; i-- > 0;) {
            double acc = b[i];
            for (size_t k = i + 1; k < m; k++) acc -= A[i][k] * x[k];
            x[i] = acc / A[i][i];
        }
        return x;
    }
    // the odd polynomial of degree deg closest to 1 on [a, b] in the maximum norm, as Chebyshev coefficients in x / b
    static std::vector<double> remez_odd(int deg, double a, double b, double &err_out, int grid = 10000) {
        const int m = (deg + 1) / 2;
        std::vector<double> xs;
        for (int i = 0; i < grid; i++) {
            xs.push_back(a * std::pow(b / a, static_cast<double>(i) / (grid - 1)));
            xs.push_back(a + (b - a) * static_cast<double>(i) / (grid - 1));
        }
        std::sort(xs.begin(), xs.end());
        xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
        const size_t G = xs.size();
        auto basis = [&](double x, int k) { return std::cos((2 * k + 1) * std::acos(std::min(1.0, std::max(-1.0, x / b)))); };
        std::vector<size_t> ref;
        for (int i = 0; i <= m; i++) {
            const double x = b / a > 50 ? a * std::pow(b / a, static_cast<double>(i) / m) : a + (b - a) * (1 - std::cos(3.14159265358979323846 * i / m)) / 2;
            size_t idx = std::lower_bound(xs.begin(), xs.end(), x) - xs.begin();
            ref.push_back(std::min(idx, G - 1));
        }
        std::sort(ref.begin(), ref.end());
        ref.erase(std::unique(ref.begin(), ref.end()), ref.end());
        if (static_cast<int>(ref.size()) < m + 1) {
            ref.clear();
            for (int i = 0; i <= m; i++) ref.push_back((G - 1) * i / m);
        }
        std::vector<double> best;
        double best_err = 1e300;
        std::vector<double> err(G);
        for (int it = 0; it < 60; it++) {
            std::vector<std::vector<double>> A(m + 1, std::vector<double>(m + 1));
            for (int i = 0; i <= m; i++) {
                for (int k = 0; k < m; k++) A[i][k] = basis(xs[ref[i]], k);
                A[i][m] = i % 2 ? -1.0 : 1.0;
            }
            std::vector<double> sol = solve(A, std::vector<double>(m + 1, 1.0));
            const double E = sol[m];
            double worst = 0;
            for (size_t g = 0; g < G; g++) {
                const double th = std::acos(std::min(1.0, std::max(-1.0, xs[g] / b)));
                double v = 0;
                for (int k = 0; k < m; k++) v += sol[k] * std::cos((2 * k + 1) * th);
                err[g] = v - 1.0;
                worst = std::max(worst, std::abs(err[g]));
            }
            if (worst < best_err) {
                best_err = worst;
                best.assign(sol.begin(), sol.begin() + m);
            }
            if (worst - std::abs(E) <= 1e-9 * worst + 1e-15 || worst < 1e-11) break;
            // one extremum per run of constant sign, then the m + 1 consecutive ones with the largest errors
            std::vector<size_t> ext;
            size_t run_best = 0;
            for (size_t g = 0; g < G; g++) {
                const bool new_run = g == 0 || ((err[g] >= 0) != (err[g - 1] >= 0));
                if (new_run) {
                    if (g) ext.push_back(run_best);
                    run_best = g;
                } else if (std::abs(err[g]) > std::abs(err[run_best]))
                    run_best = g;
            }
            ext.push_back(run_best);
            if (ext.size() > static_cast<size_t>(64 * m) && worst < 1e-9) break;
            size_t lo = 0, hi = ext.size();
            while (hi - lo > static_cast<size_t>(m + 1)) {
                if (std::abs(err[ext[lo]]) < std::abs(err[ext[hi - 1]]))
                    lo++;
                else
                    hi--;
            }
            if (hi - lo < static_cast<size_t>(m + 1)) break;
            ref.assign(ext.begin() + lo, ext.begin() + hi);
        }
        std::vector<double> full(deg + 1, 0.0);
        for (int k = 0; k < m; k++) full[2 * k + 1] = best[k];
        err_out = best_err;
        return full;
    }
    bool build(double e0) {
        double a = e0, b = 1.0;
        components.clear(), bounds.clear(), errors.clear();
        for (int d : degrees) {
            double e;
            components.push_back(remez_odd(d, a, b, e));
            bounds.push_back(b);
            errors.push_back(e);
            if (!(e < 1.0)) return false;
            a = 1.0 - e, b = 1.0 + e;
        }
        return true;
    }
    double relu_error() const {
        double worst = 0;
        auto probe = [&](double x) { worst = std::max(worst, std::abs((*this)(x) - std::max(x, 0.0))); };
        for (int i = 0; i <= 100000; i++) probe(-1.0 + 2.0 * i / 100000);
        for (int i = 0; i < 50000; i++) {
            const double x = std::pow(10.0, -6.0 + 6.0 * i / 49999);
            probe(x), probe(-x);
        }
        return worst;
    }

public:
    explicit ApproxReLU(std::vector<int> degs = {15, 15, 27}, double eps_ = 0) : degrees(std::move(degs)) {
        if (eps_ <= 0) {  // eps minimising the maximum error of the ReLU over [-1, 1]
            double top = 0;
            for (int d : degrees) top += std::log2(static_cast<double>(d));
            double best = 1e300;
            for (double le = std::max(1.0, top - 6.0); le < top; le += 0.05) {
                if (!build(std::exp2(-le))) break;
                const double e = relu_error();
                if (e < best)
                    best = e, eps_ = std::exp2(-le);
                else if (e > 4 * best)
                    break;
            }
        }
        if (!build(eps_)) throw std::invalid_argument("the degrees cannot separate [eps, 1] from its mirror image");
        eps = eps_;
        precision = relu_error();
        for (size_t i = 0; i < components.size(); i++) {
            const double nxt = i + 1 < components.size() ? bounds[i + 1] : 2.0;
            std::vector<double> c = components[i];
            for (double &v : c) v /= nxt;
            scaled.push_back(std::move(c));
        }
        depth = 1;
        for (int d : degrees) depth += std::max(2, static_cast<int>(std::ceil(std::log2(d + 1.0))));
    }
    double sign(double x) const {
        double y = x;
        for (size_t i = 0; i < components.size(); i++) y = cheb_eval(components[i], std::min(1.0, std::max(-1.0, y / bounds[i])));
        return y;
    }
    double operator()(double x) const { return x * (1.0 + sign(x)) / 2.0; }

    // levels: depth; the result has the scale of the input
    PhantomCiphertext evaluate(CKKSEvaluator &ev, const PhantomCiphertext &ct) const {
        PhantomCiphertext y = ct.clone(ev.stream.get_stream());
        for (size_t i = 0; i < scaled.size(); i++) {
            ChebyshevEvaluator cheb(&ev, scaled[i], 2, true);
            cheb.parallel = parallel;
            const bool last = i + 1 == scaled.size();
            // the last component lands on the scale q_last: (sign/2 + 1/2) * x then rescales back onto ct.scale
            y = cheb(y, last ? ev.q_last(y.chain_index() + cheb.depth()) : y.scale());
        }
        y = ev.add_const(y, 0.5);
        PhantomCiphertext tmp;
        return ev.mul_rescale(y, ev.at_level(ct, y.chain_index(), tmp));
    }
};

// ReLU_phantom, src/cnn_phantom.cu:628-644
inline void ReLU_phantom(CKKSEvaluator &ev, const TensorCipher &cnn_in, TensorCipher &cnn_out, const ApproxReLU &relu) {
    if (static_cast<long>(cnn_in.h()) * cnn_in.w() * cnn_in.c() > (1L << cnn_in.logn())) throw std::invalid_argument("hi*wi*ci should not be larger than n");
    cnn_out = TensorCipher(cnn_in.logn(), cnn_in.k(), cnn_in.h(), cnn_in.w(), cnn_in.c(), cnn_in.t(), cnn_in.p(), relu.evaluate(ev, cnn_in.cipher()));
}

// =====================================================================================================================
// parameters and the network (import_parameters_cifar10, src/cnn_phantom.cu:1474-1731; ResNet_cifar10_seal_sparse,
// examples/cnn.cu:48-426)
// =====================================================================================================================
using CnnParameters = std::map<std::string, std::vector<double>>;

inline std::vector<std::string> parameter_names(size_t end_num) {
    std::vector<std::string> names = {"conv1_weight", "bn1_bias", "bn1_running_mean", "bn1_running_var", "bn1_weight"};
    for (int j = 1; j <= 3; j++)
        for (size_t k = 0; k <= end_num; k++)
            for (int i = 1; i <= 2; i++) {
                const std::string stem = "layer" + std::to_string(j) + "_" + std::to_string(k);
                names.push_back(stem + "_conv" + std::to_string(i) + "_weight");
                for (const char *s : {"_bias", "_running_mean", "_running_var", "_weight"}) names.push_back(stem + "_bn" + std::to_string(i) + s);
            }
    names.push_back("linear_weight");
    names.push_back("linear_bias");
    return names;
}

inline CnnParameters import_parameters_cifar10(const std::string &directory, size_t end_num) {
    CnnParameters p;
    for (const std::string &name : parameter_names(end_num)) {
        std::ifstream in(directory + "/" + name + ".txt");
        if (!in) throw std::invalid_argument("file is not open: " + directory + "/" + name + ".txt");
        double v;
        std::vector<double> &dst = p[name];
        while (in >> v) dst.push_back(v);
    }
    return p;
}

class ResNetCifar10 {
public:
    CKKSEvaluator *ev;
    CnnParameters params;
    size_t end_num;
    double B, epsilon = 1e-5;
    ApproxReLU relu;
    std::map<long, Bootstrapper *> bootstrappers;  // slots -> bootstrapper; empty: no bootstrapping
    int logn, image_c, image_h, image_w, init_p;
    std::vector<int> widths{16, 32, 64};
    PlainCache cache;

    ResNetCifar10(CKKSEvaluator *ev_, CnnParameters params_, size_t end_num_, double B_, ApproxReLU relu_, int c = 3, int h = 32, int w = 32)
        : ev(ev_), params(std::move(params_)), end_num(end_num_), B(B_), relu(std::move(relu_)), image_c(c), image_h(h), image_w(w) {
        logn = static_cast<int>(ev->context->log_n()) - 1;
        init_p = static_cast<int>(cnn_detail::floor_pow2(static_cast<double>(1L << logn) / (h * w * c)));
    }
    // levels one block needs between two bootstrappings: ReLU, a convolution (2), and what the bootstrapper spends before
    // its modulus raise (the slim order's SlotToCoeff)
    int levels_after_bootstrap() const {
        size_t pre = 0;
        for (const auto &b : bootstrappers) pre = std::max(pre, b.second->pre_levels());
        return relu.depth + 2 + static_cast<int>(pre);
    }
    size_t input_level() const { return ev->bottom() - levels_after_bootstrap() - 2; }  // the first convolution comes before the first ReLU

    // image [c][h][w] in the network's units; the ciphertext holds image / B (examples/cnn.cu:263-278)
    TensorCipher encrypt_image(const std::vector<double> &image, double scale, size_t level) {
        std::vector<double> scaled(image);
        for (double &v : scaled) v /= B;
        std::vector<double> vec = pack_tensor(scaled, image_c, image_h, image_w, logn, 1, init_p);
        PhantomCiphertext ct = ev->encrypt(ev->encode(cnn_detail::to_cvec(vec), scale, level, size_t(1) << logn));
        return TensorCipher(logn, 1, image_h, image_w, image_c, image_c, init_p, std::move(ct));
    }
    std::vector<double> decrypt_logits(const TensorCipher &x) {
        cvec v = ev->decrypt_decode(x.cipher(), size_t(1) << x.logn());
        std::vector<double> out(params.at("linear_bias").size());
        for (size_t i = 0; i < out.size(); i++) out[i] = v[i].real();
        return out;
    }

private:
    void conv_bn(TensorCipher &x, const std::string &conv, const std::string &bn, int co, int st) {
        TensorCipher y;
        multiplexed_parallel_convolution_phantom(*ev, cache, conv, x, y, co, st, 3, 3, params.at(conv + "_weight"), params.at(bn + "_running_var"),
                                                 params.at(bn + "_weight"), epsilon);
        multiplexed_parallel_batch_norm_phantom(*ev, cache, bn, y, x, params.at(bn + "_bias"), params.at(bn + "_running_mean"),
                                                params.at(bn + "_running_var"), params.at(bn + "_weight"), epsilon, B);
    }
    void bootstrap(TensorCipher &x) {
        if (bootstrappers.empty()) return;
        Bootstrapper *bs = bootstrappers.at((1L << x.logn()) / x.p());
        PhantomCiphertext in = x.cipher().clone(ev->stream.get_stream()), out;
        bs->bootstrap_3(out, in);
        x.set_ciphertext(std::move(out));
    }
    void relu_inplace(TensorCipher &x) {
        TensorCipher y;
        ReLU_phantom(*ev, x, y, relu);
        x = std::move(y);
    }

public:
    TensorCipher operator()(const TensorCipher &image) {
        TensorCipher x = image.clone(ev->stream.get_stream());
        conv_bn(x, "conv1", "bn1", widths[0], 1);
        relu_inplace(x);
        for (int j = 0; j < 3; j++)
            for (size_t k = 0; k <= end_num; k++) {
                const std::string stem = "layer" + std::to_string(j + 1) + "_" + std::to_string(k);
                const int st = (j >= 1 && k == 0) ? 2 : 1;
                TensorCipher shortcut = x.clone(ev->stream.get_stream());
                conv_bn(x, stem + "_conv1", stem + "_bn1", widths[j], st);
                bootstrap(x);
                relu_inplace(x);
                conv_bn(x, stem + "_conv2", stem + "_bn2", widths[j], 1);
                if (st == 2) {
                    TensorCipher d;
                    multiplexed_parallel_downsampling_phantom(*ev, cache, stem + "_down", shortcut, d);
                    shortcut = std::move(d);
                }
                TensorCipher sum;
                cnn_add_phantom(*ev, shortcut, x, sum);
                x = std::move(sum);
                bootstrap(x);
                relu_inplace(x);
            }
        TensorCipher pooled, out;
        averagepooling_phantom_scale(*ev, cache, "avgpool", x, pooled, B);
        matrix_multiplication_phantom(*ev, cache, "linear", pooled, out, params.at("linear_weight"), params.at("linear_bias"),
                                      static_cast<int>(params.at("linear_bias").size()), widths[2]);
        return out;
    }
};

}  // namespace phantom
