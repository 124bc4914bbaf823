// C++ host layer over the C ABI (include/phantom_amd.h): the evaluator-facing objects and entry
// points of the reference, re-created natively for HIP.  Names, argument meaning and error
// behaviour follow include/evaluate.cuh, include/context.cuh, include/ciphertext.h and
// include/secretkey.h of the reference (cited per item); nothing here computes on the CPU — every
// operation is a call into libphantom_amd.so on the given HIP stream.
//
// Header-only; compile with hipcc (or any C++17 compiler plus the HIP runtime headers) and link
// -lphantom_amd -lamdhip64.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/phantom_amd.h"

namespace phantom {

// ---- error mapping: C status codes back to the exceptions the reference throws -----------------
inline void check(int rc) {
    if (rc == PHA_OK) return;
    const std::string msg = pha_last_error();
    if (rc == PHA_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (rc == PHA_ERR_LOGIC) throw std::logic_error(msg);
    throw std::runtime_error(msg);  // HIP failure (the reference prints and exits, include/cuda_wrapper.cuh:19-43)
}
inline void check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// ---- stream + device memory (include/cuda_wrapper.cuh:47-186) ----------------------------------
class hip_stream_wrapper {
    hipStream_t stream_ = nullptr;

public:
    hip_stream_wrapper() { check_hip(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate"); }
    ~hip_stream_wrapper() {
        if (stream_) (void)hipStreamDestroy(stream_);
    }
    hip_stream_wrapper(const hip_stream_wrapper &) = delete;
    hip_stream_wrapper &operator=(const hip_stream_wrapper &) = delete;
    const hipStream_t &get_stream() const { return stream_; }
};

// The reference's evaluator functions default their stream argument to one global stream created with the context
// (phantom::util::global_variables::default_stream, src/context.cu:16-18,137); same here, created on first use.
inline const hip_stream_wrapper &default_stream() {
    static hip_stream_wrapper *s = new hip_stream_wrapper;  // never destroyed: the HIP runtime may be gone at exit
    return *s;
}

// Stream-ordered recycling of temporaries (the role of the reference's cudaMallocAsync pool with an unlimited
// release threshold, include/cuda_wrapper.cuh:141-186, src/context.cu:130-135): a buffer released on stream s is
// handed to the next request of the same size on the SAME stream, whose work is enqueued after the previous
// owner's — no driver call, no synchronisation.  (hipMallocAsync / hipFreeAsync were measured first: a
// full-slot bootstrap at N = 2^16 left the GPU idle for a quarter of its 53 ms.)  Buffers are kept until exit.
class device_pool {
    std::map<std::pair<hipStream_t, size_t>, std::vector<uint64_t *>> free_;
    std::mutex m_;

public:
    static device_pool &instance() {
        static device_pool *p = new device_pool;  // never destroyed: the HIP runtime may be gone at exit
        return *p;
    }
    uint64_t *take(size_t n, hipStream_t s) {
        {
            std::lock_guard<std::mutex> g(m_);
            auto it = free_.find({s, n});
            if (it != free_.end() && !it->second.empty()) {
                uint64_t *p = it->second.back();
                it->second.pop_back();
                return p;
            }
        }
        uint64_t *p = nullptr;
        check_hip(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(uint64_t)), "hipMalloc");
        return p;
    }
    void give(uint64_t *p, size_t n, hipStream_t s) {
        std::lock_guard<std::mutex> g(m_);
        free_[{s, n}].push_back(p);
    }
};

// RAII device buffer of uint64_t words.  With a stream it comes from / returns to the stream-ordered pool;
// the release is ordered on the stream of the last set_stream().
class device_words {
    uint64_t *p_ = nullptr;
    size_t n_ = 0;
    hipStream_t stream_ = nullptr;
    bool pooled_ = false;

public:
    device_words() = default;
    explicit device_words(size_t n) { reset(n); }
    device_words(size_t n, hipStream_t stream) : n_(n), stream_(stream), pooled_(true) {
        if (n) p_ = device_pool::instance().take(n, stream);
    }
    ~device_words() { release(); }
    device_words(const device_words &) = delete;
    device_words &operator=(const device_words &) = delete;
    device_words(device_words &&o) noexcept : p_(o.p_), n_(o.n_), stream_(o.stream_), pooled_(o.pooled_) { o.p_ = nullptr, o.n_ = 0; }
    device_words &operator=(device_words &&o) noexcept {
        if (this != &o) {
            release();
            p_ = o.p_, n_ = o.n_, stream_ = o.stream_, pooled_ = o.pooled_;
            o.p_ = nullptr, o.n_ = 0;
        }
        return *this;
    }
    void release() {
        if (p_) {
            if (pooled_)
                device_pool::instance().give(p_, n_, stream_);
            else
                (void)hipFree(p_);
        }
        p_ = nullptr, n_ = 0;
    }
    void reset(size_t n) {
        release();
        pooled_ = false;
        if (n) check_hip(hipMalloc(reinterpret_cast<void **>(&p_), n * sizeof(uint64_t)), "hipMalloc");
        n_ = n;
    }
    // later work on `stream` has used the buffer: release it in that stream's order
    void set_stream(hipStream_t stream) {
        if (pooled_) stream_ = stream;
    }
    uint64_t *get() const { return p_; }
    size_t size() const { return n_; }
};

// ---- parameters (include/util/encryptionparams.h:68-207, CKKS subset) ---------------------------
namespace arith {
struct CoeffModulus {
    // src/modulus.cu:79-110
    static std::vector<uint64_t> Create(size_t poly_modulus_degree, const std::vector<int> &bit_sizes) {
        std::vector<uint64_t> out(bit_sizes.size());
        check(pha_coeff_modulus_create(poly_modulus_degree, bit_sizes.data(), bit_sizes.size(), out.data()));
        return out;
    }
};
}  // namespace arith

class EncryptionParameters {
    size_t poly_modulus_degree_ = 0;
    std::vector<uint64_t> coeff_modulus_;
    size_t special_modulus_size_ = 1;  // include/util/encryptionparams.h:375
    size_t secret_key_hamming_weight_ = 0, sparse_slots_ = 0;

public:
    EncryptionParameters() = default;
    explicit EncryptionParameters(int /*scheme_type::ckks*/) {}
    // 0 = dense ternary key; h > 0: exactly h non-zero coefficients (src/secretkey.cu:401-433)
    void set_secret_key_hamming_weight(size_t h) { secret_key_hamming_weight_ = h; }
    void set_sparse_slots(size_t s) { sparse_slots_ = s; }
    size_t secret_key_hamming_weight() const { return secret_key_hamming_weight_; }
    size_t sparse_slots() const { return sparse_slots_; }
    void set_poly_modulus_degree(size_t n) { poly_modulus_degree_ = n; }
    void set_coeff_modulus(const std::vector<uint64_t> &m) { coeff_modulus_ = m; }
    void set_special_modulus_size(size_t s) { special_modulus_size_ = s; }
    size_t poly_modulus_degree() const { return poly_modulus_degree_; }
    const std::vector<uint64_t> &coeff_modulus() const { return coeff_modulus_; }
    size_t special_modulus_size() const { return special_modulus_size_; }
};

// ---- context (include/context.cuh, src/context.cu:120-290) --------------------------------------
// chain index 0 = key level (Q ∪ P), 1 = first data level (Q), k = Q minus its last k-1 primes
class PhantomContext {
    pha_context *h_ = nullptr;
    EncryptionParameters parms_;
    size_t size_Q_ = 0;
    int log_n_ = 0;

public:
    explicit PhantomContext(const EncryptionParameters &parms) : parms_(parms) {
        const size_t n = parms.poly_modulus_degree();
        if (n == 0 || (n & (n - 1))) throw std::invalid_argument("poly_modulus_degree is invalid");
        while ((size_t(1) << log_n_) < n) ++log_n_;
        const auto &m = parms.coeff_modulus();
        if (parms.special_modulus_size() >= m.size()) throw std::invalid_argument("coeff_modulus is invalid");
        size_Q_ = m.size() - parms.special_modulus_size();
        check(pha_context_create(log_n_, m.data(), m.size(), parms.special_modulus_size(), &h_));
    }
    ~PhantomContext() { pha_context_destroy(h_); }
    PhantomContext(const PhantomContext &) = delete;
    PhantomContext &operator=(const PhantomContext &) = delete;

    pha_context *handle() const { return h_; }
    const EncryptionParameters &key_parms() const { return parms_; }
    size_t poly_degree() const { return parms_.poly_modulus_degree(); }
    int log_n() const { return log_n_; }
    size_t size_Q() const { return size_Q_; }
    size_t size_P() const { return parms_.special_modulus_size(); }
    size_t get_first_index() const { return 1; }
    size_t get_next_index(size_t chain_index) const { return chain_index + 1; }
    // number of limbs of a ciphertext at chain_index (context.cu:148-174)
    size_t coeff_mod_size(size_t chain_index) const {
        if (chain_index < 1 || chain_index > size_Q_) throw std::invalid_argument("chain_index out of range");
        return size_Q_ - (chain_index - 1);
    }
    uint64_t modulus(size_t i) const { return parms_.coeff_modulus().at(i); }
};

// ---- ciphertext container (include/ciphertext.h:6-262) ------------------------------------------
class PhantomCiphertext {
    size_t size_ = 0, coeff_modulus_size_ = 0, poly_modulus_degree_ = 0, chain_index_ = 0;
    double scale_ = 1.0;
    bool is_ntt_form_ = true;
    device_words data_;

public:
    PhantomCiphertext() = default;
    PhantomCiphertext(PhantomCiphertext &&) = default;
    PhantomCiphertext &operator=(PhantomCiphertext &&) = default;
    // Deep copies (the reference's `PhantomCiphertext destination = encrypted;`): correct from any stream — the
    // device is synchronised first, the copy runs on the default stream and is complete on return.  Hot paths use
    // clone(stream), which is stream-ordered and does not synchronise.
    PhantomCiphertext(const PhantomCiphertext &o) { *this = o; }
    PhantomCiphertext &operator=(const PhantomCiphertext &o) {
        if (this != &o) {
            check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
            const hipStream_t &s = default_stream().get_stream();
            *this = o.clone(s);
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
        return *this;
    }

    // reallocates and keeps the overlapping prefix, like ciphertext.h:56-86
    void resize(const PhantomContext &context, size_t chain_index, size_t size, const hipStream_t &stream) {
        const size_t limbs = context.coeff_mod_size(chain_index), n = context.poly_degree();
        resize(size, limbs, n, stream);
        chain_index_ = chain_index;
    }
    void resize(size_t size, size_t coeff_modulus_size, size_t poly_modulus_degree, const hipStream_t &stream) {
        const size_t words = size * coeff_modulus_size * poly_modulus_degree;
        if (words != data_.size()) {
            device_words fresh(words, stream);
            const size_t keep = std::min(words, data_.size());
            if (keep)
                check_hip(hipMemcpyAsync(fresh.get(), data_.get(), keep * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream),
                          "hipMemcpyAsync");
            if (words > keep)
                check_hip(hipMemsetAsync(fresh.get() + keep, 0, (words - keep) * sizeof(uint64_t), stream), "hipMemsetAsync");
            data_.set_stream(stream);  // the old buffer is freed after the copy, in stream order
            data_ = std::move(fresh);
        }
        size_ = size, coeff_modulus_size_ = coeff_modulus_size, poly_modulus_degree_ = poly_modulus_degree;
    }
    // uninitialised storage for an operation's destination (no copy, no memset)
    void allocate(const PhantomContext &context, size_t chain_index, size_t size, const hipStream_t &stream) {
        const size_t limbs = context.coeff_mod_size(chain_index), n = context.poly_degree();
        data_.set_stream(stream);
        data_ = device_words(size * limbs * n, stream);
        size_ = size, coeff_modulus_size_ = limbs, poly_modulus_degree_ = n, chain_index_ = chain_index;
    }
    // the first `size` polys of a larger buffer (after relinearisation), without reallocating
    void truncate(size_t size) {
        if (size > size_) throw std::invalid_argument("truncate cannot grow a ciphertext");
        size_ = size;
    }
    // every later use of this ciphertext on `stream` orders the eventual free after it
    void touch(const hipStream_t &stream) { data_.set_stream(stream); }
    PhantomCiphertext clone(const hipStream_t &stream) const {
        PhantomCiphertext c;
        c.data_ = device_words(size_ * coeff_modulus_size_ * poly_modulus_degree_, stream);
        check_hip(hipMemcpyAsync(c.data_.get(), data_.get(), size_ * coeff_modulus_size_ * poly_modulus_degree_ * sizeof(uint64_t),
                                 hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
        c.size_ = size_, c.coeff_modulus_size_ = coeff_modulus_size_, c.poly_modulus_degree_ = poly_modulus_degree_;
        c.chain_index_ = chain_index_, c.scale_ = scale_, c.is_ntt_form_ = is_ntt_form_;
        return c;
    }
    uint64_t *data() const { return data_.get(); }
    size_t size() const { return size_; }
    size_t coeff_modulus_size() const { return coeff_modulus_size_; }
    size_t poly_modulus_degree() const { return poly_modulus_degree_; }
    size_t chain_index() const { return chain_index_; }
    double scale() const { return scale_; }
    bool is_ntt_form() const { return is_ntt_form_; }
    void set_scale(double s) { scale_ = s; }
    void set_ntt_form(bool b) { is_ntt_form_ = b; }
    void set_chain_index(size_t c) { chain_index_ = c; }
};

// ---- switching keys (include/secretkey.h:85-113): dnum buffers [2][size_QP][N] + pointer array ----
class PhantomRelinKey {
    std::vector<device_words> digits_;
    device_words ptrs_;

public:
    PhantomRelinKey() = default;
    // takes ownership of already-generated key material (key generation is outside this layer)
    explicit PhantomRelinKey(std::vector<device_words> &&digits) : digits_(std::move(digits)), ptrs_(digits_.size()) {
        std::vector<uint64_t> h(digits_.size());
        for (size_t i = 0; i < h.size(); i++) h[i] = reinterpret_cast<uint64_t>(digits_[i].get());
        check_hip(hipMemcpy(ptrs_.get(), h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice), "hipMemcpy");
    }
    const uint64_t *const *public_keys_ptr() const { return reinterpret_cast<const uint64_t *const *>(ptrs_.get()); }
    size_t dnum() const { return digits_.size(); }
};

class PhantomGaloisKey {
    std::map<uint32_t, PhantomRelinKey> keys_;

public:
    void insert(uint32_t galois_elt, PhantomRelinKey &&key) { keys_[galois_elt] = std::move(key); }
    bool has(uint32_t galois_elt) const { return keys_.count(galois_elt) != 0; }
    const PhantomRelinKey &get_relin_keys(uint32_t galois_elt) const {
        auto it = keys_.find(galois_elt);
        if (it == keys_.end()) throw std::invalid_argument("Galois elt not present");
        return it->second;
    }
};

// ---- plaintext container (include/plaintext.h): RNS polynomial [l][N] in NTT form at a level ----
class PhantomPlaintext {
    size_t chain_index_ = 1, coeff_modulus_size_ = 0, poly_modulus_degree_ = 0;
    double scale_ = 1.0;
    device_words data_;

public:
    PhantomPlaintext() = default;
    PhantomPlaintext(PhantomPlaintext &&) = default;
    PhantomPlaintext &operator=(PhantomPlaintext &&) = default;
    PhantomPlaintext(const PhantomPlaintext &o) { *this = o; }  // deep copy, synchronous (see PhantomCiphertext)
    PhantomPlaintext &operator=(const PhantomPlaintext &o) {
        if (this != &o) {
            check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
            data_ = device_words(o.data_.size());
            if (o.data_.size())
                check_hip(hipMemcpy(data_.get(), o.data_.get(), o.data_.size() * sizeof(uint64_t), hipMemcpyDeviceToDevice), "hipMemcpy");
            chain_index_ = o.chain_index_, coeff_modulus_size_ = o.coeff_modulus_size_, poly_modulus_degree_ = o.poly_modulus_degree_;
            scale_ = o.scale_;
        }
        return *this;
    }
    // CKKS mod_switch_to_next on a plaintext: drop the last limb (src/evaluate.cu:702-723)
    void drop_last_limb() {
        if (coeff_modulus_size_ < 2) throw std::invalid_argument("end of modulus switching chain reached");
        coeff_modulus_size_ -= 1;  // limb-major layout: the remaining limbs are a prefix of the buffer
        chain_index_ += 1;
    }
    void resize(size_t coeff_modulus_size, size_t poly_modulus_degree, const hipStream_t &) {
        if (coeff_modulus_size * poly_modulus_degree != data_.size()) data_ = device_words(coeff_modulus_size * poly_modulus_degree);
        coeff_modulus_size_ = coeff_modulus_size, poly_modulus_degree_ = poly_modulus_degree;
    }
    uint64_t *data() const { return data_.get(); }
    size_t chain_index() const { return chain_index_; }
    size_t coeff_modulus_size() const { return coeff_modulus_size_; }
    size_t poly_modulus_degree() const { return poly_modulus_degree_; }
    double scale() const { return scale_; }
    void set_chain_index(size_t c) { chain_index_ = c; }
    void set_scale(double s) { scale_ = s; }
};

// ---- CKKS encoder (include/ckks.h, src/ckks.cu:130-283) ------------------------------------------
class PhantomCKKSEncoder {
    size_t slots_ = 0, sparse_slots_ = 0;

public:
    explicit PhantomCKKSEncoder(const PhantomContext &context)
        : slots_(context.poly_degree() >> 1), sparse_slots_(context.key_parms().sparse_slots()) {}
    // encode on an explicit slot count (a power of two <= N/2) without touching the encoder's own sparse_slots:
    // what the bootstrapper's diagonals and constants need
    void encode_slots(const PhantomContext &context, const std::vector<std::complex<double>> &values, size_t sparse_slots,
                      double scale, PhantomPlaintext &destination, size_t chain_index, const hip_stream_wrapper &stream_wrapper = default_stream()) {
        const hipStream_t &s = stream_wrapper.get_stream();
        destination.resize(context.coeff_mod_size(chain_index), context.poly_degree(), s);
        check(pha_ckks_encode(context.handle(), reinterpret_cast<const double *>(values.data()), values.size(), sparse_slots,
                              chain_index, scale, destination.data(), s));
        destination.set_chain_index(chain_index);
        destination.set_scale(scale);
    }
    size_t slot_count() const { return slots_; }
    size_t sparse_slots() const { return sparse_slots_; }
    void reset_sparse_slots() { sparse_slots_ = 0; }

    // the first encode fixes sparse_slots (next power of two >= values.size()), like src/ckks.cu:150-158
    void encode(const PhantomContext &context, const std::vector<std::complex<double>> &values, double scale,
                PhantomPlaintext &destination, size_t chain_index, const hip_stream_wrapper &stream_wrapper = default_stream()) {
        if (values.size() > slots_) throw std::invalid_argument("values_size is too large");
        if (sparse_slots_ == 0)
            for (sparse_slots_ = 1; sparse_slots_ < values.size(); sparse_slots_ <<= 1) {}
        const hipStream_t &s = stream_wrapper.get_stream();
        destination.resize(context.coeff_mod_size(chain_index), context.poly_degree(), s);
        check(pha_ckks_encode(context.handle(), reinterpret_cast<const double *>(values.data()), values.size(), sparse_slots_,
                              chain_index, scale, destination.data(), s));
        destination.set_chain_index(chain_index);
        destination.set_scale(scale);
    }
    void encode(const PhantomContext &context, const std::vector<double> &values, double scale, PhantomPlaintext &destination,
                size_t chain_index, const hip_stream_wrapper &stream_wrapper = default_stream()) {
        std::vector<std::complex<double>> c(values.begin(), values.end());
        encode(context, c, scale, destination, chain_index, stream_wrapper);
    }
    void decode(const PhantomContext &context, const PhantomPlaintext &plain, std::vector<std::complex<double>> &destination,
                const hip_stream_wrapper &stream_wrapper = default_stream()) {
        if (sparse_slots_ == 0) throw std::logic_error("decode before any encode: sparse_slots unknown");
        destination.resize(sparse_slots_);
        check(pha_ckks_decode(context.handle(), plain.data(), plain.chain_index(), plain.scale(), sparse_slots_,
                              reinterpret_cast<double *>(destination.data()), stream_wrapper.get_stream()));
    }
};

// ---- public key (include/secretkey.h:18-83, src/secretkey.cu:15-164) -------------------------------------
class PhantomPublicKey {
    friend class PhantomSecretKey;
    device_words pk_;  // [2][size_QP][N], NTT form: an encryption of zero at the key level

public:
    PhantomPublicKey() = default;
    PhantomPublicKey(PhantomPublicKey &&) = default;
    PhantomPublicKey &operator=(PhantomPublicKey &&) = default;
    uint64_t *data() const { return pk_.get(); }
    // src/secretkey.cu:133-164: fresh randomness per call (u, e0, e1)
    void encrypt_asymmetric(const PhantomContext &context, const PhantomPlaintext &plain, PhantomCiphertext &cipher,
                            const hip_stream_wrapper &sw = default_stream()) {
        if (!pk_.get()) throw std::logic_error("public key has not been generated");
        const hipStream_t &s = sw.get_stream();
        cipher.resize(context, plain.chain_index(), 2, s);
        uint64_t nonce;   // nothing derived from the key owner's state lives in the public object: a fresh OS-entropy nonce per call
        check(pha_random_nonce(&nonce));
        check(pha_encrypt_asymmetric(context.handle(), pk_.get(), plain.data(), plain.chain_index(), nonce, cipher.data(), s));
        cipher.set_ntt_form(true);
        cipher.set_scale(plain.scale());
    }
};

// ---- secret key (include/secretkey.h:150-290, src/secretkey.cu) -----------------------------------
// Randomness: a ChaCha20 stream under the process's 256-bit sampler key (OS entropy; include/phantom_amd.h).  The 64-bit
// value given at construction (default: std::random_device) only numbers this key's draws — a public nonce base, not a
// secret; with the sampler key fixed by pha_set_sampler_key (tests) runs are reproducible.  Distributions are the
// reference's (src/prng.cu).
class PhantomSecretKey {
    device_words sk_;  // secret_key_array(): [size_QP][N], NTT form
    bool gen_flag_ = false;
    uint64_t seed_, draws_ = 0;

    uint64_t next_seed() {  // distinct nonces: a bijective mix of (construction value, draw number)
        uint64_t z = seed_ + (++draws_) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    PhantomRelinKey make_key(const PhantomContext &context, int kind, uint32_t galois_elt, const hipStream_t &s) {
        const size_t size_QP = context.key_parms().coeff_modulus().size(), n = context.poly_degree();
        if (context.size_P() == 0 || context.size_Q() % context.size_P()) throw std::invalid_argument("dnum is not an integer");
        std::vector<device_words> digits;
        std::vector<uint64_t *> ptrs;
        for (size_t d = 0; d < context.size_Q() / context.size_P(); d++) {
            digits.emplace_back(2 * size_QP * n);
            ptrs.push_back(digits.back().get());
        }
        if (kind == 0)
            check(pha_gen_relinkey(context.handle(), sk_.get(), next_seed(), ptrs.data(), s));
        else
            check(pha_gen_galois_key(context.handle(), sk_.get(), galois_elt, next_seed(), ptrs.data(), s));
        return PhantomRelinKey(std::move(digits));
    }

public:
    explicit PhantomSecretKey(const PhantomContext &context, uint64_t seed = (uint64_t(std::random_device{}()) << 32) ^ std::random_device{}())
        : seed_(seed) {
        gen_secretkey(context, nullptr);
        // generated on the null stream; callers go on to use their own (non-blocking) streams
        check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }
    PhantomSecretKey(PhantomSecretKey &&) = default;
    uint64_t *secret_key_array() const { return sk_.get(); }

    void gen_secretkey(const PhantomContext &context, const hipStream_t &stream) {
        if (gen_flag_) throw std::logic_error("cannot generate secret key twice");
        sk_ = device_words(context.key_parms().coeff_modulus().size() * context.poly_degree());
        const size_t h = context.key_parms().secret_key_hamming_weight();
        if (h)
            check(pha_gen_secretkey_sparse(context.handle(), next_seed(), h, sk_.get(), stream));
        else
            check(pha_gen_secretkey(context.handle(), next_seed(), sk_.get(), stream));
        gen_flag_ = true;
    }
    PhantomRelinKey gen_relinkey(const PhantomContext &context, const hip_stream_wrapper &sw = default_stream()) {
        return make_key(context, 0, 0, sw.get_stream());
    }
    // src/secretkey.cu:437-452
    PhantomPublicKey gen_publickey(const PhantomContext &context, const hip_stream_wrapper &sw = default_stream()) {
        PhantomPublicKey pk;
        pk.pk_ = device_words(2 * context.key_parms().coeff_modulus().size() * context.poly_degree());
        check(pha_gen_publickey(context.handle(), sk_.get(), next_seed(), pk.pk_.get(), sw.get_stream()));
        return pk;
    }
    PhantomGaloisKey create_galois_keys_from_elts(const PhantomContext &context, const std::vector<uint32_t> &elts,
                                                  const hip_stream_wrapper &sw = default_stream()) {
        PhantomGaloisKey gk;
        for (uint32_t e : elts) {
            if (!(e & 1) || e >= 2 * context.poly_degree()) throw std::invalid_argument("Galois element is not valid");
            gk.insert(e, make_key(context, 1, e, sw.get_stream()));
        }
        return gk;
    }
    // add the keys of `elts` that `gk` does not hold yet
    void add_galois_keys_from_elts(const PhantomContext &context, const std::vector<uint32_t> &elts, PhantomGaloisKey &gk,
                                   const hip_stream_wrapper &sw = default_stream()) {
        for (uint32_t e : elts)
            if (!gk.has(e)) gk.insert(e, make_key(context, 1, e, sw.get_stream()));
    }
    PhantomGaloisKey create_galois_keys_from_steps(const PhantomContext &context, const std::vector<int> &steps,
                                                   const hip_stream_wrapper &sw = default_stream()) {
        std::vector<uint32_t> elts;
        for (int st : steps) {
            uint32_t e;
            check(pha_galois_elt_from_step(context.log_n(), st, &e));
            elts.push_back(e);
        }
        return create_galois_keys_from_elts(context, elts, sw);
    }
    // src/secretkey.cu:548-582
    void encrypt_symmetric(const PhantomContext &context, const PhantomPlaintext &plain, PhantomCiphertext &cipher,
                           const hip_stream_wrapper &sw = default_stream()) {
        const hipStream_t &s = sw.get_stream();
        cipher.resize(context, plain.chain_index(), 2, s);
        const uint64_t seed_a = next_seed(), seed_e = next_seed();
        check(pha_encrypt_symmetric(context.handle(), sk_.get(), plain.data(), plain.chain_index(), seed_a, seed_e,
                                    cipher.data(), s));
        cipher.set_ntt_form(true);
        cipher.set_scale(plain.scale());
    }
    // src/secretkey.cu:584-643
    void decrypt(const PhantomContext &context, const PhantomCiphertext &cipher, PhantomPlaintext &plain,
                 const hip_stream_wrapper &sw = default_stream()) {
        if (!cipher.is_ntt_form()) throw std::invalid_argument("encrypted must be in NTT form");
        const hipStream_t &s = sw.get_stream();
        plain.resize(cipher.coeff_modulus_size(), cipher.poly_modulus_degree(), s);
        check(pha_ckks_decrypt(context.handle(), sk_.get(), cipher.data(), cipher.size(), cipher.chain_index(), plain.data(), s));
        plain.set_chain_index(cipher.chain_index());
        plain.set_scale(cipher.scale());
    }
};

// ---- L4 free functions (include/ntt.cuh:47-94) ---------------------------------------------------
inline void nwt_2d_radix8_forward_inplace(uint64_t *inout, const PhantomContext &c, size_t coeff_modulus_size,
                                          size_t start_modulus_idx, const hipStream_t &s) {
    check(pha_nwt_2d_radix8_forward_inplace(c.handle(), inout, coeff_modulus_size, start_modulus_idx, s));
}
inline void nwt_2d_radix8_backward_inplace(uint64_t *inout, const PhantomContext &c, size_t coeff_modulus_size,
                                           size_t start_modulus_idx, const hipStream_t &s) {
    check(pha_nwt_2d_radix8_backward_inplace(c.handle(), inout, coeff_modulus_size, start_modulus_idx, s));
}
inline void nwt_2d_radix8_backward(uint64_t *out, const uint64_t *in, const PhantomContext &c, size_t coeff_modulus_size,
                                   size_t start_modulus_idx, const hipStream_t &s) {
    check(pha_nwt_2d_radix8_backward(c.handle(), out, in, coeff_modulus_size, start_modulus_idx, s));
}
// src/eval_key_switch.cu:91-178
inline void keyswitch_inplace(const PhantomContext &c, PhantomCiphertext &encrypted, uint64_t *c2,
                              const PhantomRelinKey &relin_keys, bool /*is_relin*/, const hipStream_t &s) {
    check(pha_keyswitch_inplace(c.handle(), encrypted.chain_index(), encrypted.data(), c2, relin_keys.public_keys_ptr(), s));
}

// ---- evaluator entry points (include/evaluate.cuh:114-294) ---------------------------------------
// src/evaluate.cu:229-280,387-440
inline void multiply_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                             const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (!(encrypted1.is_ntt_form() && encrypted2.is_ntt_form()))
        throw std::invalid_argument("encrypted1 and encrypted2 must be in NTT form");
    if (encrypted1.chain_index() != encrypted2.chain_index())
        throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    const size_t size1 = encrypted1.size(), size2 = encrypted2.size(), dest_size = size1 + size2 - 1;
    if (size1 < 1 || size2 < 1 || size1 > 4 || size2 > 4) throw std::invalid_argument("ciphertext sizes 1..4 are supported");
    const hipStream_t &s = stream_wrapper.get_stream();
    const size_t limbs = context.coeff_mod_size(encrypted1.chain_index());
    const bool square = &encrypted1 == &encrypted2;
    encrypted1.resize(context, encrypted1.chain_index(), dest_size, s);
    if (dest_size != 3)    // tensor_prod_mxn_rns_poly (src/evaluate.cu:270-275); in place like the reference
        check(pha_tensor_prod_mxn(context.handle(), encrypted1.data(), size1, square ? encrypted1.data() : encrypted2.data(), size2,
                                  limbs, encrypted1.data(), s));
    else if (square)
        check(pha_tensor_square_2x2(context.handle(), encrypted1.data(), limbs, encrypted1.data(), s));
    else
        check(pha_tensor_prod_2x2(context.handle(), encrypted1.data(), encrypted2.data(), limbs, encrypted1.data(), s));
    encrypted1.set_scale(encrypted1.scale() * encrypted2.scale());
}

// src/evaluate.cu:553-583
inline void relinearize_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomRelinKey &relin_keys,
                                const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.size() != 3) throw std::invalid_argument("destination_size must be 3");
    if (!encrypted.is_ntt_form()) throw std::invalid_argument("CKKS encrypted must be in NTT form");
    const hipStream_t &s = stream_wrapper.get_stream();
    const size_t limbs = context.coeff_mod_size(encrypted.chain_index()), n = context.poly_degree();
    keyswitch_inplace(context, encrypted, encrypted.data() + 2 * limbs * n, relin_keys, true, s);
    encrypted.resize(2, limbs, n, s);
}

inline void multiply_and_relin_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1,
                                       const PhantomCiphertext &encrypted2, const PhantomRelinKey &relin_keys,
                                       const hip_stream_wrapper &stream_wrapper = default_stream()) {
    multiply_inplace(context, encrypted1, encrypted2, stream_wrapper);
    relinearize_inplace(context, encrypted1, relin_keys, stream_wrapper);
}

// src/evaluate.cu:585-616,725-745
inline PhantomCiphertext rescale_to_next(const PhantomContext &context, const PhantomCiphertext &encrypted,
                                         const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.chain_index() == context.size_Q()) throw std::invalid_argument("end of modulus switching chain reached");
    const hipStream_t &s = stream_wrapper.get_stream();
    PhantomCiphertext destination;
    destination.resize(context, context.get_next_index(encrypted.chain_index()), encrypted.size(), s);
    check(pha_rescale_to_next(context.handle(), encrypted.chain_index(), encrypted.data(), encrypted.size(),
                              destination.data(), s));
    destination.set_ntt_form(encrypted.is_ntt_form());
    const size_t last = context.coeff_mod_size(encrypted.chain_index()) - 1;
    destination.set_scale(encrypted.scale() / static_cast<double>(context.modulus(last)));
    return destination;
}
inline void rescale_to_next_inplace(const PhantomContext &context, PhantomCiphertext &encrypted,
                                    const hip_stream_wrapper &stream_wrapper = default_stream()) {
    encrypted = rescale_to_next(context, encrypted, stream_wrapper);
}

// src/evaluate.cu:747-794
inline void apply_galois_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, uint32_t galois_elt,
                                 const PhantomGaloisKey &galois_keys, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.size() > 2) throw std::invalid_argument("encrypted size must be 2");
    const PhantomRelinKey &key = galois_keys.get_relin_keys(galois_elt);
    check(pha_apply_galois_inplace(context.handle(), encrypted.chain_index(), encrypted.data(), galois_elt,
                                   key.public_keys_ptr(), stream_wrapper.get_stream()));
}

// complex conjugation = the automorphism X -> X^(2N-1) (include/evaluate.cuh:283-294, src/evaluate.cu:855-870)
inline void complex_conjugate_inplace(const PhantomContext &context, PhantomCiphertext &encrypted,
                                      const PhantomGaloisKey &galois_key, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    apply_galois_inplace(context, encrypted, static_cast<uint32_t>(2 * context.poly_degree() - 1), galois_key, stream_wrapper);
}

// ---- element-wise evaluator entry points (include/evaluate.cuh:33-112,166-256, src/evaluate.cu:20-105,442-560,618-745)
inline void negate_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    check(pha_negate_rns_poly(context.handle(), encrypted.data(), encrypted.data(), encrypted.coeff_modulus_size(),
                              encrypted.size(), stream_wrapper.get_stream()));
}
inline void add_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                        const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted1.chain_index() != encrypted2.chain_index()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    if (encrypted1.is_ntt_form() != encrypted2.is_ntt_form()) throw std::invalid_argument("NTT form mismatch");
    if (encrypted1.size() != encrypted2.size()) throw std::invalid_argument("poly number mismatch");
    check(pha_add_rns_poly(context.handle(), encrypted1.data(), encrypted1.data(), encrypted2.data(),
                           encrypted1.coeff_modulus_size(), encrypted1.size(), encrypted2.size(), stream_wrapper.get_stream()));
}
inline void sub_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                        const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted1.chain_index() != encrypted2.chain_index()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    if (encrypted1.is_ntt_form() != encrypted2.is_ntt_form()) throw std::invalid_argument("NTT form mismatch");
    if (encrypted1.size() != encrypted2.size()) throw std::invalid_argument("poly number mismatch");
    check(pha_sub_rns_poly(context.handle(), encrypted1.data(), encrypted1.data(), encrypted2.data(),
                           encrypted1.coeff_modulus_size(), encrypted1.size(), encrypted2.size(), stream_wrapper.get_stream()));
}
inline void add_plain_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomPlaintext &plain,
                              const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.chain_index() != plain.chain_index()) throw std::invalid_argument("encrypted and plain parameter mismatch");
    check(pha_add_rns_poly(context.handle(), encrypted.data(), encrypted.data(), plain.data(), encrypted.coeff_modulus_size(), 1,
                           1, stream_wrapper.get_stream()));
}
inline void sub_plain_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomPlaintext &plain,
                              const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.chain_index() != plain.chain_index()) throw std::invalid_argument("encrypted and plain parameter mismatch");
    check(pha_sub_rns_poly(context.handle(), encrypted.data(), encrypted.data(), plain.data(), encrypted.coeff_modulus_size(), 1,
                           1, stream_wrapper.get_stream()));
}
inline void multiply_plain_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomPlaintext &plain,
                                   const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.chain_index() != plain.chain_index()) throw std::invalid_argument("encrypted and plain parameter mismatch");
    check(pha_multiply_rns_poly(context.handle(), encrypted.data(), encrypted.data(), plain.data(),
                                encrypted.coeff_modulus_size(), encrypted.size(), 1, stream_wrapper.get_stream()));
    encrypted.set_scale(encrypted.scale() * plain.scale());
}
// CKKS mod_switch_to_next: drop the last prime, no scaling (src/evaluate.cu:618-723)
inline PhantomCiphertext mod_switch_to_next(const PhantomContext &context, const PhantomCiphertext &encrypted,
                                            const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.chain_index() == context.size_Q()) throw std::invalid_argument("end of modulus switching chain reached");
    const hipStream_t &s = stream_wrapper.get_stream();
    PhantomCiphertext destination;
    destination.resize(context, context.get_next_index(encrypted.chain_index()), encrypted.size(), s);
    check(pha_mod_switch_drop_to_next(context.handle(), encrypted.chain_index(), encrypted.data(), encrypted.size(),
                                      destination.data(), s));
    destination.set_scale(encrypted.scale());
    destination.set_ntt_form(encrypted.is_ntt_form());
    return destination;
}
inline void mod_switch_to_next_inplace(const PhantomContext &context, PhantomCiphertext &encrypted,
                                       const hip_stream_wrapper &stream_wrapper = default_stream()) {
    encrypted = mod_switch_to_next(context, encrypted, stream_wrapper);
}
inline void mod_switch_to_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, size_t chain_index,
                                  const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.chain_index() > chain_index) throw std::invalid_argument("cannot switch to higher level modulus");
    while (encrypted.chain_index() != chain_index) mod_switch_to_next_inplace(context, encrypted, stream_wrapper);
}


// ---- out-of-place forms and the remaining entry points of include/evaluate.cuh:33-294 ------------------------
// sub_inplace with the reference's `negate` flag: true computes encrypted1 = encrypted2 - encrypted1
inline void sub_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                        const bool &negate, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    sub_inplace(context, encrypted1, encrypted2, stream_wrapper);
    if (negate) negate_inplace(context, encrypted1, stream_wrapper);
}
inline PhantomCiphertext negate(const PhantomContext &context, const PhantomCiphertext &encrypted,
                                const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    negate_inplace(context, destination, stream_wrapper);
    return destination;
}
inline PhantomCiphertext add(const PhantomContext &context, const PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                             const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted1.clone(stream_wrapper.get_stream());
    add_inplace(context, destination, encrypted2, stream_wrapper);
    return destination;
}
// destination = encrypteds[0] + encrypteds[1] + ...
inline void add_many(const PhantomContext &context, const std::vector<PhantomCiphertext> &encrypteds, PhantomCiphertext &destination,
                     const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypteds.empty()) throw std::invalid_argument("encrypteds cannot be empty");
    PhantomCiphertext sum = encrypteds[0].clone(stream_wrapper.get_stream());
    for (size_t i = 1; i < encrypteds.size(); i++) add_inplace(context, sum, encrypteds[i], stream_wrapper);
    destination = std::move(sum);
}
inline PhantomCiphertext sub(const PhantomContext &context, const PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                             const bool &negate = false, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted1.clone(stream_wrapper.get_stream());
    sub_inplace(context, destination, encrypted2, negate, stream_wrapper);
    return destination;
}
inline PhantomCiphertext multiply(const PhantomContext &context, const PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                                  const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted1.clone(stream_wrapper.get_stream());
    multiply_inplace(context, destination, &encrypted1 == &encrypted2 ? destination : encrypted2, stream_wrapper);
    return destination;
}
inline PhantomCiphertext relinearize(const PhantomContext &context, const PhantomCiphertext &encrypted, const PhantomRelinKey &relin_keys,
                                     const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    relinearize_inplace(context, destination, relin_keys, stream_wrapper);
    return destination;
}
inline PhantomCiphertext multiply_and_relin(const PhantomContext &context, const PhantomCiphertext &encrypted1,
                                            const PhantomCiphertext &encrypted2, const PhantomRelinKey &relin_keys,
                                            const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted1.clone(stream_wrapper.get_stream());
    multiply_and_relin_inplace(context, destination, &encrypted1 == &encrypted2 ? destination : encrypted2, relin_keys, stream_wrapper);
    return destination;
}
inline PhantomCiphertext add_plain(const PhantomContext &context, const PhantomCiphertext &encrypted, const PhantomPlaintext &plain,
                                   const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    add_plain_inplace(context, destination, plain, stream_wrapper);
    return destination;
}
inline PhantomCiphertext sub_plain(const PhantomContext &context, const PhantomCiphertext &encrypted, const PhantomPlaintext &plain,
                                   const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    sub_plain_inplace(context, destination, plain, stream_wrapper);
    return destination;
}
inline PhantomCiphertext multiply_plain(const PhantomContext &context, const PhantomCiphertext &encrypted, const PhantomPlaintext &plain,
                                        const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    multiply_plain_inplace(context, destination, plain, stream_wrapper);
    return destination;
}
inline PhantomCiphertext mod_switch_to(const PhantomContext &context, const PhantomCiphertext &encrypted, size_t chain_index,
                                       const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (encrypted.chain_index() > chain_index) throw std::invalid_argument("cannot switch to higher level modulus");
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    mod_switch_to_inplace(context, destination, chain_index, stream_wrapper);
    return destination;
}
// plaintexts (include/evaluate.cuh:211-256)
inline void mod_switch_to_next_inplace(const PhantomContext &, PhantomPlaintext &plain, const hip_stream_wrapper & = default_stream()) {
    plain.drop_last_limb();
}
inline void mod_switch_to_inplace(const PhantomContext &context, PhantomPlaintext &plain, size_t chain_index,
                                  const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (plain.chain_index() > chain_index) throw std::invalid_argument("cannot switch to higher level modulus");
    while (plain.chain_index() != chain_index) mod_switch_to_next_inplace(context, plain, stream_wrapper);
}
inline PhantomCiphertext apply_galois(const PhantomContext &context, const PhantomCiphertext &encrypted, uint32_t galois_elt,
                                      const PhantomGaloisKey &galois_keys, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    apply_galois_inplace(context, destination, galois_elt, galois_keys, stream_wrapper);
    return destination;
}
inline PhantomCiphertext complex_conjugate(const PhantomContext &context, const PhantomCiphertext &encrypted,
                                           const PhantomGaloisKey &galois_key, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    complex_conjugate_inplace(context, destination, galois_key, stream_wrapper);
    return destination;
}

// non-adjacent form of a rotation step (the reference falls back to it when no key exists for
// the step itself, src/evaluate.cu:821-839)
inline std::vector<int> naf(int value) {
    std::vector<int> res;
    const bool neg = value < 0;
    unsigned v = static_cast<unsigned>(std::abs(value));
    for (int bit = 0; v; ++bit) {
        int d = 0;
        if (v & 1) {
            d = 2 - static_cast<int>(v & 3);  // +1 or -1
            v = static_cast<unsigned>(static_cast<int>(v) - d);
        }
        if (d) res.push_back((neg ? -d : d) * (1 << bit));
        v >>= 1;
    }
    return res;
}

// src/evaluate.cu:797-852
inline void rotate_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, int step,
                           const PhantomGaloisKey &galois_key, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    if (step == 0) return;
    uint32_t elt = 0;
    check(pha_galois_elt_from_step(context.log_n(), step, &elt));
    if (galois_key.has(elt)) {
        apply_galois_inplace(context, encrypted, elt, galois_key, stream_wrapper);
        return;
    }
    const std::vector<int> parts = naf(step);
    if (parts.size() == 1) throw std::invalid_argument("Galois key not present");
    for (int part : parts) {
        if (static_cast<size_t>(std::abs(part)) != (context.poly_degree() >> 1))
            rotate_inplace(context, encrypted, part, galois_key, stream_wrapper);
    }
}

inline PhantomCiphertext rotate(const PhantomContext &context, const PhantomCiphertext &encrypted, int step,
                                const PhantomGaloisKey &galois_key, const hip_stream_wrapper &stream_wrapper = default_stream()) {
    PhantomCiphertext destination = encrypted.clone(stream_wrapper.get_stream());
    rotate_inplace(context, destination, step, galois_key, stream_wrapper);
    return destination;
}

}  // namespace phantom
