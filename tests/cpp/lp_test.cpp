// C++ host layer of the limb-parallel key switch (phantom-fhe_amd/host/limb_parallel.hpp) against the single-GPU entry
// points on the same inputs, bit for bit: HMult + relinearise, a rotation and a rescale, once with LocalComm and once
// through a real RCCL communicator.  A one-GPU box allows one RCCL rank (two ranks on one device are refused), so the
// collectives run with world size 1 here; more ranks: start one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK set
// and pass the path of a file for the RCCL id as argv[1].
#include <cstdio>
#include <cstring>

#include "../../phantom-fhe_amd/host/limb_parallel.hpp"

using namespace phantom;

static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static device_words uniform(const std::vector<uint64_t> &moduli, size_t polys, size_t limbs, size_t n, uint64_t seed) {
    std::vector<uint64_t> v(polys * limbs * n);
    uint64_t s = seed;
    for (size_t p = 0; p < polys; p++)
        for (size_t i = 0; i < limbs; i++)
            for (size_t k = 0; k < n; k++) v[(p * limbs + i) * n + k] = splitmix64(s) % moduli[i];
    device_words d(v.size());
    check_hip(hipMemcpy(d.get(), v.data(), v.size() * 8, hipMemcpyHostToDevice), "upload");
    return d;
}
static std::vector<uint64_t> download(const uint64_t *p, size_t words) {
    std::vector<uint64_t> h(words);
    check_hip(hipDeviceSynchronize(), "sync");
    check_hip(hipMemcpy(h.data(), p, words * 8, hipMemcpyDeviceToHost), "download");
    return h;
}
// equality on the limbs [q0, q0 + qc) of `polys` polys of l limbs
static bool same_window(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, size_t polys, size_t l, size_t n, size_t q0, size_t qc) {
    for (size_t p = 0; p < polys; p++)
        if (qc && memcmp(&a[(p * l + q0) * n], &b[(p * l + q0) * n], qc * n * 8) != 0) return false;
    return true;
}

template <class Comm>
static int run(const PhantomContext &ctx, const std::vector<uint64_t> &moduli, Comm &comm, const char *name) {
    const size_t chain = 1, l = ctx.coeff_mod_size(chain), n = ctx.poly_degree(), qp = moduli.size();
    const hipStream_t s = nullptr;
    const size_t alpha = ctx.size_P(), beta = (l + alpha - 1) / alpha;
    device_words a = uniform(moduli, 2, l, n, 1), b = uniform(moduli, 2, l, n, 2);
    std::vector<device_words> digits;
    for (size_t d = 0; d < beta; d++) digits.push_back(uniform(moduli, 2, qp, n, 100 + d));
    PhantomRelinKey key(std::move(digits));
    LimbParallelEvaluator<Comm> ev(ctx, comm);
    const LimbWindow w = ev.window(chain);
    int bad = 0;

    // HMult + relinearise
    device_words want(3 * l * n), got(3 * l * n);
    check_hip(hipMemcpy(want.get(), a.get(), 2 * l * n * 8, hipMemcpyDeviceToDevice), "copy");
    check(pha_multiply_and_relin_inplace(ctx.handle(), chain, want.get(), b.get(), key.public_keys_ptr(), s));
    ev.multiply_and_relin(chain, a.get(), b.get(), key, got.get(), s);
    if (!same_window(download(want.get(), 2 * l * n), download(got.get(), 2 * l * n), 2, l, n, w.q_begin, w.q_count)) bad++, printf("%s: HMult + relinearise differs\n", name);

    // rotation by 3
    uint32_t elt = 0;
    check(pha_galois_elt_from_step(ctx.log_n(), 3, &elt));
    device_words rw(2 * l * n), rg(2 * l * n), scratch(2 * l * n);
    check_hip(hipMemcpy(rw.get(), a.get(), 2 * l * n * 8, hipMemcpyDeviceToDevice), "copy");
    check_hip(hipMemcpy(rg.get(), a.get(), 2 * l * n * 8, hipMemcpyDeviceToDevice), "copy");
    check(pha_apply_galois_inplace(ctx.handle(), chain, rw.get(), elt, key.public_keys_ptr(), s));
    ev.apply_galois_inplace(chain, rg.get(), elt, key, scratch.get(), s);
    if (!same_window(download(rw.get(), 2 * l * n), download(rg.get(), 2 * l * n), 2, l, n, w.q_begin, w.q_count)) bad++, printf("%s: rotation differs\n", name);

    // rescale
    device_words sw(2 * (l - 1) * n), sg(2 * (l - 1) * n), last(2 * n);
    check(pha_rescale_to_next(ctx.handle(), chain, a.get(), 2, sw.get(), s));
    check_hip(hipMemset(sg.get(), 0, 2 * (l - 1) * n * 8), "memset");
    ev.rescale_to_next(chain, a.get(), 2, sg.get(), last.get(), s);
    const size_t rq = std::min(w.q_count, l - 1 > w.q_begin ? l - 1 - w.q_begin : 0);
    if (!same_window(download(sw.get(), 2 * (l - 1) * n), download(sg.get(), 2 * (l - 1) * n), 2, l - 1, n, w.q_begin, rq)) bad++, printf("%s: rescale differs\n", name);
    printf("%s: rank %d of %d, window Q[%zu,+%zu) P[%zu,+%zu): %s\n", name, comm.rank(), comm.world(), w.q_begin, w.q_count, w.p_begin, w.p_count,
           bad ? "FAILED" : "ok");
    return bad;
}

int main(int argc, char **argv) {
    try {
        const char *lr = getenv("LOCAL_RANK");
        check_hip(hipSetDevice(lr ? atoi(lr) : 0), "hipSetDevice");
        const size_t n = 8192;
        EncryptionParameters parms(0);
        parms.set_poly_modulus_degree(n);
        const std::vector<uint64_t> moduli = arith::CoeffModulus::Create(n, {55, 45, 45, 45, 45, 45, 45, 55, 55, 55});
        parms.set_coeff_modulus(moduli);
        parms.set_special_modulus_size(3);
        PhantomContext ctx(parms);
        int bad = 0;
        // windows of an 8-rank split tile the basis and are balanced
        const auto wins = limb_windows(36, 12, 8);
        size_t q = 0, p = 0;
        for (const auto &w : wins) {
            if (w.q_begin != q || w.p_begin != p || w.q_count + w.p_count != 6) bad++;
            q += w.q_count, p += w.p_count;
        }
        if (q != 36 || p != 12) bad++;
        if (bad) printf("limb_windows(36, 12, 8) is wrong\n");
        if (argc > 1) {   // one process per GPU, started by a launcher
            RcclComm comm = RcclComm::from_env(argv[1]);
            bad += run(ctx, moduli, comm, "rccl");
        } else {
            LocalComm local;
            bad += run(ctx, moduli, local, "local");
            ncclUniqueId id;
            check_nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
            RcclComm comm(0, 1, id);
            bad += run(ctx, moduli, comm, "rccl");
        }
        check_hip(hipDeviceSynchronize(), "sync");
        printf(bad ? "lp_test FAILED\n" : "lp_test ok\n");
        return bad ? 1 : 0;
    } catch (const std::exception &e) {
        printf("lp_test exception: %s\n", e.what());
        return 2;
    }
}
