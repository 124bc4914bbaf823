// Limb-parallel HMult + relinearise, rotation and rescale across the GPUs of one node, C++ host layer over RCCL
// (the twin of phantom-fhe_amd/limb_parallel.py, which drives the same C-ABI steps through
// torch.distributed).  No counterpart in the reference, which is single-GPU: this is its keyswitch_inplace
// (src/eval_key_switch.cu:91-178) and rescale (src/rns.cu:1157-1181) with the limbs of Q_l u P partitioned over the ranks.
//
// One process per GPU, every rank with its own PhantomContext built from the same moduli.  Buffers keep the single-GPU
// layouts and a ciphertext stays limb-sharded between operations: only the owned limbs are valid on entry and on exit.
//   ownership  balanced over Q_l u P: rank r owns floor(l / G) + [r < l mod G] consecutive Q limbs and as many special
//              limbs as bring it to ceil((l + alpha) / G) or one less (limb_windows)
//   exchanges  rank-blocked buffers padded to the largest window, so that every rank's ncclAllGather input IS block
//              `rank` of the output (in place); consumers read "limb -> row" through host tables.  A key switch is two
//              collectives: c2 in coefficient form before the modup conversion, both polys' special limbs before the
//              moddown conversion; a rescale is one broadcast.
// The communicator is a policy: RcclComm (ncclComm_t; xGMI inside a node) or LocalComm (one rank, no library).
#pragma once

#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <fstream>
#include <map>
#include <thread>

#include "phantom_amd.hpp"

namespace phantom {

struct LimbWindow {
    size_t q_begin, q_count, p_begin, p_count;
};

// balanced partition of range(total): (begin, count) of part `index`; the first total % parts parts hold one more
inline std::pair<size_t, size_t> balanced_block(size_t total, size_t parts, size_t index) {
    const size_t base = total / parts, extra = total % parts;
    return {index * base + std::min(index, extra), base + (index < extra ? 1 : 0)};
}

inline std::vector<LimbWindow> limb_windows(size_t size_ql, size_t alpha, size_t world) {
    std::vector<LimbWindow> w(world);
    size_t p0 = 0;
    for (size_t r = 0; r < world; r++) {
        const auto q = balanced_block(size_ql, world, r);
        const size_t pc = balanced_block(size_ql + alpha, world, r).second - q.second;
        w[r] = LimbWindow{q.first, q.second, p0, pc};
        p0 += pc;
    }
    if (p0 != alpha) throw std::logic_error("limb_windows: the special limbs are not covered");
    return w;
}

inline void check_nccl(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

// One rank, no exchange: the path costs what the single-GPU entry points cost.
class LocalComm {
public:
    int rank() const { return 0; }
    int world() const { return 1; }
    void all_gather_blocks(uint64_t *, size_t, hipStream_t) {}
    void broadcast(uint64_t *, size_t, int, hipStream_t) {}
};

// RCCL communicator of the node's ranks.  all_gather_blocks is in place: this rank's block of `words` words sits at
// buf + rank * words of a buffer of world * words words.
class RcclComm {
    ncclComm_t comm_ = nullptr;
    int rank_ = 0, world_ = 1;

public:
    RcclComm(int rank, int world, const ncclUniqueId &id) : rank_(rank), world_(world) {
        check_nccl(ncclCommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
    }
    // Ranks started by a launcher that sets RANK / WORLD_SIZE (torchrun, mpirun wrappers): rank 0 writes the unique id to
    // `id_file` (a path every rank of the node can read), the others wait for it.  The caller has already made its GPU
    // current (hipSetDevice(LOCAL_RANK)).
    static RcclComm from_env(const std::string &id_file) {
        const char *r = getenv("RANK"), *w = getenv("WORLD_SIZE");
        const int rank = r ? atoi(r) : 0, world = w ? atoi(w) : 1;
        ncclUniqueId id;
        if (rank == 0) {
            check_nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
            std::ofstream tmp(id_file + ".tmp", std::ios::binary);
            tmp.write(reinterpret_cast<const char *>(&id), sizeof(id));
            tmp.close();
            if (std::rename((id_file + ".tmp").c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot publish the RCCL id");
        } else {
            for (int tries = 0;; tries++) {
                std::ifstream in(id_file, std::ios::binary);
                if (in && in.read(reinterpret_cast<char *>(&id), sizeof(id))) break;
                if (tries > 600) throw std::runtime_error("no RCCL id at " + id_file);
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        return RcclComm(rank, world, id);
    }
    ~RcclComm() {
        if (comm_) (void)ncclCommDestroy(comm_);
    }
    RcclComm(const RcclComm &) = delete;
    RcclComm &operator=(const RcclComm &) = delete;
    RcclComm(RcclComm &&o) noexcept : comm_(o.comm_), rank_(o.rank_), world_(o.world_) { o.comm_ = nullptr; }

    int rank() const { return rank_; }
    int world() const { return world_; }
    void all_gather_blocks(uint64_t *buf, size_t words, hipStream_t s) {
        check_nccl(ncclAllGather(buf + static_cast<size_t>(rank_) * words, buf, words, ncclUint64, comm_, s), "ncclAllGather");
    }
    void broadcast(uint64_t *buf, size_t words, int root, hipStream_t s) {
        check_nccl(ncclBroadcast(buf, buf, words, ncclUint64, root, comm_, s), "ncclBroadcast");
    }
};

template <class Comm>
class LimbParallelEvaluator {
    const PhantomContext &ctx_;
    Comm &comm_;
    struct Level {
        std::vector<LimbWindow> wins;
        size_t bq = 0, bp = 0;
        std::vector<uint32_t> q_rows, p_rows;
        device_words c2_coeff, p_coeff, t_mod_up, cx;
    };
    std::map<size_t, Level> levels_;

    Level &level(size_t chain_index) {
        auto it = levels_.find(chain_index);
        if (it != levels_.end()) return it->second;
        Level lv;
        const size_t l = ctx_.coeff_mod_size(chain_index), alpha = ctx_.size_P(), n = ctx_.poly_degree(), G = comm_.world();
        lv.wins = limb_windows(l, alpha, G);
        for (const auto &w : lv.wins) lv.bq = std::max(lv.bq, w.q_count), lv.bp = std::max(lv.bp, w.p_count);
        for (size_t r = 0; r < G; r++) {
            for (size_t i = 0; i < lv.wins[r].q_count; i++) lv.q_rows.push_back(static_cast<uint32_t>(r * lv.bq + i));
            for (size_t i = 0; i < lv.wins[r].p_count; i++) lv.p_rows.push_back(static_cast<uint32_t>(r * 2 * lv.bp + i));
        }
        const size_t beta = (l + alpha - 1) / alpha;
        lv.c2_coeff = device_words(G * lv.bq * n);
        lv.p_coeff = device_words(std::max<size_t>(1, G * 2 * lv.bp) * n);
        lv.t_mod_up = device_words(beta * (l + alpha) * n);
        lv.cx = device_words(2 * (l + alpha) * n);
        return levels_.emplace(chain_index, std::move(lv)).first->second;
    }

public:
    LimbParallelEvaluator(const PhantomContext &context, Comm &comm) : ctx_(context), comm_(comm) {}

    const LimbWindow &window(size_t chain_index) { return level(chain_index).wins[comm_.rank()]; }
    const std::vector<LimbWindow> &windows(size_t chain_index) { return level(chain_index).wins; }

    // ct [2][l][N] += key switch of c2 [l][N] under the key, on the owned limbs (src/eval_key_switch.cu:91-178)
    void keyswitch_inplace(size_t chain_index, uint64_t *ct, const uint64_t *c2, const PhantomRelinKey &key, const hipStream_t &s) {
        Level &lv = level(chain_index);
        const LimbWindow &w = lv.wins[comm_.rank()];
        const size_t n = ctx_.poly_degree(), G = comm_.world(), rank = comm_.rank();
        pha_context *h = ctx_.handle();
        check(pha_lp_modup_begin(h, chain_index, c2, lv.c2_coeff.get(), G * lv.bq, rank * lv.bq, w.q_begin, w.q_count, s));
        comm_.all_gather_blocks(lv.c2_coeff.get(), lv.bq * n, s);
        check(pha_lp_modup_finish(h, chain_index, lv.t_mod_up.get(), c2, lv.c2_coeff.get(), G * lv.bq, lv.q_rows.data(), w.q_begin,
                                  w.q_count, w.p_begin, w.p_count, s));
        check(pha_lp_key_switch_inner_prod(h, chain_index, lv.cx.get(), lv.t_mod_up.get(), key.public_keys_ptr(), w.q_begin, w.q_count,
                                           w.p_begin, w.p_count, s));
        check(pha_lp_moddown_begin(h, chain_index, lv.cx.get(), lv.p_coeff.get(), G * 2 * lv.bp, lv.bp, rank * 2 * lv.bp, w.p_begin,
                                   w.p_count, s));
        comm_.all_gather_blocks(lv.p_coeff.get(), 2 * lv.bp * n, s);     // both polys' special limbs in one collective
        check(pha_lp_moddown_finish(h, chain_index, ct, lv.cx.get(), lv.p_coeff.get(), G * 2 * lv.bp, lv.bp, lv.p_rows.data(),
                                    w.q_begin, w.q_count, s));
    }

    // HMult + relinearise on the owned limbs (src/evaluate.cu:229-280,553-583): out3 [3][l][N] receives the tensor product,
    // its first two polys then take the key-switched third
    void multiply_and_relin(size_t chain_index, const uint64_t *ct1, const uint64_t *ct2, const PhantomRelinKey &key, uint64_t *out3,
                            const hipStream_t &s) {
        const LimbWindow &w = window(chain_index);
        const size_t l = ctx_.coeff_mod_size(chain_index), n = ctx_.poly_degree();
        check(pha_lp_tensor_prod_2x2(ctx_.handle(), ct1, ct2, l, w.q_begin, w.q_count, out3, s));
        keyswitch_inplace(chain_index, out3, out3 + 2 * l * n, key, s);
    }

    // rotation on the owned limbs (src/evaluate.cu:747-794): c0, c1 permuted in the NTT domain, the permuted c1 key-switched;
    // scratch [2][l][N]
    void apply_galois_inplace(size_t chain_index, uint64_t *ct, uint32_t galois_elt, const PhantomRelinKey &key, uint64_t *scratch,
                              const hipStream_t &s) {
        const LimbWindow &w = window(chain_index);
        const size_t l = ctx_.coeff_mod_size(chain_index), n = ctx_.poly_degree(), off = w.q_begin * n, bytes = w.q_count * n * 8;
        uint64_t *tmp = scratch, *c1 = scratch + l * n;
        if (w.q_count) {
            check(pha_apply_galois_ntt(ctx_.handle(), ct + off, w.q_count, galois_elt, tmp + off, s));
            check(pha_apply_galois_ntt(ctx_.handle(), ct + l * n + off, w.q_count, galois_elt, c1 + off, s));
            check_hip(hipMemcpyAsync(ct + off, tmp + off, bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
            check_hip(hipMemsetAsync(ct + l * n + off, 0, bytes, s), "hipMemsetAsync");
        }
        keyswitch_inplace(chain_index, ct, c1, key, s);
    }

    // dst [cipher_size][l-1][N] from src [cipher_size][l][N] on the owned limbs (src/rns.cu:1157-1181): the owner of the
    // last limb inverts it and broadcasts it; last_coeff: scratch [cipher_size][N]
    void rescale_to_next(size_t chain_index, const uint64_t *src, size_t cipher_size, uint64_t *dst, uint64_t *last_coeff,
                         const hipStream_t &s) {
        Level &lv = level(chain_index);
        const size_t l = ctx_.coeff_mod_size(chain_index), n = ctx_.poly_degree();
        int owner = 0;
        for (size_t r = 0; r < lv.wins.size(); r++)
            if (lv.wins[r].q_begin <= l - 1 && l - 1 < lv.wins[r].q_begin + lv.wins[r].q_count) owner = static_cast<int>(r);
        if (comm_.rank() == owner) check(pha_lp_rescale_begin(ctx_.handle(), chain_index, src, cipher_size, last_coeff, s));
        comm_.broadcast(last_coeff, cipher_size * n, owner, s);
        const LimbWindow &w = lv.wins[comm_.rank()];
        check(pha_lp_rescale_finish(ctx_.handle(), chain_index, src, last_coeff, cipher_size, dst, w.q_begin, w.q_count, s));
    }
};

}  // namespace phantom
