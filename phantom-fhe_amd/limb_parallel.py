"""Limb-parallel HMult + relinearise, rotation and rescale across the GPUs of one node.

The reference is single-GPU; its key switch (src/eval_key_switch.cu:91-178) is independent per
RNS limb everywhere except the two fast base conversions, which need every input limb of a digit
per coefficient.  With the limbs of Q_l u P partitioned over the ranks (one process per GPU,
``torch.distributed`` over RCCL/xGMI) that leaves exactly two exchanges per key switch and one per
rescale:

    tensor        d0, d1, d2 on the owned limbs                      (HMult only)
    modup_begin   INTT + scale of the owned Q limbs of c2
    ------------  all-gather #1: coefficient-form c2, [l][N] in total
    modup_finish  base conversion + NTT of the owned limbs of every digit
    inner_prod    owned limbs of (cx_b, cx_a)            (evk only read on the owned limbs)
    moddown_begin INTT + scale of the owned P limbs of cx
    ------------  all-gather #2: coefficient-form P limbs of cx, [2][alpha][N] in total
    moddown_finish  ct[owned] += (cx - NTT(bconv)) * P^-1

Each rank keeps buffers in the single-GPU layouts and reads/writes only its limb windows (the
288 GB of HBM make the unused rows irrelevant), so the per-rank steps are the C-ABI calls
``pha_lp_*`` and a ciphertext stays limb-sharded between operations.  Results on the owned limbs
are bit-identical to the single-GPU path.

Ownership is BALANCED over Q_l u P: rank r owns q_r = floor(l / G) + [r < l mod G] consecutive Q limbs and
p_r = t_r - q_r consecutive special limbs, t_r = floor((l + alpha) / G) + [r < (l + alpha) mod G], so every rank carries
ceil((l + alpha) / G) limbs or one less (36 + 12 limbs on 8 ranks: 5 + 1 on ranks 0-3, 4 + 2 on ranks 4-7), and both
kinds of limb are spread over all ranks, which keeps every all-gather block small (a rank holding all of P would have
to send it whole over each of its links).
Exchange buffers are rank-blocked and padded: rank r's block is rows [r * b, (r + 1) * b) with b the largest window, so
every rank's ``all_gather_into_tensor`` input IS block `rank` of the output (in place, no packing) and the consumers
read "limb -> row" through a table (q_rows, p_rows): c2_coeff [G * bq][N]; p_coeff [G][2][bp][N] with both polys of a
rank in one block, so the moddown needs ONE all-gather.  A key switch is two collectives and nothing between them can
overlap them: each feeds the step right behind it (base conversion needs every input limb).
Exchange buffers are allocated once per (level, shape) and reused.  Messages are small by xGMI standards (at N=2^16, l=36,
alpha=12, G=8: blocks of 2.5 MiB and 2 MiB), so one all-gather per exchange is the right granularity.
"""
import threading

import torch


def block(total, parts, index):
    """Balanced partition of range(total): (begin, count, largest count) of part `index` (the first total % parts parts
    hold one more)."""
    base, extra = divmod(total, parts)
    lo = index * base + min(index, extra)
    return lo, base + (1 if index < extra else 0), base + (1 if extra else 0)


def limb_windows(size_ql, alpha, world):
    """[(q_begin, q_count, p_begin, p_count)] per rank at a level with size_ql Q limbs: balanced totals, see above."""
    wins, p0 = [], 0
    for r in range(world):
        q0, qc, _ = block(size_ql, world, r)
        pc = block(size_ql + alpha, world, r)[1] - qc
        wins.append((q0, qc, p0, pc))
        p0 += pc
    assert p0 == alpha and all(w[3] >= 0 for w in wins)
    return wins


def limb_window(size_ql, alpha, world, rank):
    return limb_windows(size_ql, alpha, world)[rank]


class TorchDistComm:
    """Exchange over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" is staged through host memory and is
    there for tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.staged = dist.get_backend(group) == "gloo"

    def all_gather_blocks(self, buf, rows):
        """buf [world * rows][...]: this rank's block (rows [rank*rows, (rank+1)*rows)) goes to every rank; in place."""
        if self.world == 1:
            return
        # the in-place form relies on block k sitting at element offset k * rows * row_size of one contiguous buffer
        assert buf.is_contiguous() and buf.shape[0] == self.world * rows, (tuple(buf.shape), self.world, rows)
        k = self.rank
        if self.staged and buf.is_cuda:
            h = buf.cpu()
            self.dist.all_gather_into_tensor(h, h[k * rows:(k + 1) * rows].clone(), group=self.group)
            buf.copy_(h)
        else:
            self.dist.all_gather_into_tensor(buf, buf[k * rows:(k + 1) * rows], group=self.group)

    def broadcast(self, t, src):
        gsrc = self.dist.get_global_rank(self.group, src) if self.group else src
        if self.staged and t.is_cuda:
            h = t.cpu()
            self.dist.broadcast(h, src=gsrc, group=self.group)
            t.copy_(h)
        else:
            self.dist.broadcast(t, src=gsrc, group=self.group)


class ThreadComm:
    """In-process stand-in: `world` threads, one per simulated rank, sharing one GPU and its
    default stream (tests on a one-GPU box).  Same interface as TorchDistComm."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.slots = [None] * world
            self.barrier = threading.Barrier(world, timeout=120)  # a failed rank breaks the barrier for all

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def all_gather_blocks(self, buf, rows):
        if self.world == 1:
            return
        assert buf.is_contiguous() and buf.shape[0] == self.world * rows, (tuple(buf.shape), self.world, rows)
        s = self.shared
        k = self.rank
        s.slots[k] = buf[k * rows:(k + 1) * rows].clone()
        s.barrier.wait()
        for j in range(self.world):
            if j != k:
                buf[j * rows:(j + 1) * rows] = s.slots[j]
        s.barrier.wait()

    def broadcast(self, t, src):
        s = self.shared
        if self.rank == src:
            s.slots[src] = t
        s.barrier.wait()
        if self.rank != src:
            t.copy_(s.slots[src])
        s.barrier.wait()


class LimbParallelEvaluator:
    """Per-rank driver of the limb-parallel steps.  `ctx` is this rank's PhantomContext (same
    moduli on every rank); tensors are int64 device tensors in the single-GPU layouts of which
    only the owned limbs need to be valid on entry and are valid on exit."""

    def __init__(self, ctx, comm):
        self.ctx, self.comm = ctx, comm
        self.rank, self.world = comm.rank, comm.world
        self._bufs = {}

    def windows(self, chain_index):
        return limb_windows(self.ctx.size_Ql(chain_index), self.ctx.size_P, self.world)

    def _buffers(self, chain_index, device):
        """Exchange and scratch buffers of a level, allocated once: c2_coeff [G*bq][N], p_coeff [G*2*bp][N] (rank-blocked,
        both polys of a rank in its block), their "limb -> row" tables, t_mod_up, cx."""
        key = (chain_index, str(device))
        b = self._bufs.get(key)
        if b is None:
            ctx = self.ctx
            size_ql, alpha, n = ctx.size_Ql(chain_index), ctx.size_P, ctx.n
            wins = self.windows(chain_index)
            bq, bp = max(w[1] for w in wins), max(w[3] for w in wins)
            mk = lambda *shape: torch.empty(shape, dtype=torch.int64, device=device)  # noqa: E731
            q_rows = [r * bq + i for r, w in enumerate(wins) for i in range(w[1])]
            p_rows = [r * 2 * bp + i for r, w in enumerate(wins) for i in range(w[3])]
            assert len(q_rows) == size_ql and len(p_rows) == alpha
            b = {"bq": bq, "bp": bp, "q_rows": q_rows, "p_rows": p_rows, "c2_coeff": mk(self.world * bq, n),
                 "p_coeff": mk(self.world * 2 * bp, n),
                 "t_mod_up": mk(ctx.beta(chain_index), size_ql + alpha, n), "cx": mk(2, size_ql + alpha, n)}
            self._bufs[key] = b
        return b

    def keyswitch_inplace(self, chain_index, ct, c2, evk):
        """ct [2][l][N] += key switch of c2 [l][N] under evk, on the owned limbs
        (src/eval_key_switch.cu:91-178)."""
        ctx = self.ctx
        w = self.windows(chain_index)[self.rank]
        b = self._buffers(chain_index, ct.device)
        bq, bp = b["bq"], b["bp"]
        ctx.lp_modup_begin(chain_index, c2, b["c2_coeff"], w, row_begin=self.rank * bq)
        self.comm.all_gather_blocks(b["c2_coeff"], bq)
        ctx.lp_modup_finish(chain_index, b["t_mod_up"], c2, b["c2_coeff"], w, q_rows=b["q_rows"])
        ctx.lp_key_switch_inner_prod(chain_index, b["cx"], b["t_mod_up"], evk, w)
        ctx.lp_moddown_begin(chain_index, b["cx"], b["p_coeff"], w, row_begin=self.rank * 2 * bp, poly_stride=bp)
        self.comm.all_gather_blocks(b["p_coeff"], 2 * bp)       # both polys' P limbs in one collective
        ctx.lp_moddown_finish(chain_index, ct, b["cx"], b["p_coeff"], w, p_rows=b["p_rows"], poly_stride=bp)

    def multiply_and_relin(self, chain_index, ct1, ct2, evk, out3):
        """HMult + relinearise on the owned limbs (src/evaluate.cu:229-280,553-583): out3 [3][l][N] receives the tensor
        product, its first two polys then take the key-switched third; returns out3[:2]."""
        w = self.windows(chain_index)[self.rank]
        self.ctx.lp_tensor_prod_2x2(ct1, ct2, out3, w)
        self.keyswitch_inplace(chain_index, out3[:2], out3[2], evk)
        return out3[:2]

    def apply_galois_inplace(self, chain_index, ct, galois_elt, galois_key):
        """Rotation on the owned limbs (src/evaluate.cu:747-794): permute c0, c1 in the NTT
        domain, key-switch the permuted c1."""
        ctx = self.ctx
        q0, qc, _, _ = self.windows(chain_index)[self.rank]
        c1 = torch.empty_like(ct[1])
        tmp = torch.empty_like(ct[0])
        if qc:
            ctx.apply_galois_ntt(ct[0][q0:q0 + qc], qc, galois_elt, tmp[q0:q0 + qc])
            ctx.apply_galois_ntt(ct[1][q0:q0 + qc], qc, galois_elt, c1[q0:q0 + qc])
            ct[0][q0:q0 + qc] = tmp[q0:q0 + qc]
            ct[1][q0:q0 + qc] = 0
        self.keyswitch_inplace(chain_index, ct, c1, galois_key)

    def rescale_to_next(self, chain_index, src, cipher_size, dst):
        """dst [cipher_size][l-1][N] from src [cipher_size][l][N] on the owned limbs
        (src/rns.cu:1157-1181): the owner of the last limb inverts it and broadcasts it."""
        ctx = self.ctx
        size_ql = ctx.size_Ql(chain_index)
        wins = self.windows(chain_index)
        owner = next(r for r, (q0, qc, _, _) in enumerate(wins) if q0 <= size_ql - 1 < q0 + qc)
        last = torch.empty((cipher_size, ctx.n), dtype=torch.int64, device=src.device)
        if self.rank == owner:
            ctx.lp_rescale_begin(chain_index, src, cipher_size, last)
        self.comm.broadcast(last, owner)
        ctx.lp_rescale_finish(chain_index, src, last, cipher_size, dst, wins[self.rank])

    def gather_limbs(self, chain_index, polys, size_ql=None):
        """All-gather the owned Q windows of the given [limbs][N] polys so that every rank holds
        them whole (hand-off to a single-GPU consumer, tests).  size_ql: limb count of `polys` if
        it differs from the level's (after a rescale the partition is still the source level's)."""
        wins = self.windows(chain_index)
        bq = max(w[1] for w in wins)
        limit = polys[0].shape[0] if size_ql is None else size_ql
        q0, qc = wins[self.rank][:2]
        mine = max(0, min(qc, limit - q0))
        for pl in polys:
            pad = torch.zeros((self.world * bq, pl.shape[-1]), dtype=pl.dtype, device=pl.device)
            pad[self.rank * bq:self.rank * bq + mine] = pl[q0:q0 + mine]
            self.comm.all_gather_blocks(pad, bq)
            for r, (r0, rc, _, _) in enumerate(wins):
                cnt = max(0, min(rc, limit - r0))
                pl[r0:r0 + cnt] = pad[r * bq:r * bq + cnt]
