"""Limb-parallel key switch and rescale across the GPUs of one node.

The reference is single-GPU; its key switch (src/eval_key_switch.cu:91-178) is independent per
RNS limb everywhere except the two fast base conversions, which need every input limb of a digit
per coefficient.  With the limbs of Q_l u P partitioned over the ranks (one process per GPU,
``torch.distributed`` over RCCL/xGMI) that leaves exactly two exchanges per key switch and one per
rescale:

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

Windows are contiguous: rank r owns Q limbs ``split(l, G, r)`` and special limbs
``split(alpha, G, G-1-r)`` (reversed so that the ranks with one Q limb fewer take the extra P
limb).  Messages are tiny by xGMI standards (18 MiB + 12 MiB in total at N=2^16, l=36, alpha=12),
so one all-gather per exchange, padded to the largest window, is the right granularity.
"""
import threading

import torch


def split(total, parts, index):
    """Contiguous block partition of range(total): (begin, count) of block `index`."""
    base, rem = divmod(total, parts)
    return index * base + min(index, rem), base + (1 if index < rem else 0)


def limb_window(size_ql, alpha, world, rank):
    """(q_begin, q_count, p_begin, p_count) owned by `rank` at a level with size_ql Q limbs."""
    q0, qc = split(size_ql, world, rank)
    p0, pc = split(alpha, world, world - 1 - rank)
    return q0, qc, p0, pc


class TorchDistComm:
    """Exchange over a torch.distributed process group (backend "nccl" = RCCL on ROCm; "gloo" is
    staged through host memory and is there for tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.staged = dist.get_backend(group) == "gloo"

    def all_gather(self, t):
        send = t.cpu() if self.staged and t.is_cuda else t
        outs = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(outs, send.contiguous(), group=self.group)
        return [o.to(t.device) for o in outs] if send is not t else outs

    def broadcast(self, t, src):
        if self.staged and t.is_cuda:
            h = t.cpu()
            self.dist.broadcast(h, src=self.dist.get_global_rank(self.group, src) if self.group else src, group=self.group)
            t.copy_(h)
        else:
            self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, src) if self.group else src, group=self.group)


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

    def all_gather(self, t):
        s = self.shared
        s.slots[self.rank] = t
        s.barrier.wait()
        outs = [x.clone() for x in s.slots]
        s.barrier.wait()
        return outs

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

    def windows(self, chain_index):
        size_ql = self.ctx.size_Ql(chain_index)
        return [limb_window(size_ql, self.ctx.size_P, self.world, r) for r in range(self.world)]

    def _exchange(self, views, wins):
        """views: 2-D tensors [limbs][N] sharing the per-rank windows wins[r] = (begin, count);
        after the call every view holds every rank's window."""
        n = views[0].shape[-1]
        maxc = max(c for _, c in wins)
        if maxc == 0:
            return
        b, c = wins[self.rank]
        send = torch.zeros((len(views), maxc, n), dtype=views[0].dtype, device=views[0].device)
        for i, v in enumerate(views):
            send[i, :c] = v[b:b + c]
        recv = self.comm.all_gather(send)
        for r, (rb, rc) in enumerate(wins):
            if r == self.rank or rc == 0:
                continue
            for i, v in enumerate(views):
                v[rb:rb + rc] = recv[r][i, :rc]

    def keyswitch_inplace(self, chain_index, ct, c2, evk):
        """ct [2][l][N] += key switch of c2 [l][N] under evk, on the owned limbs
        (src/eval_key_switch.cu:91-178)."""
        ctx = self.ctx
        size_ql, alpha, n = ctx.size_Ql(chain_index), ctx.size_P, ctx.n
        wins = self.windows(chain_index)
        w = wins[self.rank]
        dev = ct.device
        c2_coeff = torch.empty((size_ql, n), dtype=torch.int64, device=dev)
        ctx.lp_modup_begin(chain_index, c2, c2_coeff, w)
        self._exchange([c2_coeff], [(q0, qc) for q0, qc, _, _ in wins])
        t_mod_up = torch.empty((ctx.beta(chain_index), size_ql + alpha, n), dtype=torch.int64, device=dev)
        ctx.lp_modup_finish(chain_index, t_mod_up, c2, c2_coeff, w)
        cx = torch.empty((2, size_ql + alpha, n), dtype=torch.int64, device=dev)
        ctx.lp_key_switch_inner_prod(chain_index, cx, t_mod_up, evk, w)
        ctx.lp_moddown_begin(chain_index, cx, w)
        self._exchange([cx[0], cx[1]], [(size_ql + p0, pc) for _, _, p0, pc in wins])
        ctx.lp_moddown_finish(chain_index, ct, cx, w)

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
        limit = polys[0].shape[0] if size_ql is None else size_ql
        clipped = [(min(q0, limit), max(0, min(q0 + qc, limit) - min(q0, limit))) for q0, qc, _, _ in wins]
        self._exchange(list(polys), clipped)
