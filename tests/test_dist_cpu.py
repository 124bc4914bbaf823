"""N > 1 bookkeeping of bench.py on CPU: world_size-2 gloo run of the barrier / max-over-ranks /
aggregate path and the unit sharding (the data path itself has no collective)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_units_partitions_everything():
    sys.path.insert(0, ROOT)
    import bench
    for total in (64, 7, 1, 0):
        for world in (1, 2, 4, 8):
            got = [bench.shard_units(total, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        rank, world, local_rank, dist = bench.dist_setup("gloo")
        assert world == 2 and dist is not None
        lo, hi = bench.shard_units(64, world, rank)
        dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
        elapsed = bench.max_over_ranks(time.perf_counter() - t0, dist, "cpu")
        value = bench.aggregate_value(hi - lo, 1, world, elapsed)
        if rank == 0:
            print(json.dumps({{"elapsed": elapsed, "value": value, "units": hi - lo}}))
        dist.barrier()
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["units"] == 32
    assert res["elapsed"] >= 0.1          # the max over ranks, not rank 0's own 0.05 s
    assert abs(res["value"] - 64 / res["elapsed"]) < 1e-6


def test_bench_gpus_flag_launches_ranks():
    """`bench.py --gpus 2` with no launcher environment must start two ranks itself (a child torch.distributed.run)
    and relay ONE JSON line whose n_gpus is the number of ranks that took part (gloo rehearsal, no GPU work)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--dry-run", "--backend", "gloo"], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["config"]["rank0_units_of_64"] == 32
    # one rank: no child process, same line shape
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0",
                          "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1


def test_limb_windows_tile_the_basis():
    sys.path.insert(0, ROOT)
    from phantom_fhe_amd.limb_parallel import limb_windows
    for size_ql, alpha in ((36, 12), (35, 12), (6, 2), (5, 3), (1, 1), (3, 1), (20, 4)):
        for world in (1, 2, 3, 4, 8):
            wins = limb_windows(size_ql, alpha, world)
            assert [l for q0, qc, _, _ in wins for l in range(q0, q0 + qc)] == list(range(size_ql))   # contiguous, in rank order
            assert [l for _, _, p0, pc in wins for l in range(p0, p0 + pc)] == list(range(alpha))
            # balanced: every rank carries ceil((l + alpha) / G) limbs or one less, and the Q limbs alone are balanced too
            tot = [qc + pc for _, qc, _, pc in wins]
            assert max(tot) == -(-(size_ql + alpha) // world) and max(tot) - min(tot) <= 1
            assert max(w[1] for w in wins) - min(w[1] for w in wins) <= 1
    assert limb_windows(36, 12, 8)[0] == (0, 5, 0, 1) and limb_windows(36, 12, 8)[7] == (32, 4, 10, 2)


def test_limb_parallel_exchange_world_size_2_gloo(tmp_path):
    """The two all-gathers and the broadcast of the limb-parallel key switch over a real
    torch.distributed group (gloo, CPU tensors; the per-rank compute is covered on the GPU by
    tests/test_gpu_limb_parallel.py)."""
    script = tmp_path / "lp_worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from phantom_fhe_amd.limb_parallel import LimbParallelEvaluator, TorchDistComm
        dist.init_process_group("gloo")
        comm = TorchDistComm()
        assert comm.world == 2 and comm.staged

        class Ctx:  # geometry only: 5 Q limbs, alpha = 3, N = 8
            size_P, n = 3, 8
            def size_Ql(self, chain_index): return 6 - chain_index
        ev = LimbParallelEvaluator(Ctx(), comm)
        wins = ev.windows(1)
        assert wins == [(0, 3, 0, 1), (3, 2, 1, 2)]                   # balanced: 4 limbs each
        full = torch.arange(5 * 8, dtype=torch.int64).reshape(5, 8) + 1
        q0, qc, p0, pc = wins[comm.rank]
        bq, bp = 3, 2                                                  # the largest windows: block sizes of the exchanges
        mine = torch.zeros(2 * bq, 8, dtype=torch.int64)               # rank-blocked, padded: rank r's limbs from row r * bq
        mine[comm.rank * bq:comm.rank * bq + qc] = full[q0:q0 + qc]    # what lp_modup_begin(row_begin=rank*bq) does
        comm.all_gather_blocks(mine, bq)
        q_rows = [0, 1, 2, 3, 4]                                       # rank 1's block starts at row 3 = its first limb here
        assert torch.equal(mine[q_rows], full)
        plimbs = torch.arange(2 * 3 * 8, dtype=torch.int64).reshape(2, 3, 8) + 100   # [poly][special limb]
        pm = torch.zeros(2 * 2 * bp, 8, dtype=torch.int64)             # [rank][poly][bp] rows: ONE gather for both polys
        for poly in range(2):                                          # what lp_moddown_begin(row_begin=rank*2*bp, poly_stride=bp) does
            pm[comm.rank * 2 * bp + poly * bp:comm.rank * 2 * bp + poly * bp + pc] = plimbs[poly, p0:p0 + pc]
        comm.all_gather_blocks(pm, 2 * bp)
        p_rows = [0, 4, 5]                                             # the row table lp_moddown_finish reads poly 0 by
        assert torch.equal(pm[p_rows], plimbs[0]) and torch.equal(pm[[r + bp for r in p_rows]], plimbs[1])
        try:
            comm.all_gather_blocks(pm[:, ::2], 2 * bp)                 # a strided view must be refused, not gathered wrongly
            raise SystemExit("non-contiguous buffer accepted")
        except AssertionError:
            pass
        last = full[4].clone() if comm.rank == 0 else torch.zeros(8, dtype=torch.int64)
        comm.broadcast(last, 0)
        assert torch.equal(last, full[4])
        dist.barrier()
        dist.destroy_process_group()
        print("ok", comm.rank)
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("ok") == 2
