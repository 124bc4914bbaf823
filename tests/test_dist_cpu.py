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


def test_limb_windows_tile_the_basis():
    sys.path.insert(0, ROOT)
    from phantom_fhe_amd.limb_parallel import limb_window
    for size_ql, alpha in ((36, 12), (35, 12), (6, 2), (5, 3), (1, 1), (3, 1)):
        for world in (1, 2, 3, 4, 8):
            wins = [limb_window(size_ql, alpha, world, r) for r in range(world)]
            assert sorted(l for q0, qc, _, _ in wins for l in range(q0, q0 + qc)) == list(range(size_ql))
            assert sorted(l for _, _, p0, pc in wins for l in range(p0, p0 + pc)) == list(range(alpha))
            load = [qc + pc for _, qc, _, pc in wins]
            assert max(load) - min(load) <= 1 or (size_ql % world == 0 or alpha % world == 0)


def test_limb_parallel_exchange_world_size_2_gloo(tmp_path):
    """The two all-gathers and the broadcast of the limb-parallel key switch over a real
    torch.distributed group (gloo, CPU tensors; the per-rank compute is covered on the GPU by
    tests/test_gpu_limb_parallel.py)."""
    script = tmp_path / "lp_worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from phantom_fhe_amd.limb_parallel import LimbParallelEvaluator, TorchDistComm, limb_window
        dist.init_process_group("gloo")
        comm = TorchDistComm()
        assert comm.world == 2 and comm.staged

        class Ctx:  # geometry only: 5 Q limbs, alpha = 3, N = 8
            size_P, n = 3, 8
            def size_Ql(self, chain_index): return 6 - chain_index
        ev = LimbParallelEvaluator(Ctx(), comm)
        wins = ev.windows(1)
        full = torch.arange(2 * 8 * 8, dtype=torch.int64).reshape(2, 8, 8) + 1   # [poly][limb][N]
        mine = torch.zeros_like(full)
        q0, qc, p0, pc = wins[comm.rank]
        mine[:, q0:q0 + qc] = full[:, q0:q0 + qc]
        mine[:, 5 + p0:5 + p0 + pc] = full[:, 5 + p0:5 + p0 + pc]
        ev._exchange([mine[0], mine[1]], [(a, b) for a, b, _, _ in wins])            # ragged Q windows (3 + 2)
        assert torch.equal(mine[:, :5], full[:, :5]) and (mine[:, 5:] == 0).sum() > 0
        ev._exchange([mine[0], mine[1]], [(5 + c, d) for _, _, c, d in wins])        # ragged P windows (1 + 2)
        assert torch.equal(mine, full)
        last = full[0, 4].clone() if comm.rank == 0 else torch.zeros(8, dtype=torch.int64)
        comm.broadcast(last, 0)
        assert torch.equal(last, full[0, 4])
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
