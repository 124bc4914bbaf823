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
