"""The N>1 path of bench.py (leansdr_amd/shard.py): independent captures, no data-path collective.
Two CPU processes over gloo check rank/seed assignment and the whole-job aggregation
(sum of units ÷ max of times)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys, json
        sys.path.insert(0, {ROOT!r})
        from leansdr_amd.shard import Shard
        s = Shard(backend="gloo")
        assert s.world == 2
        s.barrier()
        units = 1000.0 * (s.rank + 1)          # rank 0: 1000, rank 1: 2000
        secs = 0.5 if s.rank == 0 else 2.0     # the slow rank sets the clock
        total, dt, rate = s.aggregate(units, secs)
        s.barrier()
        # one file per rank: two processes printing to one pipe can interleave their lines
        open({str(tmp_path)!r} + "/rank%d.json" % s.rank, "w").write(json.dumps(dict(rank=s.rank, seed=s.capture_seed(), total=total, dt=dt, rate=rate)))
        s.close()
    """))
    import socket
    with socket.socket() as sk:          # a free port: fixed ones collide with a rendezvous still in TIME_WAIT
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=280)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    import json
    rows = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in (0, 1)]
    assert sorted(r["rank"] for r in rows) == [0, 1]
    assert sorted(r["seed"] for r in rows) == [1, 2]          # one capture per rank
    for r in rows:
        assert r["total"] == 3000.0 and r["dt"] == 2.0 and r["rate"] == 1500.0


def test_single_rank_needs_no_torch():
    from leansdr_amd.shard import Shard
    s = Shard()
    assert (s.rank, s.world) == (0, 1)
    assert s.aggregate(10.0, 2.0) == (10.0, 2.0, 5.0)
