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


def test_bench_self_launches_ranks():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (torch.distributed.run on 127.0.0.1, gloo) and
    rank 0 prints ONE line for the whole job; --dry-run exercises exactly that plumbing without a GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=280)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["units"] == 3000.0 and j["seconds"] == 2.0


def test_bench_c1_workload_self_launches_ranks():
    """BASELINE config 4's launch path: `python bench.py --workload c1 --gpus 2` (one rank per GPU, every rank its own captures,
    gloo for the barrier and the totals) — plumbing only (--dry-run), on CPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--gpus", "2", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=280)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["workload"] == "c1" and j["n_gpus"] == 2 and j["ranks"] == 2 and j["units"] == 3000.0
    # every rank gives its captures their own noise seeds (gathered from the ranks, computed by the function the real run calls)
    seeds = j["capture_seeds_by_rank"]
    assert len(seeds) == 2 and all(len(s) == 32 for s in seeds)      # --c1-captures defaults to 32 per GPU
    flat = [v for s in seeds for v in s]
    assert len(set(flat)) == len(flat) == 64 and seeds[0][0] == 1000 and seeds[1][0] == 2000


def test_c1_job_seeds_come_from_capture_seeds():
    """C1Job(seed0=capture_seeds(rank, 1)[0]) numbers its captures seed0 + k = capture_seeds(rank, n)[k]."""
    sys.path.insert(0, ROOT)
    import bench_c1
    for rank in (0, 1, 7):
        s = bench_c1.capture_seeds(rank, 16)
        assert s == [s[0] + k for k in range(16)] and s[0] == 1000 * (rank + 1)
    assert not set(bench_c1.capture_seeds(0, 16)) & set(bench_c1.capture_seeds(1, 16))


def test_bench_rejects_mismatched_launcher():
    """--gpus must equal the number of ranks the launcher started (n_gpus stays honest)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode != 0 and b"--gpus 4" in p.stderr


def test_no_rccl_in_the_control_plane():
    src = open(os.path.join(ROOT, "leansdr_amd", "shard.py")).read()
    assert 'init_process_group("gloo")' in src and 'init_process_group("nccl"' not in src


def test_single_rank_needs_no_torch():
    from leansdr_amd.shard import Shard
    s = Shard()
    assert (s.rank, s.world) == (0, 1)
    assert s.aggregate(10.0, 2.0) == (10.0, 2.0, 5.0)


def test_bench_default_geometry():
    """bench.py's C2 defaults: one capture per GPU with 256 Mi-sample batches (north_star's sharding); whatever the number of
    captures, a GPU's batch is 256 Mi samples and its step 24 Gi."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    for caps, want in ((1, (256, 96)), (2, (128, 96)), (4, (64, 96)), (8, (32, 96))):
        a = bench.resolve_defaults(argparse.Namespace(captures=caps, batch_msamples=None, batches_per_step=None))
        assert (a.batch_msamples, a.batches_per_step) == want and a.batch_msamples * caps * a.batches_per_step == 24 * 1024
    a = bench.resolve_defaults(argparse.Namespace(captures=2, batch_msamples=16, batches_per_step=None))
    assert (a.batch_msamples, a.batches_per_step) == (16, 768)
    a = bench.resolve_defaults(argparse.Namespace(captures=1, batch_msamples=256, batches_per_step=6))
    assert a.batches_per_step == 6 and a.more_batch_msamples == 64


def test_bench_line_stays_small():
    """The driver parses the LAST stdout line and keeps an 8 KB tail: round 4's 28.7 KB line came back `parsed: null`.  The
    line bench.py prints is compact_line() of the full record — under 4 KB whatever the full record holds, with the contract's
    keys, `roofline` and `cpu_baseline` in it; the full record goes to bench_full.json."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    for rec in ("profiles/r04_bench/bench.json", "profiles/r04_bench/c1.json"):
        full = json.load(open(os.path.join(ROOT, rec)))
        full["more"] = {f"extra{i}": {"value": 1.0, "note": "x" * 4000} for i in range(40)}       # however much detail there is
        full.setdefault("verified", {})["per_capture"] = [{"blob": "y" * 3000}] * 16
        line = bench.compact_line(full)
        assert len(line.encode()) <= bench.LINE_LIMIT < 8192 and "\n" not in line
        got = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in got, k
        assert "workload" in got["config"] and "more" not in got
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in got["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in got["cpu_baseline"], k
    # a summary that outgrew the line is dropped before the contract's keys are
    full["summary"] = {f"row{i}": [1, 2, True] for i in range(2000)}
    got = json.loads(bench.compact_line(full))
    assert "summary" not in got and "roofline" in got and "cpu_baseline" in got


def test_recorded_traffic_is_keyed_by_kernel():
    """roofline.traffic is a RECORDED figure (PMC passes under profiles/): it must come from the same kernel at the same launch
    size, never from another kernel that happened to run the same batch."""
    sys.path.insert(0, ROOT)
    import bench
    t, src = bench.pmc_traffic("k_fir_mfma_stream", 268369920)
    assert t and "pmc_traffic.json" in src
    assert bench.pmc_traffic("k_fir_persist", 268369920) == (None, None) or "k_fir_persist" in open(os.path.join(ROOT, bench.pmc_traffic("k_fir_persist", 268369920)[1])).read()
    assert bench.pmc_traffic("k_fir_mfma_stream", 12345) == (None, None)


def test_numa_pinning_degrades_gracefully():
    """pin_to_gpu_numa: None (and no exception, affinity untouched) where the PCI device or its NUMA node cannot be told."""
    from leansdr_amd.shard import Shard
    before = os.sched_getaffinity(0)
    assert Shard().pin_to_gpu_numa("ffff:ff:1f.7") is None
    assert os.sched_getaffinity(0) == before


def test_arena_window_is_a_buffer_view():
    """capi.ArenaWindow: what lsdr_arena_place (Arena.place) hands a pipeline for a window of the arena — the pointer, .at() as a DevBuf's;
    free() gives the window back to its arena (here a closed one: nothing to call)."""
    import ctypes
    sys.path.insert(0, ROOT)
    import leansdr_amd.capi as capi

    class _Arena:
        h = None
    w = capi.ArenaWindow(_Arena, 0x7000_0020_0000, 4 << 20, probe_ms=0.36)
    assert w.ptr == 0x7000_0020_0000 and isinstance(w.at(0), ctypes.c_void_p) and w.nbytes == 4 << 20 and w.probe_ms == 0.36
    assert w.at(4096).value == 0x7000_0020_0000 + 4096
    w.free()
    assert w.ptr is None
