"""CPU-side drop-in checks: the host pipe compiles against the REFERENCE's headers when
/root/reference is mounted (dev container only), and the sharded bench path works under gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree not mounted")
def test_pipe_compiles_in_reference_tree():
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-DSAYURI_IN_TREE", "-I/root/reference/src",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "sayuri_amd", "csrc", "host"),
           os.path.join(ROOT, "sayuri_amd", "csrc", "host", "hip_forward_pipe.cc")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_sharded_stats_gather_gloo_world2(tmp_path):
    """The only multi-GPU exchange of the path: barrier + max-over-ranks time + stats gather
    (bench.py / DESIGN.md section 6), exercised with 2 gloo ranks on CPU."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from sayuri_amd.shard import shard_range, gather_stats\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "lo, hi = shard_range(1000, r, w)\n"
        "stats = gather_stats({'games_done': hi - lo, 'nn_queries': 10 * (r + 1), 'elapsed': 1.0 + r})\n"
        "if r == 0:\n"
        "    assert stats['games_done'] == 1000, stats\n"
        "    assert stats['nn_queries'] == 30, stats\n"
        "    assert stats['elapsed_max'] == 2.0, stats\n"
        "    assert stats['per_rank'][1]['games_done'] == 500\n"
        "    print('OK')\n"
        "dist.barrier()\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_selfplay_sharded_by_rank_gloo_world2(tmp_path):
    """Self-play shards by games: every rank runs its own engine (here on the dummy backend) and writes its own
    chunk directory; the only exchange is the stats gather.  Two gloo ranks on CPU."""
    out = tmp_path / "data"
    out.mkdir()
    script = tmp_path / "sp.py"
    script.write_text(
        "import os, sys, glob\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from sayuri_amd import search as S\n"
        "from sayuri_amd.shard import shard_range, gather_stats\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "lo, hi = shard_range(6, r, w)\n"
        f"opts = dict(playouts=80, parallel_games=hi - lo, num_games=hi - lo, seed=100 + r, selfplay_query=['bkp:9:7:1'], target_directory={str(out)!r})\n"
        "st = S.selfplay(None, opts, name_suffix=f'-r{r}')\n"
        "tot = gather_stats({'games_done': st['games_done'], 'nn_queries': st['nn_queries'], 'moves': st['moves'], 'records': st['records'], 'elapsed': st['elapsed']})\n"
        "dist.barrier()\n"
        "if r == 0:\n"
        "    assert tot['games_done'] == 6 and tot['moves'] > 100 and tot['records'] > 100, tot\n"
        f"    dirs = sorted(os.path.basename(d) for d in glob.glob({str(out)!r} + '/tdata/*'))\n"
        "    assert len(dirs) == 2 and dirs[0].endswith('-r0') or dirs[0].endswith('-r1'), dirs\n"
        f"    assert len(glob.glob({str(out)!r} + '/tdata/*/game_*.txt.gz')) == 6\n"
        "    print('OK')\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29534", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_newer_weights_on_one_rank_halt_every_rank_gloo_world2(tmp_path):
    """The periodic exchange (sayuri_amd.shard.PeriodicGather, every 0.25 s here): each rank contributes its counters and
    its own halt wish; rank 1 alone sees a newer network appear, both ranks wind down (reference ShouldHalt semantics,
    src/selfplay/engine.cc:88-90, made collective because the games are sharded over processes)."""
    script = tmp_path / "halt.py"
    script.write_text(
        "import os, sys, threading, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from sayuri_amd import search as S\n"
        "from sayuri_amd.shard import PeriodicGather\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        f"wdir = os.path.join({str(tmp_path)!r}, f'weights{{r}}')\n"
        "os.makedirs(wdir)\n"
        "cur = os.path.join(wdir, 'net-0001.bin'); open(cur, 'w').write('old')\n"
        "if r == 1:\n"
        "    threading.Thread(target=lambda: (time.sleep(1.0), open(os.path.join(wdir, 'net-0002.bin'), 'w').write('new'))).start()\n"
        "pg = PeriodicGather()\n"
        "opts = dict(playouts=30, parallel_games=4, num_games=100000, seed=20 + r, selfplay_query=['bkp:7:7:1'], weights_dir=wdir, weights_file=cur)\n"
        "def hook(st, halt):\n"
        "    return pg.tick(dict(games_done=st['games_done'], nn_queries=st['nn_queries'], moves=st['moves'], playouts=st['playouts'], elapsed=st['elapsed']), halt=halt)\n"
        "st = S.selfplay(None, opts, on_stats=hook, stats_interval=0.25)\n"
        "tot = pg.drain(dict(games_done=st['games_done'], moves=st['moves'], playouts=st['playouts'], elapsed=st['elapsed']))\n"
        "assert st['games_done'] == st['max_games'] < 100000, (r, st)\n"
        "assert pg.any_halt and pg.rounds >= 2, (r, pg.rounds, pg.any_halt)\n"
        "if r == 0:\n"
        "    assert tot['games_done'] >= 25 and len(tot['per_rank']) == 2, tot\n"
        "    print('OK', pg.rounds, tot['games_done'])\n"
        "dist.barrier()\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29535", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_eight_ranks_on_the_fake_device_halt_together(tmp_path):
    """Eight gloo ranks, each with its own queue / games (fibers) / cache over the serial fake device (tools/fake8.py, the
    readiness run of profiles/r03_fake8_*.json in small): every rank takes part in every exchange round, one rank sees a
    newer network and ALL of them wind down (reference Engine::ShouldHalt + pipe.cc:246-258, made collective)."""
    out = tmp_path / "fake8.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fake8.py"), "--ranks", "8", "--games", "16", "--playouts", "16",
                        "--board", "7", "--seconds", "40", "--serial-us", "500", "--halt-rank", "5", "--halt-after", "4", "--out", str(out)],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(out.read_text())
    assert d["ranks_reporting"] == 8 and d["halt_seen_by"] == list(range(8)), d
    assert d["stopped_early_by_halt"] == list(range(8)), d
    assert len(set(d["exchange_rounds"])) == 1 and d["exchange_rounds"][0] >= 2, d["exchange_rounds"]
    assert all(x["nn_evals"] > 0 and x["mean_batch"] > 1 for x in d["per_rank"])


def test_a_run_does_not_halt_over_its_own_network_spelled_differently(tmp_path):
    """weights_dir with a trailing slash, weights_file without one (tools/fake8.py passes them so): ShouldHalt compares the
    newest file of the directory with the loaded one by identity (st_dev, st_ino), so the run plays until its clock ends
    (the reference returns the explicit weights_file from SelectWeights and never halts, engine.cc:63-90)."""
    out = tmp_path / "fake1.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fake8.py"), "--ranks", "1", "--games", "8", "--playouts", "8",
                        "--board", "7", "--seconds", "6", "--serial-us", "300", "--out", str(out)],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(out.read_text())
    assert d["ranks_reporting"] == 1 and d["stopped_early_by_halt"] == [] and d["halt_seen_by"] == [], d
    assert d["per_rank"][0]["games_done"] > 8, d["per_rank"][0]  # more than one game per worker: worker 0 checked at least once


def _bench_on_the_fake_device(tmp_path, world, port, extra=(), launcher=True, devices=None, seconds=10, check=True):
    """launcher=True: the driver's command line for N > 1 (torch.distributed.run around bench.py); launcher=False: the PLAIN
    command `python bench.py --gpus N ...`, which has to start its N ranks itself."""
    fake = str(tmp_path / "libfake_hip.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", os.path.join(ROOT, "tests", "fake_hip", "fake_hip.c"), "-o", fake, "-lpthread"])
    env = dict(os.environ, SAYURI_FAKE_HIP_LIB=fake, FAKE_HIP_DEVICES=str(world if devices is None else devices),
               FAKE_HIP_SERIAL_US="3800", FAKE_HIP_CHEAP="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", str(world), "--steps", "5", "--warmup", "2", "--dist-backend", "gloo", "--selfplay-seconds", str(seconds),
            "--selfplay-games", "64", "--selfplay-visits", "16"] + list(extra)
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--master-port", str(port)] + args
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    if not check:
        return r
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_on_the_fake_device(tmp_path):
    """bench.py's own multi-rank path (the command line the driver uses for N > 1: torch.distributed.run, one rank per
    device, barrier + max-over-ranks timing, the stats all-gather, the periodic exchange of the self-play segment), executed
    on two gloo ranks over the serial fake device -- so that the first run on an 8-GPU node is not this code's first run."""
    d = _bench_on_the_fake_device(tmp_path, 2, 29541, extra=["--no-config5"])  # (configs[4] on two ranks: the plain-command test below)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"] and d["config"]["parallelism"] == "dp2"
    # value = the units ALL ranks processed / the max-over-ranks time: two serial devices at 3.8 ms per 256-batch
    assert 0.5 * 2 * 256 / 3.8e-3 < d["value"] < 1.05 * 2 * 256 / 3.8e-3, d["value"]
    sp = d["selfplay"]
    assert sp["exchange_rounds"] > 0 and sp["nn_evals_per_sec"] > 0 and sp["moves_per_sec"] > 0 and not sp["halt_seen"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_plain_bench_command_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher around it (the shape of the driver's N = 1 command with another N): the
    script starts the two ranks itself and the line says n_gpus 2, with both ranks in every exchange round.  Rounds 1-5 parsed
    --gpus and never read it: such a command ran one rank and printed n_gpus 1."""
    d = _bench_on_the_fake_device(tmp_path, 2, 29551, launcher=False)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 512
    assert 0.5 * 2 * 256 / 3.8e-3 < d["value"] < 1.05 * 2 * 256 / 3.8e-3, d["value"]
    sp = d["selfplay"]
    assert sp["exchange_rounds"] > 0 and sp["exchange"]["world"] == 2 and sp["exchange"]["rounds"] == sp["exchange_rounds"]
    assert sp["nn_evals_per_sec"] > 0 and sp["games_per_hour_empty_board_27min"] is not None
    assert d["tower"] == "per-layer fallback" and "hipcc" in d  # the fake device has no persistent kernel, and the line says so
    # configs[4] runs on every rank of a multi-rank launch (it is an 8-GPU configuration): the sum over the ranks on the line
    c5 = d["config5"]
    assert c5["n_gpus"] == 2 and len(c5["per_rank_evals_per_sec"]) == 2 and c5["evals_per_sec"] > max(c5["per_rank_evals_per_sec"])


def test_plain_bench_command_with_eight_ranks(tmp_path):
    """The same for the configuration the scaling run uses: `python bench.py --gpus 8` on eight fake devices."""
    d = _bench_on_the_fake_device(tmp_path, 8, 29553, launcher=False, seconds=6, extra=["--no-pump", "--no-config5"])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 8 * 256
    assert d["selfplay"]["exchange"]["world"] == 8 and d["selfplay"]["exchange_rounds"] > 0
    assert "cpu_baseline" not in d and "config5" not in d


def test_bench_refuses_more_ranks_than_devices(tmp_path):
    """--gpus 4 on a box with two devices fails loudly before anything is started; and a launcher whose world size differs
    from --gpus is refused by the ranks (the line's n_gpus would not be what was asked for)."""
    r = _bench_on_the_fake_device(tmp_path, 4, 29555, launcher=False, devices=2, check=False)
    assert r.returncode != 0 and "--gpus 4 but this box has 2" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    fake = str(tmp_path / "libfake_hip.so")
    env = dict(os.environ, SAYURI_FAKE_HIP_LIB=fake, FAKE_HIP_DEVICES="2", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29557")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr, r.stderr[-2000:]


def test_exchange_round_that_misses_its_deadline_gloo_world2(tmp_path):
    """PeriodicGather never blocks its caller for longer than its deadline: a round whose peer is late stays in flight, the
    caller goes on with the last completed totals, the next tick collects it first (no new round beside a pending one, so the
    two ranks' sequences of collectives stay identical), and drain() ends with every round landed on both ranks.  Rank 1 is
    0.6 s late to every tick, the deadline is 0.15 s: every round of rank 0 is a late one."""
    script = tmp_path / "late.py"
    script.write_text(
        "import os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from sayuri_amd.shard import PeriodicGather\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "pg = PeriodicGather(timeout=0.15)\n"
        "blocked = []\n"
        "for k in range(5):\n"
        "    if r == 1: time.sleep(0.6)\n"
        "    t0 = time.perf_counter()\n"
        "    pg.tick({'games_done': k + 1, 'nn_queries': 100 * (k + 1) * (r + 1), 'elapsed': float(k)}, halt=(r == 1 and k == 3))\n"
        "    blocked.append(time.perf_counter() - t0)\n"
        "    if r == 0: time.sleep(0.6)\n"
        "tot = pg.drain({'games_done': 6, 'nn_queries': 600 * (r + 1), 'elapsed': 9.0})\n"
        "assert pg._inflight is None and pg.all_done\n"
        "assert tot['games_done'] == 12 and tot['nn_queries'] == 1800 and tot['done'] == 2, tot\n"
        "assert pg.any_halt, 'rank 1 raised its wish in round 3: both ranks must have seen it'\n"
        "if r == 0:\n"
        "    assert max(blocked) < 0.5, blocked            # never the peer's 0.6 s\n"
        "    assert pg.late_rounds >= 3, (pg.late_rounds, pg.latencies_ms)\n"
        "    assert max(pg.latencies_ms) > 300.0           # issue -> landed of a late round is the peer's delay\n"
        "s = pg.latency_summary()\n"
        "assert s['rounds'] == pg.rounds == len(pg.latencies_ms) and s['backend'] == 'gloo' and s['world'] == 2\n"
        "rounds = torch.tensor([pg.rounds]); both = [torch.zeros_like(rounds) for _ in range(w)]\n"
        "dist.all_gather(both, rounds)\n"
        "assert int(both[0]) == int(both[1]), both          # the same number of collectives on both ranks\n"
        "if r == 0: print('OK', pg.rounds, pg.late_rounds, pg.skipped_ticks)\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
