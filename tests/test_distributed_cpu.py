"""CPU, world_size 2 over gloo: the N > 1 layout of the hot path -- streams sharded by
``stream_id % world``, no data-path collective, barrier + max-time + stats all-reduce -- with
the pinned tracker oracle standing in for the per-rank GPU work."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, q):
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import rtmodt_amd  # noqa: F401
    pkg = sys.modules["rtmodt_amd"]
    from oracle import tracker_oracle as T
    sync = pkg.streams.NodeSync(world=world, rank=rank, local_rank=rank, backend="gloo", init_method=f"tcp://127.0.0.1:{port}")
    mine = pkg.streams.shard(n_streams, world, rank)
    sync.barrier()
    births = frames = 0
    digests = {}
    for sid in mine:                                   # independent streams: nothing crosses ranks
        xy, cf, cl = pkg.synth.box_sequence(20 + sid, 320, 6, seed=1234 + sid)
        trk = T.TrackerOracle()
        for f in range(6):
            trk.update(xy[f], cf, cl)
            frames += 1
        births += trk.next_id - 1
        digests[sid] = T.state_digest(trk.snapshot()).tolist()
    sync.barrier()
    t = sync.max_time(0.25 * (rank + 1))
    tot = sync.sum_stats([frames, births, len(mine)])
    q.put((rank, mine, t, tot, digests))
    sync.close()


def test_two_ranks_shard_streams_and_reduce(pkg):
    world, n_streams = 2, 7
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(shards[0] + shards[1]) == list(range(n_streams))            # every stream exactly once
    assert all(pkg.streams.owner(s, world) == r for r in range(world) for s in shards[r])
    assert all(abs(r[2] - 0.5) < 1e-9 for r in res)                           # MAX over ranks
    assert res[0][3] == res[1][3] == [n_streams * 6, res[0][3][1], n_streams]  # SUM identical on every rank
    # a stream's result does not depend on which rank ran it
    from oracle import tracker_oracle as T
    xy, cf, cl = pkg.synth.box_sequence(20 + 3, 320, 6, seed=1234 + 3)
    trk = T.TrackerOracle()
    for f in range(6):
        trk.update(xy[f], cf, cl)
    assert res[1][4][3] == T.state_digest(trk.snapshot()).tolist()


def test_shard_properties(pkg):
    for world in (1, 2, 4, 8):
        got = [pkg.streams.shard(64, world, r) for r in range(world)]
        assert sorted(sum(got, [])) == list(range(64))
        assert all(len(g) == 64 // world for g in got)
    assert pkg.streams.shard(64, 8, 3)[:3] == [3, 11, 19]
    s = pkg.streams.NodeSync(world=1, rank=0)
    assert s.max_time(1.5) == 1.5 and s.sum_stats([1, 2]) == [1, 2]
    s.barrier()


def test_bench_starts_its_own_ranks_when_run_bare():
    """the driver runs `python bench.py --gpus N` WITHOUT a launcher.  bench.py must then start the N
    ranks itself (a child `torch.distributed.run`, before anything touches a GPU) and print ONE line that reports N --
    never a 1-GPU measurement labelled N.  `--launch-check` stops after the rendez-vous, so this runs on CPUs (gloo)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_joined"] == 2 and rec["launch_check"] is True
    # under a launcher whose world disagrees with --gpus the run is refused instead of mislabelled
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                         env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_eight_ranks_join_and_share_one_weight_file():
    """what can be proven about the 8-GPU run without the node.  `bench.py --gpus 8` run bare starts 8 ranks (gloo here,
    nobody touches a GPU), all of them join, the single-writer weight file (temporary name + rename by rank 0) is read WHOLE by all 8 after
    the barrier, every rank contributes its own entry to `per_rank_frames_s`, and the one line printed says n_gpus = 8."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--one-device", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["ranks_joined"] == 8
    assert rec["per_rank_frames_s"] == [float(r) for r in range(8)]
    assert rec["ranks_that_read_the_whole_weight_file"] == 8


def _tune_cache_writer(path, rank, q):
    # what engine.hip's autotune_ops does with RTMODT_TUNE_CACHE: whole file to "<path>.tmp.<pid>", then rename()
    body = "#rtmodt-tune tiles=25 table=0\n" + "".join(f"key{k}\t{rank} {rank} {rank}\n" for k in range(2000))
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(body)
    os.replace(tmp, path)
    q.put(len(body))


def test_eight_concurrent_creators_never_leave_half_a_file(tmp_path):
    """Eight processes rewriting ONE tune-cache / weight file at once through temporary name + rename (the engine's and bench.py's protocol):
    a reader sees one writer's complete file, never a mixture, and no temporary is left behind."""
    import multiprocessing as mp
    path = str(tmp_path / "shared.txt")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tune_cache_writer, args=(path, r, q)) for r in range(8)]
    for p in procs:
        p.start()
    sizes = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    txt = open(path).read()
    assert len(txt) in sizes
    owners = {line.split("\t")[1] for line in txt.splitlines()[1:]}
    assert len(owners) == 1, "mixed writers in one file"
    assert [f for f in os.listdir(tmp_path) if ".tmp." in f] == []
