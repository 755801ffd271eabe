"""gloo tests (world sizes 2 and 4) of the multi-GPU path: stream sharding, the global-shutter barrier, the step loop."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_pkg()
    streams = importlib.import_module("ppg_slam_amd.streams")
    mine = streams.shard_streams(8, world, rank)
    bar = streams.ShutterBarrier(dist, "cpu")
    released = []
    for k in range(5):
        done = k if rank == 0 else max(0, k - 1)  # rank 1 lags one frame group behind
        bar.release(done)
        released.append(bar.drain())              # release() itself does not wait for the collective
    # every rank sees the same release sequence = the slowest rank's progress
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, released))
    q.put((rank, gathered))
    dist.destroy_process_group()


def test_sharding_and_barrier_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, gathered in res:
        (s0, r0), (s1, r1) = gathered
        assert sorted(s0 + s1) == list(range(8)) and not set(s0) & set(s1)
        assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]
        assert r0 == r1 == [0, 0, 1, 2, 3]


def _stereo_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_pkg()
    streams = importlib.import_module("ppg_slam_amd.streams")
    ex = streams.StereoExchange(dist, "cpu", max_num=64)
    got = []
    for frame in range(3):  # ragged counts, the second call smaller than the first
        n = (40, 7, 64)[frame] if rank == 0 else (0, 33, 5)[frame]
        g = torch.Generator().manual_seed(100 * rank + frame)
        mine = torch.rand(64, 256, generator=g)
        theirs, m = ex.swap(mine, n)
        got.append((m, theirs.clone()))
    q.put((rank, [(m, t.numpy()) for m, t in got]))
    dist.destroy_process_group()


def test_stereo_exchange_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_stereo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts = {0: (40, 7, 64), 1: (0, 33, 5)}
    for rank in (0, 1):
        other = rank ^ 1
        for frame, (m, t) in enumerate(res[rank]):
            assert m == counts[other][frame]
            g = torch.Generator().manual_seed(100 * other + frame)
            want = torch.rand(64, 256, generator=g)[:m].numpy()
            assert t.shape == (m, 256) and (t == want).all()


def test_shard_streams_covers_everything():
    load_pkg()
    streams = importlib.import_module("ppg_slam_amd.streams")
    for world in (1, 2, 4, 8):
        got = sorted(s for r in range(world) for s in streams.shard_streams(19, world, r))
        assert got == list(range(19))
    with pytest.raises(ValueError):
        streams.shard_streams(4, 2, 2)


def _loop_worker(rank, world, port, q):
    """bench.py's step loop (streams.run_steps) with a stub in place of the GPU context: rank 1 is slower."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_pkg()
    streams = importlib.import_module("ppg_slam_amd.streams")
    bar = streams.ShutterBarrier(dist, "cpu", depth=3)
    log, seen = [], []

    def submit(k):
        log.append(("s", k))

    def wait(k):
        time.sleep(0.002 * (1 + 2 * rank))
        log.append(("w", k))
        seen.append(bar.released)   # what was known to be released before this group's barrier started

    total = 0
    for first, count in ((0, 9), (9, 1), (10, 6)):
        total += streams.run_steps(submit, wait, bar, first, count, depth=4)
        assert bar.drain() == first + count - 1        # every rank has finished every group of the pass
    q.put((rank, total, bar.count, log, seen))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_step_loop_barrier_counts(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, count, log, seen in res:
        assert total == 16 and count == 16                      # one barrier per collected group on every rank
        assert [k for t, k in log if t == "s"] == list(range(16)) and [k for t, k in log if t == "w"] == list(range(16))
        # at most four groups are in flight: group k is collected before group k + 4 is submitted
        pos = {e: i for i, e in enumerate(log)}
        assert all(pos[("w", k)] < pos[("s", k + 4)] for k in range(12) if k + 4 < 9 or k >= 10)
        assert seen == sorted(seen) and all(v < k for k, v in enumerate(seen))   # never ahead of the slowest rank
