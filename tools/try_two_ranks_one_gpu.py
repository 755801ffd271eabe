"""Rehearsal: two ranks of the C-ABI shutter (ppg_shutter_*, ppg_stereo_swap) on ONE GPU, each in a process of its own.
RCCL may refuse two ranks on one device; then this prints its message and exits 3.  usage: try_two_ranks_one_gpu.py"""
import importlib
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, conn, q):
    try:
        import numpy as np
        from __graft_entry__ import load_pkg
        pkg = load_pkg()
        streams = importlib.import_module("ppg_slam_amd.streams")
        ctx = pkg.Context(pkg.default_config(128, 96, max_num=64), matcher_only=True)
        if rank == 0:
            uid = streams.shutter_unique_id()
            conn.send(uid)
        else:
            uid = conn.recv()
        sh = streams.Shutter(ctx, rank, world, uid)
        seen = []
        for k in range(12):
            sh.release(k if rank == 0 else max(0, k - 2))   # rank 1 lags two groups behind
            seen.append(sh.released)
        last = sh.drain()
        rng = np.random.default_rng(10 + rank)
        rows = (40, 7, 0, 64)
        other = (13, 64, 5, 0)
        got = []
        for f in range(4):
            n = rows[f] if rank == 0 else other[f]
            d = rng.normal(size=(n, 256)).astype(np.float32)
            g = sh.stereo_swap(rank ^ 1, d)
            got.append((g.shape[0], float(g.sum()) if g.size else 0.0, float(d.sum()) if d.size else 0.0))
        sh.close()
        ctx.close()
        q.put((rank, "ok", last, seen, got))
    except Exception as e:   # noqa: BLE001
        q.put((rank, "error", str(e)[:400], None, None))


if __name__ == "__main__":
    ctxm = mp.get_context("spawn")
    a, b = ctxm.Pipe()
    q = ctxm.Queue()
    ps = [ctxm.Process(target=worker, args=(r, 2, (a, b)[r], q)) for r in range(2)]
    for p in ps:
        p.start()
    res = []
    try:
        for _ in range(2):
            res.append(q.get(timeout=90))
    except Exception:
        print("timeout: no result within 90 s", res)
        for p in ps:
            p.kill()
        sys.exit(4)
    for p in ps:
        p.join(timeout=30)
    res.sort()
    for r in res:
        print(r)
    if any(r[1] != "ok" for r in res):
        sys.exit(3)
    (r0, _, last0, seen0, got0), (r1, _, last1, seen1, got1) = res
    assert last0 == last1 == 9, (last0, last1)              # min(11, 9)
    for f in range(4):
        assert got0[f][0] == (13, 64, 5, 0)[f] and got1[f][0] == (40, 7, 0, 64)[f]
        assert abs(got0[f][1] - got1[f][2]) < 1e-3 and abs(got1[f][1] - got0[f][2]) < 1e-3
    print("two ranks on one GPU: barrier and stereo swap ok")
