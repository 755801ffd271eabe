"""Long run of the reference's thread pattern: one thread drives the extractor's pipeline (four batches in flight, match
against the previous frame), two more threads call the matcher entry points (projection matcher, SearchForInitialization,
the point-pair-graph matcher, the entry point that allocates nothing) on matcher contexts of their own.  Every result is
compared bit for bit with the single-threaded one.  usage: soak_threads.py [seconds] [frames/s of the tracking thread]"""
import ctypes as C
import importlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
pace = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0    # frames/s of the tracking thread; 0 = as fast as the pipeline goes
W, H = 640, 480
scene = synth.make_scene(H, W, 4242, "dense")
imgs = [synth.frame_from_scene(scene, H, W, t) for t in range(6)]
ext = pkg.Context(pkg.default_config(W, H, max_batch=1, max_num=1024))
f0, f1 = ext.extract(imgs[0]), ext.extract(imgs[1])
rng = np.random.default_rng(5)
g = synth.random_map_graph(rng, f1["desc"], f1["un"], min(600, f1["n"]), 1500)
m = g["desc"].shape[0]
uv = f0["un"] + np.array([3.0, 1.0], np.float32)


def proj(ctx):
    return ctx.match_projected(f1["un"], f1["desc"], f0["desc"], uv, 15.0, 0.8, 0.0, None,
                               np.ones(f0["n"], np.uint8), np.zeros(f1["n"], np.uint8))


def extend(ctx):
    return ctx.extend_map_matches(0.8, 5.0, g, f1["desc"], f1["un"], f1["edge_idx"], f1["conn_off"], f1["conn_idx"],
                                  np.zeros(m, np.uint8), np.full(f1["n"], -1, np.int32), np.full(f1["n_edges"], -1, np.int32))


def init(ctx):
    return ctx.search_for_initialization(0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 50)


def allpairs(ctx):
    return ctx.match_nn(f0["desc"], f1["desc"], th=0.7, ratio=0.9, mode=pkg.MATCH_INIT)


def same(a, b):
    return all((np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y) for x, y in zip(a, b))


want = {fn: fn(ext) for fn in (proj, extend, init, allpairs)}
want_frames = []
ext.extract(imgs[-1])
for im in imgs:
    fr = ext.extract(im)
    want_frames.append((fr, ext.match_prev_batch(1, 0.9)))
errors, counts = [], {"frames": 0, "local mapping": 0, "loop closing": 0}
spent, calls = {}, {}
stop = time.time() + seconds
start = threading.Barrier(3)


def tracking():
    try:
        start.wait()
        ext.submit_batch([imgs[-1]], match_prev=False)
        ext.wait_batch()
        k = done = 0
        t_first = time.time()
        while time.time() < stop or done < k:
            if time.time() < stop:
                if pace > 0:       # a camera's frame rate instead of flat out
                    time.sleep(max(0.0, t_first + k / pace - time.time()))
                ext.submit_batch([imgs[k % 6]], match_prev=True)
                k += 1
            if k - done >= 4 or time.time() >= stop:
                frames, cnts, m12, n = ext.wait_batch()
                fr, (wc, wm) = want_frames[done % 6]
                a = frames[0]
                if not (np.array_equal(a["xy"], fr["xy"]) and np.array_equal(a["edge_idx"], fr["edge_idx"]) and np.array_equal(a["desc"], fr["desc"])
                        and cnts[0] == wc[0] and np.array_equal(m12[0], wm[0])):
                    errors.append("tracking: pipelined frame %d differs from the blocking calls" % done)
                done += 1
                counts["frames"] = done
    except Exception as e:  # noqa: BLE001
        errors.append("tracking: %r" % (e,))


def mapping(name, fns):
    def run():
        ctx = None
        try:
            ctx = pkg.Context(pkg.default_config(W, H, max_num=1024), matcher_only=True)
            start.wait()
            while time.time() < stop:
                for fn in fns:
                    ta = time.perf_counter()
                    res = fn(ctx)
                    key = name + " " + fn.__name__
                    spent[key] = spent.get(key, 0.0) + time.perf_counter() - ta
                    calls[key] = calls.get(key, 0) + 1
                    if not same(res, want[fn]):
                        errors.append(name + ": a concurrent " + fn.__name__ + " differs from the single-threaded one")
                    counts[name] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(name + ": %r" % (e,))
        finally:
            if ctx is not None:
                ctx.close()
    return run


ts = [threading.Thread(target=tracking), threading.Thread(target=mapping("local mapping", [proj, init, allpairs])),
      threading.Thread(target=mapping("loop closing", [extend, proj, allpairs]))]
t0 = time.time()
for t in ts:
    t.start()
while any(t.is_alive() for t in ts):
    time.sleep(20)
    print("%.0f s: %s, %d errors" % (time.time() - t0, counts, len(errors)), flush=True)
    if time.time() > stop + 60:
        print("a thread is stuck", flush=True)
        os._exit(2)
ext.close()
print("%.0f s, three threads: %d pipelined frames, %d + %d matcher calls, %d differ from the single-threaded results"
      % (time.time() - t0, counts["frames"], counts["local mapping"], counts["loop closing"], len(errors)))
print("mean call time beside the other threads: " + ", ".join("%s %.2f ms" % (k, 1e3 * spent[k] / calls[k]) for k in sorted(spent)))
for e in errors[:5]:
    print(e)
sys.exit(1 if errors else 0)
