"""The per-GPU work of the stereo configuration (two 512 x 512 fisheye streams, one per GPU) on one GPU: the stream's
pipelined extraction, the per-frame barrier through ppg_shutter_release, the descriptor swap with the partner rank
(ppg_stereo_swap; in a world of one rank the partner is the rank itself, so the message makes the full trip through
RCCL's send and receive) and the left-right match on this side (ppg_match_nn on the two descriptor sets).
usage: prof_config4.py [steps]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
streams = importlib.import_module("ppg_slam_amd.streams")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H = W = 512
cfg = pkg.default_config(W, H, max_batch=1, max_num=1024)
for i, k in enumerate((190.978477, 0.0, 254.931706, 0.0, 190.973307, 256.897442, 0.0, 0.0, 1.0)):
    cfg.K[i] = k
cfg.fisheye = 1
for i, d in enumerate((0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736)):
    cfg.D[i] = d
ctx = pkg.Context(cfg)
mctx = pkg.Context(cfg, matcher_only=True)      # the left-right match runs on a context of its own, beside the pipeline
sh = streams.Shutter(ctx, 0, 1, streams.shutter_unique_id())
scene = synth.make_scene(H, W, 11, "dense")
frames = [synth.frame_from_scene(scene, H, W, t) for t in range(8)]
ctx.extract(frames[0])


def run(with_barrier, with_swap, with_match, threaded=False):
    """threaded: swap and left-right match on a second host thread with the matcher context (the calls release the GIL),
    so that the thread driving the pipeline never waits for them."""
    import queue
    import threading
    depth = 4
    acc = {"swap": 0.0, "match": 0.0, "nm": 0}

    def stereo(desc):
        ta = time.perf_counter()
        other = sh.stereo_swap(0, desc)
        acc["swap"] += time.perf_counter() - ta
        if with_match:
            ta = time.perf_counter()
            m, best, second, cnt = mctx.match_nn(desc, other, th=0.7, ratio=0.9, mode=pkg.MATCH_INIT)
            acc["match"] += time.perf_counter() - ta
            acc["nm"] += int(cnt)

    q = queue.Queue(maxsize=4)

    def worker():
        while True:
            d = q.get()
            if d is None:
                return
            stereo(d)

    th = threading.Thread(target=worker) if threaded else None
    if th:
        th.start()
    t0 = time.perf_counter()
    for t in range(steps + depth):
        if t < steps:
            ctx.submit_batch([frames[t % 8]], match_prev=True)
        if t >= depth - 1 and t - (depth - 1) < steps:
            outs, cnts, m12, n = ctx.wait_batch()
            if with_barrier:
                sh.release(t - (depth - 1))
            if with_swap:
                if th:
                    q.put(outs[0]["desc"])
                else:
                    stereo(outs[0]["desc"])
    if th:
        q.put(None)
        th.join()
    if with_barrier:
        sh.drain()
    dt = time.perf_counter() - t0
    return steps / dt, 1e3 * acc["swap"] / steps, 1e3 * acc["match"] / steps, acc["nm"] / steps


def run_resident(overlapped):
    """ppg_stereo_match after every collected frame: swap GPU to GPU and the left-right match on the shutter's own matcher;
    overlapped: ppg_stereo_match_start for this frame, ppg_stereo_match_finish for the one before."""
    depth = 4
    started = False
    t_call, nm = 0.0, 0
    t_sub = t_wait = t_rel = 0.0
    t0 = time.perf_counter()
    for t in range(steps + depth):
        if t < steps:
            ta = time.perf_counter()
            ctx.submit_batch([frames[t % 8]], match_prev=not os.environ.get("PROF_NO_MATCH_PREV"))
            t_sub += time.perf_counter() - ta
        if t >= depth - 1 and t - (depth - 1) < steps:
            ta = time.perf_counter()
            ctx.wait_batch(to_dict=False)
            t_wait += time.perf_counter() - ta
            ta = time.perf_counter()
            sh.release(t - (depth - 1))
            t_rel += time.perf_counter() - ta
            ta = time.perf_counter()
            if overlapped:      # fetch the previous frame's result, start this frame's
                if started:
                    nm += sh.stereo_match_finish()[1]
                sh.stereo_match_start(0, 0, 0.9)
                started = True
            else:
                nm += sh.stereo_match(0, 0, 0.9)[1]
            t_call += time.perf_counter() - ta
    if started:
        nm += sh.stereo_match_finish()[1]
    sh.drain()
    dt = time.perf_counter() - t0
    print("    host per frame: submit %.3f ms, wait %.3f ms, release %.3f ms" % (1e3 * t_sub / steps, 1e3 * t_wait / steps, 1e3 * t_rel / steps))
    return steps / dt, 1e3 * t_call / steps, nm / steps


only = os.environ.get("PROF_ONLY")   # just the last variant
for label, flags in () if only else (("extraction + match against the previous frame", (False, False, False)),
                     ("+ barrier per frame", (True, False, False)),
                     ("+ descriptor swap", (True, True, False)),
                     ("+ left-right match", (True, True, True)),
                     ("the same, swap and match on a second thread", (True, True, True, True))):
    fps, swap_ms, match_ms, nm = run(*flags)
    print("%-48s %7.1f frames/s   swap %.3f ms   left-right match %.3f ms (%d matches)" % (label, fps, swap_ms, match_ms, nm), flush=True)
ctx.extract(frames[0])
if only:
    print(run_resident(True), flush=True)
    sh.close(); mctx.close(); ctx.close()
    sys.exit(0)
ta = time.perf_counter()
for _ in range(100):
    sh.stereo_match(0, 0, 0.9)
print("ppg_stereo_match on an idle GPU: %.3f ms per call; ppg_stereo_swap %.3f ms" % (
    1e1 * (time.perf_counter() - ta), 1e3 * min(__import__("timeit").repeat(lambda: sh.stereo_swap(0, np.zeros((1024, 256), np.float32)), number=1, repeat=20))), flush=True)
for label, ov in (("+ barrier + ppg_stereo_match (resident)", False), ("+ barrier + ppg_stereo_match_start / _finish", True)):
    fps, call_ms, nm = run_resident(ov)
    print("%-48s %7.1f frames/s   stereo calls %.3f ms per frame (%d matches)" % (label, fps, call_ms, nm), flush=True)
sh.close()
mctx.close()
ctx.close()
