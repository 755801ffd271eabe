"""Long run of the pipelined entry points (ppg_submit_batch / ppg_wait_batch, four batches in flight, match against the
previous batch) at 32 streams per step: every collected batch - key points, scores, descriptors, edges, adjacency, collinear
pairs, match vectors - is compared bit for bit (CRC) with what the blocking calls return for the same images.
usage: soak_pipeline.py [steps] [precision ...]"""
import importlib
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
H, W, B, T = 480, 640, 32, 8
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
modes = [int(a) for a in sys.argv[2:]] or [0, 1]
scenes = [synth.make_scene(H, W, 1000 * s + 11, "dense" if s % 4 else "sparse") for s in range(B)]
frames = [[synth.frame_from_scene(scenes[s], H, W, t) for s in range(B)] for t in range(T)]
KEYS = ("xy", "score", "un", "out", "desc", "edge_idx", "edge_score", "conn_off", "conn_idx", "coline_off", "coline_pairs")


def digest(outs, cnts, m12):
    d = 0
    for o in outs:
        for k in KEYS:
            d = zlib.crc32(np.ascontiguousarray(o[k]).tobytes(), d)
    if cnts is not None:
        d = zlib.crc32(np.ascontiguousarray(cnts).tobytes(), d)
        d = zlib.crc32(np.ascontiguousarray(m12).tobytes(), d)
    return d


for prec in modes:
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=1024, precision=prec))
    # blocking reference: frame group t after frame group t-1 (mod T), matched against it
    want = {}
    ctx.extract_batch(frames[T - 1])
    for t in range(T):
        outs = ctx.extract_batch(frames[t])
        cnts, m12 = ctx.match_prev_batch(B, 0.9)
        want[t] = digest(outs, cnts, m12)
    bad, depth, done = 0, 4, 0
    t0 = time.perf_counter()
    ctx.submit_batch(frames[T - 1], match_prev=False)
    ctx.wait_batch()
    for k in range(N + depth - 1):
        if k < N:
            ctx.submit_batch(frames[k % T], match_prev=True, nnratio=0.9)
        if k >= depth - 1:
            outs, cnts, m12, n = ctx.wait_batch()
            if digest(outs, cnts, m12) != want[done % T]:
                bad += 1
                print("precision", prec, "step", done, "differs", flush=True)
            done += 1
            if done % 200 == 0:
                print("precision", prec, done, "steps, %.0f s" % (time.perf_counter() - t0), flush=True)
    ctx.close()
    print("precision %d: %d pipelined steps of %d frames, %d differ from the blocking calls" % (prec, done, B, bad), flush=True)
    if bad:
        sys.exit(1)
print("soak ok")
