"""The windowed search of Matcher::SearchByProjection(Frame&, const Frame&, th) (matching/src/Matcher.cpp:31-87) on two
dense VGA frames: every key point of the last frame, projected into the current one, searches a 15-pixel window; accepted
rows claim their slot.  Repeated: under `rocprofv3 --kernel-trace --stats` the statistics show how the blocking call
divides into kernels and copies."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
H, W = 480, 640
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
radius = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ctx = pkg.Context(pkg.default_config(W, H, max_batch=2, max_num=1024))
scene = synth.make_scene(H, W, 11, "dense")
f0, f1 = ctx.extract_batch([synth.frame_from_scene(scene, H, W, 0), synth.frame_from_scene(scene, H, W, 1)])
uv = f0["un"] + np.array([3.0, 1.0], np.float32)   # the synthetic camera motion between the frames
n1, n2 = f0["n"], f1["n"]
call = lambda: ctx.match_projected(f1["un"], f1["desc"], f0["desc"], uv, radius, 0.8, 0.0, None, np.ones(n1, np.uint8),
                                   np.zeros(n2, np.uint8))
call()
ts = []
for _ in range(reps):
    t = time.perf_counter()
    r = call()
    ts.append(time.perf_counter() - t)
print("SearchByProjection window %.0f: best %.3f ms, median %.3f ms, %d matches" % (radius, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, r[0]))
ctx.close()
