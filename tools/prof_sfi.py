"""SearchForInitialization with the reference's 50-pixel windows (Tracking.cpp:525) on two dense VGA frames, repeated:
under `rocprofv3 --kernel-trace --stats` the statistics show how the blocking call divides into kernels and copies."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
H, W = 480, 640
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
window = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = pkg.Context(pkg.default_config(W, H, max_batch=2, max_num=1024))
scene = synth.make_scene(H, W, 11, "dense")
f0, f1 = ctx.extract_batch([synth.frame_from_scene(scene, H, W, 0), synth.frame_from_scene(scene, H, W, 1)])
call = lambda: ctx.search_for_initialization(0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], window)
call()
ts = []
for _ in range(reps):
    t = time.perf_counter()
    r = call()
    ts.append(time.perf_counter() - t)
print("SearchForInitialization window %d: best %.3f ms, median %.3f ms" % (window, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3))
ctx.close()
