"""ExtendMapMatches on the problem bench.py times (two dense VGA frames, a local map of 600 points and 1500 edges),
repeated: under `rocprofv3 --kernel-trace --stats` the statistics show how the blocking call divides into the
distance kernels, the one-wave walk and the copies around them."""
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
ctx = pkg.Context(pkg.default_config(W, H, max_batch=2, max_num=1024))
scene = synth.make_scene(H, W, 11, "dense")
f0, f1 = ctx.extract_batch([synth.frame_from_scene(scene, H, W, 0), synth.frame_from_scene(scene, H, W, 1)])
rng = np.random.default_rng(1)
g = synth.random_map_graph(rng, f1["desc"], f1["un"], min(600, f1["n"]), 1500)
m = g["desc"].shape[0]
xargs = (0.8, 5.0, g, f1["desc"], f1["un"], f1["edge_idx"], f1["conn_off"], f1["conn_idx"], np.zeros(m, np.uint8),
         np.full(f1["n"], -1, np.int32), np.full(f1["n_edges"], -1, np.int32))
ctx.extend_map_matches(*xargs)
ts = []
for _ in range(reps):
    t = time.perf_counter()
    r = ctx.extend_map_matches(*xargs)
    ts.append(time.perf_counter() - t)
print("ExtendMapMatches: best %.3f ms, median %.3f ms, %d matches" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, int(r[0])))
ctx.close()
