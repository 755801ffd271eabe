"""Blocking one-call latency and pipelined rate at a small batch size: ab_lat.py B REPS (environment switches apply)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
H, W, B = 480, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=1024))
scenes = [synth.make_scene(H, W, 1000 * s + 11, "dense") for s in range(B)]
frames = [[synth.frame_from_scene(scenes[s], H, W, t) for s in range(B)] for t in range(4)]
for k in range(20):
    ctx.extract_batch(frames[k % 4], to_dict=False)
t0 = time.perf_counter()
for k in range(reps):
    ctx.extract_batch(frames[k % 4], to_dict=False)
lat = (time.perf_counter() - t0) / reps
depth = 4
t0 = time.perf_counter()
for t in range(reps + depth):
    if t < reps:
        ctx.submit_batch(frames[t % 4], match_prev=True)
    if t >= depth - 1 and t - (depth - 1) < reps:
        ctx.wait_batch(to_dict=False)
dt = time.perf_counter() - t0
print("B=%d: blocking extract %.4f ms, pipelined %.1f frames/s" % (B, lat * 1e3, B * reps / dt))
ctx.close()
