"""Pipelined submit/wait at a small batch size (four deep, host images): under `rocprofv3 --kernel-trace` the trace shows
which stream bounds the rate."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
H, W, B = 480, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
idle = None
if os.environ.get("PROF_IDLE_CTX"):   # a second, idle context of 32 streams beside the measured one (what bench.py's sweep runs beside)
    idle = pkg.Context(pkg.default_config(W, H, max_batch=32, max_num=1024))
ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=1024))
scenes = [synth.make_scene(H, W, 1000 * s + 11, "dense") for s in range(B)]
frames = [[synth.frame_from_scene(scenes[s], H, W, t) for s in range(B)] for t in range(8)]
depth = 4
ctx.extract_batch(frames[0], to_dict=False)
t0 = time.perf_counter()
t_sub = t_wait = 0.0
for t in range(steps + depth):
    if t < steps:
        ta = time.perf_counter()
        ctx.submit_batch(frames[t % 8], match_prev=True)
        t_sub += time.perf_counter() - ta
    if t >= depth - 1 and t - (depth - 1) < steps:
        ta = time.perf_counter()
        ctx.wait_batch(to_dict=False)
        t_wait += time.perf_counter() - ta
dt = time.perf_counter() - t0
print("B=%d: %.1f frames/s, %.3f ms per step (host: %.3f ms in submit_batch, %.3f ms in wait_batch per step)"
      % (B, B * steps / dt, 1e3 * dt / steps, 1e3 * t_sub / steps, 1e3 * t_wait / steps))
ctx.close()
if idle is not None:
    idle.close()
