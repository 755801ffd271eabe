"""Blocking 32-frame batches (no pipelining): under `rocprofv3 --kernel-trace --stats` this gives every kernel's duration
without a concurrent convolution beside it."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
H, W, B = 480, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=1024, precision=prec))
scenes = [synth.make_scene(H, W, 1000 * s + 11, "dense") for s in range(B)]
for t in range(6):
    ctx.extract_batch([synth.frame_from_scene(scenes[s], H, W, t) for s in range(B)], to_dict=False)
    if t:
        ctx.match_prev_batch(B, 0.9)
ctx.close()
