"""Long run of the wave-specialised convolutions against the one-tile-per-work-group kernels: N batches of 32 frames in mode 1
(and a shorter leg in modes 2 and 3 with the kernels forced on), every batch's network outputs and extracted frames compared
bit for bit with the reference results of the same images.  usage: soak_ws.py [batches]"""
import importlib
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
H, W, B, T = 480, 640, 32, 6
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
scenes = [synth.make_scene(H, W, 9000 + 13 * s, "dense" if s % 4 else "sparse") for s in range(B)]
frames = [[synth.frame_from_scene(scenes[s], H, W, t) for s in range(B)] for t in range(T)]


def digest(ctx, outs):
    d = 0
    for which in (pkg.T_FEAT, pkg.T_JUNC, pkg.T_HEAT_RAW, pkg.T_DESCMAP):
        for f in (0, B // 2, B - 1):
            d = zlib.crc32(ctx.read_tensor(which, f).tobytes(), d)
    for o in outs:
        d = zlib.crc32(o["xy"].tobytes(), d)
        d = zlib.crc32(o["desc"].tobytes(), d)
        d = zlib.crc32(o["edge_idx"].tobytes(), d)
    return d


for prec, batches in ((1, N), (3, max(N // 5, 10)), (2, max(N // 5, 10))):
    os.environ["PPG_CONV_WS"] = "0"
    ref = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=1024, precision=prec))
    want = [digest(ref, ref.extract_batch(frames[t])) for t in range(T)]
    ref.close()
    os.environ["PPG_CONV_WS"] = "1"
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=1024, precision=prec))
    bad = 0
    for k in range(batches):
        t = (k * 5) % T
        if digest(ctx, ctx.extract_batch(frames[t])) != want[t]:
            bad += 1
            print("precision", prec, "batch", k, "differs", flush=True)
        if k % 50 == 49:
            print("precision", prec, k + 1, "batches", flush=True)
    ctx.close()
    print("precision %d: %d batches of %d frames, %d differ" % (prec, batches, B, bad), flush=True)
    if bad:
        sys.exit(1)
print("soak ok")
