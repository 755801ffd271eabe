"""One camera stream per GPU with the lens model of the two multi-GPU configurations (Kannala-Brandt, the coefficients of
config/TUM-VI.yaml, scaled for the 1024 x 1024 variant) against the same stream with D = 0: pipelined frames/s four deep
and blocking extract latency, host images in.  With D != 0 every frame adds the heat-map remap and the iterative key-point
undistortion.  usage: prof_fisheye.py [steps]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for (H, W, scale) in ((512, 512, 1.0), (1024, 1024, 2.0)):
    scene = synth.make_scene(H, W, 11, "dense")
    frames = [synth.frame_from_scene(scene, H, W, t) for t in range(8)]
    for lens in ("D = 0", "Kannala-Brandt (TUM-VI)"):
        cfg = pkg.default_config(W, H, max_batch=1, max_num=1024)
        K = [190.978477 * scale, 0.0, 254.931706 * scale, 0.0, 190.973307 * scale, 256.897442 * scale, 0.0, 0.0, 1.0]
        for i in range(9):
            cfg.K[i] = K[i]
        if lens != "D = 0":
            cfg.fisheye = 1
            for i, d in enumerate((0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736)):
                cfg.D[i] = d
        ctx = pkg.Context(cfg)
        for t in range(20):
            ctx.extract(frames[t % 8])
        t0 = time.perf_counter()
        for t in range(200):
            out = ctx.extract_batch([frames[t % 8]], to_dict=False)
        lat = (time.perf_counter() - t0) / 200
        depth = 4
        ctx.extract_batch([frames[0]], to_dict=False)
        t0 = time.perf_counter()
        for t in range(steps + depth):
            if t < steps:
                ctx.submit_batch([frames[t % 8]], match_prev=True)
            if t >= depth - 1 and t - (depth - 1) < steps:
                ctx.wait_batch(to_dict=False)
        dt = time.perf_counter() - t0
        n = ctx.extract(frames[3])["n"]
        ctx.close()
        print("%dx%d %-24s blocking extract %.3f ms, pipelined %.1f frames/s (%d key points in a frame)"
              % (W, H, lens, 1e3 * lat, steps / dt, n), flush=True)
