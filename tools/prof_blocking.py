"""Runs the blocking batch API (CNN, then post-processing, then matching; nothing overlaps) a few times, so that
a kernel trace of this script shows every kernel's duration when it has the GPU to itself."""
import importlib.util, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ppg_slam_amd", os.path.join(root, "ppg-slam_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(root, "ppg-slam_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["ppg_slam_amd"] = pkg; spec.loader.exec_module(pkg)
from ppg_slam_amd import synth
H, W, B = int(os.environ.get("PB_H", 480)), int(os.environ.get("PB_W", 640)), int(os.environ.get("PB_B", 16))
cfg = pkg.default_config(W, H, max_num=int(os.environ.get("PB_MAXNUM", 1024)), precision=int(os.environ.get("PB_PREC", 0))); cfg.max_batch = B
ctx = pkg.Context(cfg)
scenes = [synth.make_scene(H + 64, W + 128, 100 + s, "dense") for s in range(B)]
prev = None
for t in range(6):
    frames = [synth.frame_from_scene(scenes[s], H, W, t) for s in range(B)]
    ctx.submit_batch(frames, match_prev=t > 0)
    outs = ctx.wait_batch()
print("ok", len(outs), outs[0].n if hasattr(outs[0], "n") else "")
