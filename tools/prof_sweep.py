"""bench.py's batch sweep on its own (no 32-stream context, no torch unless PROF_TORCH=1): isolates what else in the
benchmark process costs the small-batch rate."""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from __graft_entry__ import load_pkg

if os.environ.get("PROF_TORCH"):
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(1 << 20, device="cuda")
    torch.cuda.synchronize()
pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
args = argparse.Namespace(kind="dense", max_num=1024, precision=0)
out = bench.batch_sweep(pkg, synth, 480, 640, args, 0)
for r in out["rows"]:
    print(r)
