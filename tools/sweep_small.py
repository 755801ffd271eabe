"""The batch sweep of bench.py for the small batches only (one and two frames per call): pipelined frames/s and blocking
latency.  usage: sweep_small.py [precision] [batches, e.g. 1,2]"""
import argparse
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
args = argparse.Namespace(kind="dense", max_num=1024, precision=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
only = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2)
bench.SWEEP_BATCHES = only
out = bench.batch_sweep(pkg, synth, 480, 640, args, 0)
for r in out["rows"]:
    print(json.dumps(r))
