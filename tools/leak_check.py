"""Create, use and destroy contexts repeatedly and watch the free device memory: a context must give back everything it
took (device arrays, the grown work arrays of the matcher entry points, pinned blocks)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg

pkg = load_pkg()
synth = importlib.import_module("ppg_slam_amd.synth")
hip = C.CDLL("libamdhip64.so.7")


def free_mib():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20


H, W = 480, 640
img = [synth.make_frame(H, W, 11, t, "dense") for t in (0, 1)]
rng = np.random.default_rng(3)
first = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=4, max_num=1024, precision=(0, 1, 3)[it % 3]))
    f0, f1 = ctx.extract_batch(img)
    ctx.match_prev_batch(2, 0.9) if False else None
    ctx.search_for_initialization(0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 50)
    ctx.search_for_initialization(0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 0)
    g = synth.random_map_graph(rng, f1["desc"], f1["un"], 500, 1200, f1["edge_idx"])
    m = g["desc"].shape[0]
    ctx.extend_map_matches(0.8, 5.0, g, f1["desc"], f1["un"], f1["edge_idx"], f1["conn_off"], f1["conn_idx"], np.zeros(m, np.uint8),
                           np.full(f1["n"], -1, np.int32), np.full(f1["n_edges"], -1, np.int32))
    ctx.match_projected(f1["un"], f1["desc"], f0["desc"], f0["un"], 15.0, 0.8, 0.0, None, np.ones(f0["n"], np.uint8), np.zeros(f1["n"], np.uint8))
    for k in range(6):
        ctx.submit_batch(img, match_prev=k > 0)
        if k >= 3:
            ctx.wait_batch(to_dict=False)
    for k in range(3):
        ctx.wait_batch(to_dict=False)
    ctx.close()
    fm = free_mib()
    if first is None:
        first = fm
    if it % 5 == 0 or fm < first - 64:
        print("iteration %d: %.0f MiB free (first %.0f)" % (it, fm, first), flush=True)
    assert fm > first - 64, "device memory is not given back"
print("no growth over the iterations: %.0f MiB free at the end, %.0f after the first" % (fm, first))
