"""Deviation of every precision mode from the fp32 goldens of the reference nets (tests/golden/cnn_*.npz): per tensor
max |gpu - golden| / max |golden|, score-map max difference, key-point list equality.  Needs the GPU."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from __graft_entry__ import load_pkg
pkg = load_pkg()
nhwc = lambda a: np.ascontiguousarray(np.transpose(a, (1, 2, 0)))
names = ["cnn_96x128_s1_sparse.npz", "cnn_128x160_s7_dense.npz", "cnn_64x96_s3_sparse.npz"]
for prec, label in ((0, "fp32"), (3, "fp16 split"), (1, "bf16 split"), (2, "bf16")):
    worst = {}
    same = 0
    for n in names:
        g = np.load(os.path.join(root, "tests", "golden", n))
        H, W = g["img"].shape
        ctx = pkg.Context(pkg.default_config(W, H, precision=prec))
        r = ctx.extract(g["img"])
        for which, k in ((pkg.T_FEAT, "feat"), (pkg.T_JUNC, "junc"), (pkg.T_HEAT_RAW, "heat"), (pkg.T_DESCMAP, "desc")):
            ref = nhwc(g[k])
            worst[k] = max(worst.get(k, 0.0), float(np.abs(ctx.read_tensor(which) - ref).max() / np.abs(ref).max()))
        worst["score"] = max(worst.get("score", 0.0), float(np.abs(ctx.read_tensor(pkg.T_SCORE) - g["score"]).max()))
        same += int(np.array_equal(r["xy"], g["kp_xy"]))
        if np.array_equal(r["xy"], g["kp_xy"]):
            worst["desc_l2"] = max(worst.get("desc_l2", 0.0), float(np.linalg.norm(r["desc"] - g["kp_desc"], axis=1).max()))
        ctx.close()
    print(f"{label:11s} " + " ".join(f"{k} {v:.2e}" for k, v in worst.items()) + f"  key-point lists identical {same}/{len(names)}", flush=True)
