"""Builds profiles/traffic.json (what bench.py reports as roofline.traffic) from the counter summaries of
tools/collect_profiles.sh: per arithmetic mode the HBM bytes per launch of the fused first convolution, the kernel the
roofline of that mode is quoted on.  usage: make_traffic_json.py TAG   (reads profiles/TAG_hbm_counters[_precisionN].csv)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
B, W, H = 32, 640, 480
# what one launch must move at least: the u8 images in, the pooled 64-channel fp32 tensor out, the packed weights once
ALGORITHMIC = B * W * H + B * (W // 2) * (H // 2) * 64 * 4 + 9 * 64 * 64 * 4 + (9 * 64 + 64) * 4
out = {"batch": B, "width": W, "height": H,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python3 bench.py --no-cpu-baseline "
                 "--no-split-pass --no-sweep --steps 3 --warmup 1 --precision P` (tools/collect_profiles.sh, summaries in "
                 "profiles/%s_hbm_counters*.csv); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts "
                 "128-byte requests as 64 bytes)" % tag,
       "by_precision": {}}
for prec, name in ((0, "%s_hbm_counters.csv" % tag), (1, "%s_hbm_counters_precision1.csv" % tag), (3, "%s_hbm_counters_precision3.csv" % tag)):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        continue
    best = None
    for r in csv.DictReader(open(path)):
        k = r["kernel"]
        fused = ("conv_mfma_kernel<3, 2, 8, true" in k) or ("conv_bf16_kernel<2, true" in k) or ("conv_ws_kernel<true" in k)
        if fused and (best is None or int(r["launches"]) > int(best["launches"])):
            best = r
    if best is None:
        continue
    out["by_precision"][str(prec)] = {
        "kernel": best["kernel"], "fetch_size_kb": float(best["FETCH_SIZE_mean"]), "write_size_kb": float(best["WRITE_SIZE_mean"]),
        "hbm_bytes_per_launch": float(best["hbm_bytes_per_launch_corrected"]), "algorithmic_bytes_per_launch": ALGORITHMIC,
        "from": "profiles/" + name}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["by_precision"], indent=1))
