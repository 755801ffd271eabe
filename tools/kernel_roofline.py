"""Joins the isolated kernel durations (profiles/rNN_kernel_stats_isolated.csv: blocking 32-frame batches, nothing beside a
kernel) with the HBM bytes per launch from the counter passes (profiles/rNN_hbm_counters.csv) into one table: duration,
bytes, GB/s and the share of the 8 TB/s HBM roof per kernel.  usage: kernel_roofline.py [tag]"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
HBM_PEAK = 8.0e12


def short(name):
    name = name.replace("void ", "").replace("ppg::", "")
    return name.split("(")[0]


dur = {}
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_isolated.csv"))):
    dur.setdefault(short(r["Name"]), []).append((float(r["AverageNs"]) * 1e-9, int(r["Calls"]), float(r["MinNs"]) * 1e-9, float(r["MaxNs"]) * 1e-9))
rows = {}
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_hbm_counters.csv"))):
    k = short(r["kernel"])
    b, n = float(r["hbm_bytes_per_launch_corrected"]), int(r["launches"])
    e = rows.setdefault(k, [0.0, 0])
    e[0] += b * n
    e[1] += n
print("%-44s %10s %12s %10s %8s" % ("kernel (32 frames per launch)", "mean us", "HBM MB", "GB/s", "of 8 TB/s"))
out = []
for k, (bytes_total, launches) in rows.items():
    if k not in dur:
        continue
    t = sum(a * c for a, c, _, _ in dur[k]) / sum(c for _, c, _, _ in dur[k])
    mb = bytes_total / launches
    out.append((t, k, mb))
for t, k, mb in sorted(out, reverse=True):
    print("%-44s %10.1f %12.1f %10.0f %7.1f%%" % (k[:44], t * 1e6, mb / 1e6, mb / t / 1e9, 100 * mb / t / HBM_PEAK))
