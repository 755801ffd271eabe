"""Report of tools/gpu_iso.sh: per-kernel time per 32-frame batch from the statistics (total duration / number of batches) and every
convolution launch of the last batch with its algorithmic TFLOP/s.  usage: iso_report.py STATS.csv TRACE.csv"""
import csv
import sys

stats, trace = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(stats)))
batches = max(int(r["Calls"]) for r in rows if "conv_ws_kernel<true" in r["Name"] or "8, true" in r["Name"] or "<2, true" in r["Name"])
conv = post = 0.0
print("per 32-frame batch (%d batches in the run):" % batches)
for r in rows:
    per = int(r["TotalDurationNs"]) / 1e3 / batches
    name = r["Name"].replace("void ", "").replace("ppg::", "")
    if "conv_" in name:
        conv += per
    elif "rocclr" not in name and "upload" not in name:
        post += per
    print("  %-62s %8.1f us (%s launches, mean %.1f us)" % (name[:62], per, r["Calls"], float(r["AverageNs"]) / 1e3))
print("convolutions %.0f us, post-processing and match %.0f us per batch" % (conv, post))
tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
convs = [r for r in tr if "conv_" in r["Kernel_Name"]]
per = len(convs) // batches
gf = [736.1, 181.2, 181.2, 90.6, 181.2, 45.3, 45.3, 271.8, 5.1, 45.3, 11.3, 20.1]
names = ["conv1a+conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "heads (3 x 128->256)", "convPb 1x1", "EdgeHeatmap 1", "EdgeHeatmap 2", "convDb 1x1"]
print("\nconvolution launches of the last batch:")
for i, r in enumerate(convs[-per:]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("  %-22s %-50s %8.1f us %7.1f TFLOP/s" % (names[i] if i < len(names) else "?", r["Kernel_Name"].replace("void ppg::", "")[:50], d, (gf[i] if i < len(gf) else 0) / d * 1e3))
