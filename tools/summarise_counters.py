"""Summarises rocprofv3 --pmc output (one or more *_counter_collection.csv) per (kernel, grid size): mean counter values per
launch, plus the derived columns the profiles/ files carry:
  hbm_bytes_per_launch_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE tallies 128-byte read
                                    requests at 64 bytes; WRITE_SIZE is exact for 16-byte-per-lane stores)
  mfma_busy_fraction             = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (four SIMDs per CU)
usage: summarise_counters.py OUT.csv DIR [DIR ...]"""
import collections
import csv
import glob
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.OrderedDict()   # (kernel, grid) -> counter -> [sum, dispatches]
    order = []
    for d in dirs:
        for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
            per_dispatch = collections.defaultdict(float)
            meta = {}
            for r in csv.DictReader(open(f)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                meta[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["Grid_Size"]))
            for (disp, cname), v in per_dispatch.items():
                k = meta[disp]
                if cname not in order:
                    order.append(cname)
                slot = acc.setdefault(k, {}).setdefault(cname, [0.0, 0])
                slot[0] += v
                slot[1] += 1
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        head = ["kernel", "grid_size", "launches"] + [c + "_mean" for c in order]
        derived = []
        if "FETCH_SIZE" in order and "WRITE_SIZE" in order:
            derived.append("hbm_bytes_per_launch_corrected")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in order and "SQ_BUSY_CU_CYCLES" in order:
            derived.append("mfma_busy_fraction")
        w.writerow(head + derived)
        rows = sorted(acc.items(), key=lambda kv: -max(v[0] for v in kv[1].values()))
        for (kernel, grid), cs in rows:
            n = max(v[1] for v in cs.values())
            mean = {c: (cs[c][0] / cs[c][1] if c in cs else float("nan")) for c in order}
            row = [kernel, grid, n] + ["%.1f" % mean[c] for c in order]
            if "hbm_bytes_per_launch_corrected" in derived:
                row.append("%.0f" % ((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024))
            if "mfma_busy_fraction" in derived:
                b = mean["SQ_BUSY_CU_CYCLES"]
                row.append("%.4f" % (mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * b) if b else 0.0))
            w.writerow(row)


if __name__ == "__main__":
    main()
