#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters from a counter_collection.csv (one pass = one --pmc set).

usage: python profiles/pmc_summary.py <dir or csv> [kernel-substring ...]
Prints, per kernel name and counter, dispatch count and mean value per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- the corrected
figure is printed next to the raw one."""
import collections
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    filt = sys.argv[2:]
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if filt and not any(x in name for x in filt):
                    continue
                key = (name[:90], row["Counter_Name"])
                agg[key][0] += 1
                agg[key][1] += float(row["Counter_Value"])
    print(f"{'dispatches':>10} {'mean/dispatch':>16}  counter  kernel")
    for (name, ctr), (n, tot) in sorted(agg.items()):
        mean = tot / n
        extra = ""
        if ctr == "FETCH_SIZE":
            extra = f"   (= {mean * 1024 / 1e6:.2f} MB raw, {2 * mean * 1024 / 1e6:.2f} MB with the gfx950 x2 read correction)"
        elif ctr == "WRITE_SIZE":
            extra = f"   (= {mean * 1024 / 1e6:.2f} MB)"
        print(f"{n:10d} {mean:16.1f}  {ctr}  {name}{extra}")


if __name__ == "__main__":
    main()
