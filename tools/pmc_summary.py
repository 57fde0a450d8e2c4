#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs found under a directory:
per kernel name and counter, the number of dispatches and the mean counter value
per dispatch.  Usage: python tools/pmc_summary.py gpurun_out/<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "").replace("yl::", "")
    i = name.find("(")
    return (name[:i] if i > 0 else name)[:60]


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            for row in rd:
                k = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                c = row.get("Counter_Name") or row.get("Counter Name") or ""
                v = row.get("Counter_Value") or row.get("Counter Value") or "0"
                try:
                    val = float(v)
                except ValueError:
                    continue
                a = acc[(short(k), c)]
                a[0] += 1
                a[1] += val
    print("# files: %d" % len(files))
    print("%-62s %-28s %8s %16s" % ("kernel", "counter", "dispatch", "mean/dispatch"))
    for (k, c), (n, s) in sorted(acc.items()):
        if "conv" in k or "shortcut" in k or "quantize" in k or "pack" in k:
            print("%-62s %-28s %8d %16.1f" % (k, c, n, s / max(n, 1)))


if __name__ == "__main__":
    main()
