#!/usr/bin/env python
"""Per-DISPATCH view of rocprofv3 --pmc counter_collection CSVs under a directory (pmc_summary.py averages
per kernel name; a micro network launches one kernel instance with several shapes).
Usage: python tools/pmc_dispatch.py <dir> [substring of kernel names to keep]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    keep = sys.argv[2] if len(sys.argv) > 2 else "conv"
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        rows = defaultdict(dict)
        names = {}
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if keep not in k:
                    continue
                d = int(row.get("Dispatch_Id", 0))
                names[d] = k.replace("void ", "").replace("yl::", "")[:70]
                try:
                    rows[d][row["Counter_Name"]] = float(row["Counter_Value"])
                except (KeyError, ValueError):
                    pass
        print("#", f)
        for d in sorted(rows):
            print("%5d %-70s %s" % (d, names[d], " ".join("%s=%.4g" % kv for kv in sorted(rows[d].items()))))


if __name__ == "__main__":
    main()
