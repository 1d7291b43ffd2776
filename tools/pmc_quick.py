#!/usr/bin/env python3
"""Per kernel and counter: dispatches and the average per dispatch, from the *_counter_collection.csv files under a rocprofv3 -d directory.
    python tools/pmc_quick.py DIR [kernel-substring ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    subs = sys.argv[2:]
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"]
                if subs and not any(s in name for s in subs):
                    continue
                m = re.search(r"(\w+_kernel(?:<[^(]*>)?)", name)
                per[(m.group(1) if m else name[:60], r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for (name, counter), dd in sorted(per.items()):
        vals = list(dd.values())
        print(f"{name:60s} {counter:24s} n={len(vals):3d} avg={sum(vals) / len(vals):.4g}")


if __name__ == "__main__":
    main()
