#!/usr/bin/env python3
"""Per-dispatch mean of every PMC counter of the kernels whose name contains <substr>, from a rocprofv3
`--pmc ... --output-format csv` result directory.  Usage: python tools/pmc_summary.py <dir> <substr>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, sub):
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", root)
        return
    acc = defaultdict(lambda: [0.0, 0])
    names = set()
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                kn = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                if sub not in kn:
                    continue
                names.add(kn[:90])
                c = row.get("Counter_Name") or row.get("Counter Name")
                v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                a = acc[c]
                a[0] += v
                a[1] += 1
    print(f"# {root}: kernels matching '{sub}': {sorted(names)}")
    for c in sorted(acc):
        s, n = acc[c]
        print(f"{c:<28} dispatches={n:>5} mean={s / n:>18.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
