#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv -> per-kernel, per-counter mean over dispatches (profiles/*.csv)."""
import collections
import csv
import sys


def kname(s):
    s = s.replace("(anonymous namespace)::", "")
    return s.split("(")[0]


def main(src, dst):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(src)):
        acc[kname(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    with open(dst, "w") as f:
        f.write("kernel,counter,dispatches,mean_per_dispatch\n")
        for k in sorted(acc):
            for c, v in sorted(acc[k].items()):
                f.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
