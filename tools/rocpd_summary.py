#!/usr/bin/env python3
"""rocprofv3 results database (rocpd sqlite, the default output of ROCm 7.2) -> the two CSV summaries kept in profiles/:
   <out>_kernel_stats.csv  name, calls, total/mean/min/max duration (what `--stats` prints), then every dispatch of the solver kernels
   <out>_pmc_<name>.csv    per kernel and counter: dispatches, mean and sum of the counter value
Usage: rocpd_summary.py <results.db> <out_prefix> [pmc_name]"""
import collections
import sqlite3
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0]


def main(db_path, out, pmc=None):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, scratch_size from kernels order by start"))
    if pmc is None:
        acc = collections.OrderedDict()
        for r in rows:
            acc.setdefault(short(r[0]), []).append((r[2] - r[1]) / 1e3)
        tot = sum(sum(v) for v in acc.values())
        with open(out + "_kernel_stats.csv", "w") as f:
            f.write("kernel,calls,total_us,mean_us,min_us,max_us,percent\n")
            for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                f.write(f"{k},{len(v)},{sum(v):.3f},{sum(v) / len(v):.3f},{min(v):.3f},{max(v):.3f},{100 * sum(v) / tot:.3f}\n")
            f.write("\ndispatch,kernel,duration_us,grid,workgroup,lds_bytes,arch_vgpr,accum_vgpr,scratch_bytes\n")
            for i, r in enumerate(rows):
                if "rocclr" not in r[0]:
                    f.write(f"{i},{short(r[0])},{(r[2] - r[1]) / 1e3:.3f},{r[3]},{r[4]},{r[5]},{r[6]},{r[7]},{r[8]}\n")
    else:
        cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
        q = list(cur.execute("select * from counters_collection"))
        ik, ic, iv = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        idisp = cols.index("dispatch_id")
        per = collections.defaultdict(float)
        for r in q:
            per[(short(r[ik]), r[ic], r[idisp])] += float(r[iv])  # a counter comes in one row per XCD / instance: sum them
        for (k, c, _), v in per.items():
            acc[k][c].append(v)
        with open(f"{out}_pmc_{pmc}.csv", "w") as f:
            f.write("kernel,counter,dispatches,mean_per_dispatch,values_per_dispatch\n")
            for k in sorted(acc):
                for c, v in sorted(acc[k].items()):
                    f.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g},{' '.join('%.6g' % x for x in v)}\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
