#!/bin/bash
# Go / no-go of a lane-map variant: lone-instance phase stamps and the bench line (default workload and the throughput regime)
# for the product library and for variant libraries.  Usage (on the box): bash tools/gpu_lps_check.sh <tag> <stamps libs...> -- <plain libs...>
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  for g in 9 17; do timeout 120 python tools/lone_stamps.py $1 $g 2>&1 | tail -2; done
  shift
done
shift
for lib in "$@"; do
  echo "== $lib"
  n=$(basename $lib .so)
  CFZ_LIBRARY=$R/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --no-seeds > $O/${tag}_${n}_bench.json 2> $O/${tag}_${n}_bench.err
  CFZ_LIBRARY=$R/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --no-seeds --scenarios 8192 --raw-starts > $O/${tag}_${n}_bench_s8192.json 2>> $O/${tag}_${n}_bench.err
  python - <<PY
import json
for f in ("$O/${tag}_${n}_bench.json", "$O/${tag}_${n}_bench_s8192.json"):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        c = b["config"]; its = c.get("ipm_iterations_rank0") or 0
        print(f.split("/")[-1], "value", round(b["value"]), "ms/step", round(b["ms_per_step"], 3), "ipm iters/s", round(its / (b["ms_per_step"] * b["steps"] / 1e3)), "mean iters", c.get("mean_ipm_iters_timed_region"), "per CU", c.get("instances_per_cu"))
    except Exception as ex:
        print(f, "failed", ex)
PY
done
