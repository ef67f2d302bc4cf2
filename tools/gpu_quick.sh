#!/bin/bash
# Quick GPU check of a kernel change: parity tests of the MPC path, the bench line (no CPU leg), kernel trace + WRITE/FETCH passes.
# Usage (on the box): bash tools/gpu_quick.sh <tag>    -> gpurun_out/<tag>_*
tag=${1:-q}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py -x -q --timeout 200 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout -s KILL 200 python bench.py --no-cpu-baseline > $O/${tag}_bench.json 2> $O/${tag}_bench.err
python - <<PY
import json
b=json.loads(open("$O/${tag}_bench.json").read().strip().splitlines()[-1])
print("value", b["value"], "ms/step", b["ms_per_step"], "cold", b["cold_step"]["value"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag/trace -o t -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>$O/${tag}_trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/prof_$tag/$c -o t -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>$O/${tag}_$c.err
done
cd $R
python tools/rocpd_summary.py $O/prof_$tag/trace/t_results.db $O/$tag > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python tools/rocpd_summary.py $O/prof_$tag/$c/t_results.db $O/$tag $c > /dev/null; done
grep -h loop_kernel $O/${tag}_kernel_stats.csv | head -3; grep -h loop_kernel $O/${tag}_pmc_*.csv | cut -c1-160
find $O/prof_$tag -type f -size +8M -delete
