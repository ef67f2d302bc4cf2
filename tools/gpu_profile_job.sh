#!/bin/bash
# One GPU-box job: smoke, GPU tests, the default bench line, rocprofv3 kernel trace and the PMC passes.
# Usage (from the repo root on the box): bash tools/gpu_profile_job.sh <tag>     -> gpurun_out/<tag>_*
tag=${1:-r1x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 600 python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err; tail -c 1500 $O/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-seeds > $O/${tag}_bench_traced.json 2>$O/${tag}_trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/prof_$tag/$c -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-seeds > /dev/null 2>$O/${tag}_$c.err
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof_$tag/SQ -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-seeds > /dev/null 2>$O/${tag}_SQ.err
# the FP64 instruction mix and the matrix pipe: wave-level instruction counts by class, MFMA operations (x 512 = flops)
# and the cycles the matrix pipe is busy; eight SQ slots per pass, so two passes.  A counter this ROCm does not know fails its pass only.
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $O/prof_$tag/FP64 -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-seeds > /dev/null 2>$O/${tag}_FP64.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/prof_$tag/MFMA -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-seeds > /dev/null 2>$O/${tag}_MFMA.err
cd $R
# which kernel sources these numbers belong to (bench.py quotes them only on a library with the same hash)
# ... and the interior-point iterations of the timed launch (the same deterministic workload in every pass): bench.py divides the FP64 pass's flops by them
python - <<PY
import json
from conflict_rez_amd import engine
try:
    its = json.loads(open("$O/${tag}_bench_traced.json").read().strip().splitlines()[-1])["config"]["ipm_iterations_rank0"]
except Exception:
    its = None
json.dump({"csrc_sha16": engine.source_hash(), "command": "python bench.py --no-cpu-baseline --no-extras --no-seeds", "ipm_iterations_timed_launch": its}, open("$O/${tag}_meta.json", "w"))
PY
# ROCm 7.2 writes a rocpd sqlite database; turn it into the CSV summaries kept under profiles/
python tools/rocpd_summary.py $O/prof_$tag/trace/t_results.db $O/$tag
for c in FETCH_SIZE WRITE_SIZE SQ FP64 MFMA; do python tools/rocpd_summary.py $O/prof_$tag/$c/t_results.db $O/$tag $c; done
head -12 $O/${tag}_kernel_stats.csv; grep -h loop_kernel $O/${tag}_pmc_*.csv | cut -c1-160
find $O/prof_$tag -type f -size +8M -delete
