#!/bin/bash
# rocprofv3 passes over the planning extras of the bench line (`python bench.py --extras-only`: BASELINE configs[1] and configs[3] exactly
# as `extra` of the driver line measures them): kernel trace, FETCH_SIZE, WRITE_SIZE and the SQ counters, per dispatch of colloc_kernel.
# bench.py quotes them as extra["configs[k]"].roofline.traffic / valu_active_frac when the source hash matches (<tag>_extras_meta.json).
#   Usage (GPU box, repo root): bash tools/gpu_profile_extras.sh <tag>
tag=${1:-r5x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_${tag}_ex
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_ex/trace -o t -- python3 $R/bench.py --extras-only --no-cpu-baseline > $O/${tag}_extras.json 2>$O/${tag}_ex.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/prof_${tag}_ex/$c -o t -- python3 $R/bench.py --extras-only --no-cpu-baseline > /dev/null 2>$O/${tag}_ex_$c.err
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof_${tag}_ex/SQ -o t -- python3 $R/bench.py --extras-only --no-cpu-baseline > /dev/null 2>$O/${tag}_ex_SQ.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $O/prof_${tag}_ex/FP64 -o t -- python3 $R/bench.py --extras-only --no-cpu-baseline > /dev/null 2>$O/${tag}_ex_FP64.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/prof_${tag}_ex/MFMA -o t -- python3 $R/bench.py --extras-only --no-cpu-baseline > /dev/null 2>$O/${tag}_ex_MFMA.err
cd $R
python tools/rocpd_summary.py $O/prof_${tag}_ex/trace/t_results.db $O/${tag}_extras > /dev/null
for c in FETCH_SIZE WRITE_SIZE SQ FP64 MFMA; do python tools/rocpd_summary.py $O/prof_${tag}_ex/$c/t_results.db $O/${tag}_extras $c > /dev/null; done
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from conflict_rez_amd import engine
import hashlib
json.dump({"csrc_sha16": engine.source_hash(), "bench_sha16": hashlib.sha256(open("$R/bench.py", "rb").read()).hexdigest()[:16],
           "command": "python bench.py --extras-only --no-cpu-baseline"}, open("$O/${tag}_extras_meta.json", "w"))
PY
tail -c 600 $O/${tag}_extras.json; echo; grep -h "colloc_kernel" $O/${tag}_extras_kernel_stats.csv $O/${tag}_extras_pmc_*.csv | cut -c1-300
find $O/prof_${tag}_ex -type f -size +8M -delete
