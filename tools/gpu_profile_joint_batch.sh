#!/bin/bash
# rocprofv3 passes over BASELINE configs[3] at its named batch (tools/joint_batch_timing.py 256): kernel trace, FETCH_SIZE,
# WRITE_SIZE and the SQ counters of colloc_kernel.   Usage (GPU box, repo root): bash tools/gpu_profile_joint_batch.sh <tag>
tag=${1:-r2x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_${tag}_jb
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_jb/trace -o t -- python3 $R/tools/joint_batch_timing.py 256 > $O/${tag}_jb.log 2>$O/${tag}_jb.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/prof_${tag}_jb/$c -o t -- python3 $R/tools/joint_batch_timing.py 256 > /dev/null 2>$O/${tag}_jb_$c.err
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof_${tag}_jb/SQ -o t -- python3 $R/tools/joint_batch_timing.py 256 > /dev/null 2>$O/${tag}_jb_SQ.err
cd $R
python tools/rocpd_summary.py $O/prof_${tag}_jb/trace/t_results.db $O/${tag}_jointbatch > /dev/null
for c in FETCH_SIZE WRITE_SIZE SQ; do python tools/rocpd_summary.py $O/prof_${tag}_jb/$c/t_results.db $O/${tag}_jointbatch $c > /dev/null; done
tail -1 $O/${tag}_jb.log; grep -h "colloc_kernel" $O/${tag}_jointbatch_kernel_stats.csv $O/${tag}_jointbatch_pmc_*.csv | cut -c1-200
find $O/prof_${tag}_jb -type f -size +8M -delete
