#!/bin/bash
# rocprofv3 kernel trace of the planning kernels (state_ws, collocation plans, joint plan) -> gpurun_out/<tag>_planning_*
# Usage (on the GPU box, from the repo root): bash tools/gpu_profile_planning.sh <tag> [agents]
tag=${1:-r1x}
agents=${2:-vehicle_0,vehicle_1,vehicle_2,vehicle_3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_${tag}_planning
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_planning -o t -- python3 $R/tools/joint_timing.py $agents > $O/${tag}_planning.log 2>$O/${tag}_planning.err
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/prof_${tag}_planning_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/prof_${tag}_planning_$c -o t -- python3 $R/tools/joint_timing.py $agents > /dev/null 2>$O/${tag}_planning_$c.err
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do python tools/rocpd_summary.py $O/prof_${tag}_planning_$c/t_results.db $O/${tag}_planning $c; grep colloc_kernel $O/${tag}_planning_pmc_$c.csv | cut -c1-200; done
tail -3 $O/${tag}_planning.log
python tools/rocpd_summary.py $O/prof_${tag}_planning/t_results.db $O/${tag}_planning
cat $O/${tag}_planning_kernel_stats.csv | cut -c1-160
find $O/prof_${tag}_planning* -type f -size +8M -delete
