# Samples the sclk level of every card while (a) the FMA microbenchmark and (b) the solver bench run.
cd $GRAFT_REPO_ROOT
sample() { for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo -n "$(dirname $(dirname $f) | xargs basename):$(grep '\*' $f | tr -d '\n') "; done; echo; }
echo "idle:"; sample
( for i in 1 2 3 4 5 6 7 8; do ./tools/_clock_test > /dev/null; done ) &
P=$!; sleep 2; echo "microbench:"; for i in 1 2 3; do sample; sleep 1; done; wait $P
python bench.py --no-cpu-baseline --steps 150 --warmup 5 --scenarios 4096 > gpurun_out/clk_bench.json 2>&1 &
P=$!; sleep 25; echo "solver:"; for i in 1 2 3; do sample; sleep 1; done; wait $P
python tools/bench_line.py < gpurun_out/clk_bench.json
