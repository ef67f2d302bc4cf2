#!/bin/bash
# same-box A/B of two libraries: bench quick (no cpu, no extras, no seeds), 8192-scenario throughput
cd $GRAFT_REPO_ROOT
for l in tools/_lib_head.so conflict_rez_amd/libconfrez_hip.so tools/_lib_head.so conflict_rez_amd/libconfrez_hip.so; do
  CFZ_LIBRARY=$l timeout 300 python bench.py --no-cpu-baseline --no-extras --no-seeds 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', 'value %.0f all %.0f ms %.4f kernel_ms %.3f cold_ms %.3f its %d' % (b['value'], b['value_all'], b['ms_per_step'], b['roofline']['kernel_ms_per_launch'], b['cold_step']['kernel_ms'], b['config']['ipm_iterations_rank0']))"
done
for l in tools/_lib_head.so conflict_rez_amd/libconfrez_hip.so; do
  CFZ_LIBRARY=$l timeout 300 python bench.py --no-cpu-baseline --no-extras --no-seeds --scenarios 8192 --raw-starts 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l 8192:', 'all %.0f ms %.4f its/s %.3e' % (b['value_all'], b['ms_per_step'], b['config']['ipm_iterations_rank0'] / (b['ms_per_step'] * b['steps'] * 1e-3)))"
done
