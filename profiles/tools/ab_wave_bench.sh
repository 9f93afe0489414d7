#!/bin/bash
# the headline step with k_sor_stream (flow_opts_off bit 3) and with k_sor_wave at several band targets / slice counts, alternating, inside one box
set -e
mkdir -p gpurun_out
run() { timeout -k 10 300 python3 bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-sequence-leg --no-dropin-leg --no-small-step-leg --flow-opts-off $1 --flow-slices $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('opts_off', $1, 'slices', $2, 'value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],1), 'solver busy ms/step', round(r.get('solver_busy_ms_per_step') or 0,1), 'avg_launch_us', round(r.get('avg_launch_us') or 0,1))"; }
for rep in 1 2; do
  run 8 0
  run 0 0
  run $((680*256)) 0
  run $((1024*256)) 0
  run 0 2
  run 0 1
done 2>&1 | tee gpurun_out/ab_wave_bench.txt
