#!/bin/bash
# k_peac_grow's LDS level capacity (PG_LDS_N seeds x 40 bytes: 2048 = 93 KB per workgroup, 1024 = 52 KB, 512 = 32 KB): what the solver's waves find free on a compute unit beside a grow workgroup
set -e
mkdir -p gpurun_out
run() { timeout -k 10 300 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sequence-leg --no-dropin-leg --no-small-step-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; s=d['stage_ms_per_step']
print('PG_LDS_N', '$1', round(d['value'],1), round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'solver busy', round(r['solver_busy_ms_per_step'],1))"; }
for rep in 1 2; do for n in 2048 1024 512; do
  (cd sindslam_amd/csrc && touch peac_kernels.hip && make EXTRA=-DPG_LDS_N=$n > /dev/null 2>&1)
  run $n
done; done
