#!/bin/bash
# Small steps by dense-flow slice count: frame pairs/s of the streams workload at S x T pairs per step with the flow cut into 1..4 concurrent slices (bench.py --flow-slices).
# One gpurun call:  bash profiles/tools/small_step_slices.sh "8 4" "16 2" > gpurun_out/small_step_slices.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for cfg in "$@"; do
  set -- $cfg
  for p in ${SLICES:-1 2 3 4}; do
    line=$(timeout -k 10 300 python3 bench.py --streams $1 --frames-per-step $2 --steps 16 --warmup 3 --flow-slices $p --no-cpu-baseline --no-sequence-leg 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('%3d pairs/step (%2d streams x %d), %d flow slice(s): %7.1f pairs/s  %6.1f ms/step  dense flow %6.1f  tails %6.1f  host cores %.1f' % (d['config']['frame_pairs_per_step'], $1, $2, $p, d['value'], d['ms_per_step'], st['dense_flow'], st['tails'], d['host_cores_busy']))"
  done
done
