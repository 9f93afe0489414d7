#!/bin/bash
# What a step of S chunks x T frames costs on one GPU (streams workload, pipelined): the table the N-rank sequence job's chunk count is chosen from (bench.py sequence_streams).
# One gpurun call:  bash profiles/tools/step_shape_sweep.sh > gpurun_out/step_shape_sweep.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for cfg in "4 8" "6 6" "8 4" "8 8" "12 3" "12 4" "16 2" "16 3" "16 4" "24 2" "24 3" "32 2" "32 3" "48 2" "64 2" "28 8"; do
  set -- $cfg
  line=$(timeout -k 10 300 python3 bench.py --streams $1 --frames-per-step $2 --steps 16 --warmup 3 --no-cpu-baseline --no-sequence-leg 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('%3d pairs/step (%2d chunks x %d frames): %7.1f pairs/s  %6.1f ms/step  dense flow %6.1f  tails %6.1f  host cores %.1f' % (d['config']['frame_pairs_per_step'], $1, $2, d['value'], d['ms_per_step'], st['dense_flow'], st['tails'], d['host_cores_busy']))"
done
