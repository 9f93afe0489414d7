#!/bin/bash
# Small steps (what a rank of a multi-GPU sequence job sees): frame pairs/s of the streams workload at 32 / 64 pairs per step with the step cut into 1, 2, 3, 4
# independent pipelines (bench.py --pipelines; sindslam_amd.pipeline.PipelineGroup).  One gpurun call:  bash profiles/tools/small_steps.sh > gpurun_out/small_steps.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for cfg in "8 4" "16 4" "6 5" "24 4"; do
  set -- $cfg
  for p in 1 2 3 4; do
    [ $p -gt $1 ] && continue
    line=$(timeout -k 10 300 python3 bench.py --streams $1 --frames-per-step $2 --steps 16 --warmup 3 --pipelines $p --no-cpu-baseline --no-sequence-leg 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('%3d pairs/step (%2d streams x %d), %d pipeline(s): %7.1f pairs/s  %6.1f ms/step  dense flow %6.1f  tails %6.1f  host cores %.1f' % (d['config']['frame_pairs_per_step'], $1, $2, $p, d['value'], d['ms_per_step'], st['dense_flow'], st['tails'], d['host_cores_busy']))"
  done
done
