#!/bin/bash
# rounds whose next k-means waits only for the depth halves of the tails (default) against whole tails (flow_opts_off bit 4): small shapes and the headline, alternating
set -e
mkdir -p gpurun_out
run() { timeout -k 10 300 python3 bench.py --streams $1 --frames-per-step $2 --steps ${3:-16} --warmup 3 --flow-opts-off $4 --no-cpu-baseline --no-sequence-leg --no-dropin-leg --no-small-step-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); st=d['stage_ms_per_step']
print('%3d x %d  opts_off %2d: %7.1f pairs/s  %6.1f ms/step  dense flow %6.1f  tails %6.1f  host cores %.1f  iou %s' % ($1, $2, $4, d['value'], d['ms_per_step'], st['dense_flow'], st['tails'], d['host_cores_busy'], (d.get('parity') or {}).get('mask_iou_min')))"; }
for shape in "8 4" "5 6" "16 2" "32 2"; do set -- $shape; for o in 16 0 16 0; do run $1 $2 16 $o; done; done
for o in 16 0 16 0; do run 128 4 10 $o; done
