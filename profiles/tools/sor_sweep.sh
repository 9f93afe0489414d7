#!/bin/bash
# A/B timing of solver variants through the bench (sync mode, 4 steps): bash profiles/tools/sor_sweep.sh <tag> "<MODE FUSE TILEW TILEH>" ...
tag=$1; shift; mkdir -p gpurun_out/$tag
for cfg in "$@"; do
  set -- $cfg; t=m$1_f$2_$3x$4
  SIND_SOR_MODE=$1 SIND_SOR_FUSE=$2 SIND_SOR_TILEW=$3 SIND_SOR_TILEH=$4 timeout -k 10 300 python bench.py --no-cpu-baseline --sync --steps 4 --warmup 2 > gpurun_out/$tag/$t.json 2>> gpurun_out/$tag/bench.err || exit 1
  python - "$tag" "$t" <<'P'
import json,sys
tag,t=sys.argv[1:3]
l=[x for x in open(f'gpurun_out/{tag}/{t}.json') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
print(t, round(d['value'],1), 'solver_busy', round(r['solver_busy_ms_per_step'],1), 'avg_us', round(r['avg_launch_us'],1), flush=True)
P
done
