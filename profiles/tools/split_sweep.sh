cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sequence-leg $EXTRA > gpurun_out/ss_$tag.json 2> gpurun_out/ss_$tag.err; python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ss_$tag.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']
print('$tag', '$*', '$EXTRA', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'cores', round(d['host_cores_busy'],1))
" >> gpurun_out/split_sweep.txt; }
rm -f gpurun_out/split_sweep.txt
EXTRA=""; run s3 SIND_FLOW_SPLIT=3; run s4 SIND_FLOW_SPLIT=4; run s2 SIND_FLOW_SPLIT=2; run s3b SIND_FLOW_SPLIT=3
EXTRA="--streams 28 --frames-per-step 8"; run q3 SIND_FLOW_SPLIT=3; run q2 SIND_FLOW_SPLIT=2; run q4 SIND_FLOW_SPLIT=4; run q1 SIND_FLOW_SPLIT=1
cat gpurun_out/split_sweep.txt
