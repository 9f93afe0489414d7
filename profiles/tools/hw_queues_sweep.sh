cd $GRAFT_REPO_ROOT
for q in 8 6 10 8 4; do GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequence-leg 2>/dev/null > gpurun_out/hq.json; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/hq.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']; print('queues', $q, round(d['value'],1), round(d['ms_per_step'],1), round(s['dense_flow'],1), round(s['tails'],1))"; done
