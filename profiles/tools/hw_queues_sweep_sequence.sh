cd $GRAFT_REPO_ROOT
for q in 8 6 8 6; do GPU_MAX_HW_QUEUES=$q python3 bench.py --workload sequence --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg 2>/dev/null > gpurun_out/hq.json; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/hq.json') if l.startswith('{')][0]); s=d['sequence']; print('sequence job, queues', $q, round(d['value'],1), s['exact'], round(s['verify']['lockstep_seconds'],2), round(s['verify']['repair_seconds'],2))"; done
