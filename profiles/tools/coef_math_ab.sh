cd $GRAFT_REPO_ROOT
for m in 1 2 1 2 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequence-leg --coef-kernel $m 2>/dev/null > gpurun_out/cab.json; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/cab.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']; print('coef kernel', $m, round(d['value'],1), round(d['ms_per_step'],1), round(s['dense_flow'],1), round(s['tails'],1), round(d['roofline']['solver_busy_ms_per_step'],1))"; done
