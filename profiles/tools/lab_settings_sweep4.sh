cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-sequence-leg > gpurun_out/ls.json 2>/dev/null; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ls.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']
print('$tag', '$*', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'cores', round(d['host_cores_busy'],1), 'thr', d['cpu_quota']['throttled_periods'])" >> gpurun_out/lab_settings_sweep4.txt; }
rm -f gpurun_out/lab_settings_sweep4.txt
for i in 1 2 3; do run occw14 SIND_OCC_WORKERS=14; run occw20 SIND_OCC_WORKERS=20; run occw24 SIND_OCC_WORKERS=24; done
cat gpurun_out/lab_settings_sweep4.txt
