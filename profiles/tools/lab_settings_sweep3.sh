cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-sequence-leg $EXTRA > gpurun_out/ls.json 2>/dev/null; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ls.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']
print('$tag', '$*', '$EXTRA', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'cores', round(d['host_cores_busy'],1), 'thr', d['cpu_quota']['throttled_periods'])" >> gpurun_out/lab_settings_sweep3.txt; }
rm -f gpurun_out/lab_settings_sweep3.txt
EXTRA=""
run base A=1
run occw16 SIND_OCC_WORKERS=16
run occw20 SIND_OCC_WORKERS=20
run w40 SIND_WORKERS=40
run w48 SIND_WORKERS=48
run base2 A=1
EXTRA="--config d455_720p"
run base720 A=1
run occ192_720 SIND_OCC_CHUNK=192
run occ96_720 SIND_OCC_CHUNK=96
run w48_720 SIND_WORKERS=48
cat gpurun_out/lab_settings_sweep3.txt
