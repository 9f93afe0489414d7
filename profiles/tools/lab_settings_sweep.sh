# lab build (make lab): the remaining settings at the round's final balance, one call; bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-sequence-leg
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-sequence-leg > gpurun_out/ls.json 2>/dev/null; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ls.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']
print('$tag', '$*', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'cores', round(d['host_cores_busy'],1))" >> gpurun_out/lab_settings_sweep.txt; }
rm -f gpurun_out/lab_settings_sweep.txt
run base A=1
run occ32 SIND_OCC_CHUNK=32
run occ128 SIND_OCC_CHUNK=128
run ahead2 SIND_LAUNCH_AHEAD=2
run ahead5 SIND_LAUNCH_AHEAD=5
run occw10 SIND_OCC_WORKERS=10
run base2 A=1
cat gpurun_out/lab_settings_sweep.txt
