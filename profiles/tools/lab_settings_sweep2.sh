cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-sequence-leg > gpurun_out/ls.json 2>/dev/null; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ls.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']
print('$tag', '$*', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'cores', round(d['host_cores_busy'],1), 'grow q', d['region_grow_gpu_quarters'])" >> gpurun_out/lab_settings_sweep2.txt; }
rm -f gpurun_out/lab_settings_sweep2.txt
run occ128 SIND_OCC_CHUNK=128
run occ256 SIND_OCC_CHUNK=256
run occ64 SIND_OCC_CHUNK=64
run occ512 SIND_OCC_CHUNK=512
run occ192 SIND_OCC_CHUNK=192
run occ128b SIND_OCC_CHUNK=128
cat gpurun_out/lab_settings_sweep2.txt
