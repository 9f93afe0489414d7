cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sequence-leg > gpurun_out/ps_$tag.json 2> gpurun_out/ps_$tag.err; python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ps_$tag.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']
print('$tag', '$*', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'wait', round(s['tails_wait_after_phase_a'],1), 'cores', round(d['host_cores_busy'],1))
" >> gpurun_out/prio_sweep.txt; }
rm -f gpurun_out/prio_sweep.txt
run base A=1
run w64 SIND_WORKERS=64
run tp0 SIND_TAIL_PRIORITY=0
run tp0w64 SIND_TAIL_PRIORITY=0 SIND_WORKERS=64
run tp0w96 SIND_TAIL_PRIORITY=0 SIND_WORKERS=96
run allp0w96 SIND_TAIL_PRIORITY=0 SIND_PHASEA_PRIORITY=0 SIND_OCC_PRIORITY=0 SIND_WORKERS=96
run base2 A=1
cat gpurun_out/prio_sweep.txt
