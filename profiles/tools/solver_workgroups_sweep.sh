cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_flow_gpu.py -x -q > gpurun_out/wg_tests.log 2>&1 || { tail -20 gpurun_out/wg_tests.log; exit 1; }
tail -1 gpurun_out/wg_tests.log
rm -f gpurun_out/wg_sweep.txt
for cap in 0 170 160 128; do SOLVER_WGS=$cap python3 profiles/tools/flow_slices_alone.py 3 171 3 2>/dev/null | tail -1 >> gpurun_out/wg_sweep.txt; done
for cap in 0 170 160 144 128 102; do
python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sequence-leg --solver-workgroups $cap > gpurun_out/wg_$cap.json 2> gpurun_out/wg_$cap.err; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/wg_$cap.json') if l.startswith('{')][0]); s=d['stage_ms_per_step']
print('cap $cap', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'flow', round(s['dense_flow'],1), 'tails', round(s['tails'],1), 'wait', round(s['tails_wait_after_phase_a'],1), 'cores', round(d['host_cores_busy'],1), 'solver busy', round(d['roofline']['solver_busy_ms_per_step'],1), 'grow q', d['region_grow_gpu_quarters'])
" >> gpurun_out/wg_sweep.txt; done
cat gpurun_out/wg_sweep.txt
