mkdir -p gpurun_out/r3h
timeout -k 10 600 python -m pytest tests/test_flow_gpu.py -x -q > gpurun_out/r3h/tests.log 2>&1; tail -2 gpurun_out/r3h/tests.log
for c in F5 4 7 10 14 20; do
  if [ $c = F5 ]; then e="SIND_SOR_FUSE=5"; else e="SIND_SOR_FUSE=0 SIND_SOR_PLAN_COST=$c"; fi
  env $e python bench.py --no-cpu-baseline --sync --steps 4 --warmup 2 > gpurun_out/r3h/b.json 2>> gpurun_out/r3h/err.txt
  python - "$c" <<'P'
import json,sys
l=[x for x in open('gpurun_out/r3h/b.json') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']
print('plan_cost', sys.argv[1], round(d['value'],1), 'solver_busy', round(r['solver_busy_ms_per_step'],1), 'launches', r['launches'], flush=True)
P
done
