cd $GRAFT_REPO_ROOT
for s in 28 20 14 10; do
  python3 bench.py --workload sequence --steps 20 --warmup 5 --streams $s --no-cpu-baseline --no-exact-leg > gpurun_out/r4_seq_s$s.json 2> gpurun_out/r4_seq_s$s.err
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r4_seq_s$s.json").read().strip().splitlines()[-1]); s=d["sequence"]; v=s["verify"]
print("chunks %d T %d: value %.1f  lockstep %.2f s  repair %.2f s  mismatched %d/%d  max conv %d  dense %.1f tails %.1f per step" % (s["chunks"], s["frames_per_step_per_chunk"], s["value"], v["lockstep_seconds"], v["repair_seconds"], v["mismatched_seams"], v["seams"], v["max_frames_to_converge"], d["stage_ms_per_step"]["dense_flow"], d["stage_ms_per_step"]["tails"]))
PY
done
