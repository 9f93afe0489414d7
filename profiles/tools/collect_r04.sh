#!/bin/bash
# Round-4 evidence under gpurun_out/<tag>/ on the GPU box (copy what is to be judged into profiles/r04/ afterwards):
#   bash profiles/tools/collect_r04.sh <tag> [part]
# part 1 : kernel trace + stats of the default bench command, per-kernel table, the solver's per-level table, GPU idle time
# part 1b: the two HBM PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only next to them), SQ counters of the solver alone
# part 2 : the plain bench lines: default with the driver's arguments (streams headline + sequence leg), sync, bonn, d455_720p, the sequence job stand-alone
#          (world size 1 with the RCCL calls, torch and C ABI), the sequence job with other warm-ups, the in-order mode's own timing
set -o pipefail
tag=${1:-r04}; part=${2:-all}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$tag; mkdir -p $O
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-6}
cd /tmp && export TMPDIR=/tmp
if [ "$part" = all ] || [ "$part" = 1 ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequence-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
  cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
  python3 $R/profiles/tools/kernel_table.py $O/kernel_stats.csv 4 512 > $O/kernel_table.txt
  python3 $R/profiles/sor_by_grid.py $O/prof k_sor_ > $O/sor_by_level.txt; python3 $R/profiles/kernel_duration_dist.py $O/prof > $O/tail_kernel_durations.txt; python3 $R/profiles/tail_gpu_busy.py $O/prof 4 > $O/tail_gpu_busy.txt
  python3 $R/profiles/tools/gpu_idle.py $O/prof 0.4 > $O/gpu_idle.txt
  rm -rf $O/prof
fi
if [ "$part" = all ] || [ "$part" = 1b ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sequence-leg > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
    python3 $R/profiles/pmc_sum.py k_sor_stream $O/pmc_$c > $O/pmc_$c.txt; python3 $R/profiles/pmc_sum.py k_sor_fused $O/pmc_$c >> $O/pmc_$c.txt; python3 $R/profiles/pmc_sum.py k_coef $O/pmc_$c >> $O/pmc_$c.txt; python3 $R/profiles/pmc_sum.py k_peac_grow $O/pmc_$c >> $O/pmc_$c.txt; rm -rf $O/pmc_$c
  done
  rm -f $O/sq_counters_solver.txt; i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/sor_only.py 170 1 384 288 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 $O/pmc$i.log; }
    python3 $R/profiles/tools/pmc_table.py k_sor_stream $O/pmc$i >> $O/sq_counters_solver.txt 2>&1; rm -rf $O/pmc$i
  done
fi
cd $R
if [ "$part" = all ] || [ "$part" = 2 ]; then
  timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench_default_driver_args.json 2> $O/bench_default_driver_args.err || exit 1
  timeout -k 10 300 python3 bench.py --sync --no-cpu-baseline --no-sequence-leg > $O/bench_sync.json 2> $O/bench_sync.err || exit 1
  timeout -k 10 300 python3 bench.py --config bonn --no-sequence-leg > $O/bench_bonn.json 2> $O/bench_bonn.err || exit 1
  timeout -k 10 400 python3 bench.py --config d455_720p --no-sequence-leg > $O/bench_d455_720p.json 2> $O/bench_d455_720p.err || exit 1
  timeout -k 10 600 python3 bench.py --workload sequence --steps 20 --warmup 5 --no-cpu-baseline --collective-at-1 > $O/bench_sequence_driver_args.json 2> $O/bench_sequence_driver_args.err || exit 1
  timeout -k 10 600 python3 bench.py --workload sequence --steps 20 --warmup 5 --no-cpu-baseline --collective-at-1 --collective cabi --no-exact-leg > $O/bench_sequence_cabi.json 2> $O/bench_sequence_cabi.err || exit 1
  timeout -k 10 600 python3 bench.py --workload sequence --steps 20 --warmup 5 --no-cpu-baseline --sequence-frames 830 --exact-leg-frames 830 > $O/bench_sequence_830_frames.json 2> $O/bench_sequence_830_frames.err || exit 1
  SIND_TAIL_TIMING=1 timeout -k 10 300 python3 profiles/tools/exact_mode_timing.py 641 32 > $O/exact_mode_timing.txt 2>&1 || exit 1
fi
echo collected
