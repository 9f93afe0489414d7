#!/bin/bash
# Round-5 evidence under gpurun_out/<tag>/ on the GPU box (copy what is to be judged into profiles/r05/ afterwards):
#   bash profiles/tools/collect_r05.sh <tag> [part]
# part 1 : kernel trace + stats of the default bench command (csv), per-kernel table, GPU busy / concurrency
# part 1b: the two HBM PMC passes of the bench command (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only next to them), SQ counters of the streaming solver alone
# part 2 : plain bench lines: default with the driver's arguments (streams headline + dropin + small_step + sequence legs), bonn, d455_720p, the sequence job stand-alone
# part 3 : the regimes of round 5: DeepFlow alone at 1 / 8 / 32 pairs (kernel stats), the reference's call pattern (dropin), small steps by slice count
set -o pipefail
tag=${1:-r05}; part=${2:-all}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$part" = all ] || [ "$part" = 1 ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequence-leg --no-dropin-leg --no-small-step-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
  cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
  python3 $R/profiles/tools/kernel_table.py $O/kernel_stats.csv 4 512 > $O/kernel_table.txt
  python3 $R/profiles/tools/gpu_idle.py $O/prof 0.4 > $O/gpu_idle.txt
  rm -rf $O/prof
fi
if [ "$part" = all ] || [ "$part" = 1b ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sequence-leg --no-dropin-leg --no-small-step-leg > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
    for k in k_sor_wave k_sor_stream k_sor_fused k_sor_tile k_coarse_chain k_coef; do python3 $R/profiles/pmc_sum.py $k $O/pmc_$c; done > $O/pmc_$c.txt
    rm -rf $O/pmc_$c
  done
  rm -f $O/sq_counters_solver.txt; i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/sor_only.py 170 1 384 288 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 $O/pmc$i.log; }
    python3 $R/profiles/tools/pmc_table.py k_sor_wave $O/pmc$i >> $O/sq_counters_solver.txt 2>&1; rm -rf $O/pmc$i
  done
fi
cd $R
if [ "$part" = all ] || [ "$part" = 2 ]; then
  timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_default_driver_args.json 2> $O/bench_default_driver_args.err || exit 1
  timeout -k 10 300 python3 bench.py --config bonn --no-sequence-leg --no-dropin-leg --no-small-step-leg > $O/bench_bonn.json 2> $O/bench_bonn.err || exit 1
  timeout -k 10 400 python3 bench.py --config d455_720p --no-sequence-leg --no-small-step-leg > $O/bench_d455_720p.json 2> $O/bench_d455_720p.err || exit 1
  timeout -k 10 600 python3 bench.py --workload sequence --steps 20 --warmup 5 --no-cpu-baseline --collective-at-1 > $O/bench_sequence_driver_args.json 2> $O/bench_sequence_driver_args.err || exit 1
  timeout -k 10 600 python3 bench.py --workload sequence --steps 20 --warmup 5 --no-cpu-baseline --seq-driver python --no-exact-leg > $O/bench_sequence_python_driver.json 2> $O/bench_sequence_python_driver.err || exit 1
fi
if [ "$part" = all ] || [ "$part" = 3 ]; then
  python3 profiles/tools/dropin_latency.py 120 > $O/dropin_and_flow_alone.json 2> $O/dropin.err || exit 1
  python3 profiles/tools/dropin_latency.py 120 --batches 1 --no-overlap > $O/dropin_no_overlap.json 2>> $O/dropin.err || exit 1
  cd /tmp
  for b in 1 8 32; do
    timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_b$b -o b -- python3 $R/profiles/tools/dropin_latency.py --flow-only --batches $b > $O/prof_b$b.log 2>&1 || exit 1
    python3 $R/profiles/tools/db_kernel_stats.py $O/prof_b$b/b_results.db 13 12 > $O/flow_alone_B$b.txt; rm -rf $O/prof_b$b
  done
  cd $R
  SLICES="1 2 3" bash profiles/tools/small_step_slices.sh "8 4" "16 2" "32 1" "32 2" > $O/small_step_slices.txt 2>&1
fi
echo collected
