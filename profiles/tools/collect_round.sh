#!/bin/bash
# Collects the per-round evidence under gpurun_out/<tag>/ on the GPU box (copy what is to be judged into profiles/rNN/ afterwards):
#   bash profiles/tools/collect_round.sh <tag>
# kernel trace + stats of the default bench command, the solver's per-level table, the tail kernels' duration table, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only next to them), and the plain bench lines of every config.
set -o pipefail
tag=${1:-round}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python3 $R/profiles/sor_by_grid.py $O/prof > $O/sor_by_level.txt; python3 $R/profiles/kernel_duration_dist.py $O/prof > $O/tail_kernel_durations.txt; python3 $R/profiles/tail_gpu_busy.py $O/prof 4 > $O/tail_gpu_busy.txt
rm -rf $O/prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
  python3 $R/profiles/pmc_sum.py k_sor_fused $O/pmc_$c > $O/pmc_$c.txt; rm -rf $O/pmc_$c
done
cd $R
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --sync --no-cpu-baseline > $O/bench_sync.json 2> $O/bench_sync.err || exit 1
timeout -k 10 300 python3 bench.py --config bonn > $O/bench_bonn.json 2> $O/bench_bonn.err || exit 1
timeout -k 10 400 python3 bench.py --config d455_720p > $O/bench_d455_720p.json 2> $O/bench_d455_720p.err || exit 1
timeout -k 10 300 python3 bench.py --workload sequence --no-cpu-baseline > $O/bench_sequence_1gpu.json 2> $O/bench_sequence_1gpu.err || exit 1
timeout -k 10 600 python3 bench.py --workload sequence --steps 20 --warmup 5 --no-cpu-baseline --collective-at-1 > $O/bench_sequence_driver_args.json 2> $O/bench_sequence_driver_args.err || exit 1
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_driver_args.json 2> $O/bench_default_driver_args.err || exit 1
SIND_TAIL_TIMING=1 timeout -k 10 300 python3 profiles/tools/exact_mode_timing.py 641 32 > $O/exact_mode_timing.txt 2>&1 || exit 1
echo collected
