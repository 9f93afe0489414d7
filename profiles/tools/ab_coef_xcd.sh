#!/bin/bash
# k_coef_lanes: the tiles of a pair on one XCD (COEF=1, default) against plain grid order (COEF=3); pyramid alone, kernel table + FETCH_SIZE
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_flow_gpu.py -x -q -k "coefficients or wave_solver_on" > gpurun_out/coef_tests.txt 2>&1 || { tail -30 gpurun_out/coef_tests.txt; exit 1; }
tail -1 gpurun_out/coef_tests.txt
R=$PWD; O=$R/gpurun_out/ab_coef; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for cf in 3 1; do
  COEF=$cf timeout -k 10 300 rocprofv3 --kernel-trace -d $O/w -o t -- python3 $R/profiles/tools/flow_slices_alone.py ${SL:-1} ${PB:-512} 2 > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
  echo "COEF=$cf: $(tail -1 $O/w.log)"; python3 $R/profiles/tools/db_kernel_stats.py $(find $O/w -name "*.db" | head -1) 3 3 | tail -3; rm -rf $O/w
done; done
for cf in 3 1; do
  COEF=$cf timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p -- python3 $R/profiles/tools/flow_slices_alone.py ${SL:-1} ${PB:-512} 1 > $O/p.log 2>&1 || { tail -5 $O/p.log; exit 1; }
  echo "COEF=$cf"; python3 $R/profiles/tools/pmc_table.py k_coef_lanes $O/p; rm -rf $O/p
done
