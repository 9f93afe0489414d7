#!/bin/bash
# k_sor_wave: rows in flight 1 / 2 / 3 (WAVE = 2 / 3 / 4), pyramid alone; kernel table per variant
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_flow_gpu.py -x -q -k "wave" > gpurun_out/wave_tests.txt 2>&1 || { tail -30 gpurun_out/wave_tests.txt; exit 1; }
tail -1 gpurun_out/wave_tests.txt
R=$PWD; O=$R/gpurun_out/wave_pf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wv in ${WAVES:-2 3 4}; do
  WAVE=$wv timeout -k 10 300 rocprofv3 --kernel-trace -d $O/w$wv -o t -- python3 $R/profiles/tools/flow_slices_alone.py ${SL:-1} ${PB:-512} 2 > $O/w$wv.log 2>&1 || { tail -5 $O/w$wv.log; exit 1; }
  tail -1 $O/w$wv.log
  python3 $R/profiles/tools/db_kernel_stats.py $(find $O/w$wv -name "*.db" | head -1) 3 2 | tail -2
  rm -rf $O/w$wv
done
