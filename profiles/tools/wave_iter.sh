#!/bin/bash
# one iteration on k_sor_wave: parity tests, kernel table of the pyramid alone, HBM counters  (WAVE_ONLY=1: skip the k_sor_stream side)
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_flow_gpu.py -x -q -k "wave" > gpurun_out/wave_tests.txt 2>&1 || { tail -30 gpurun_out/wave_tests.txt; exit 1; }
tail -1 gpurun_out/wave_tests.txt
R=$PWD; O=$R/gpurun_out/wave_iter; mkdir -p $O; rm -f $O/table.txt
cd /tmp && export TMPDIR=/tmp
for wv in ${WAVES:-1}; do
  WAVE=$wv timeout -k 10 300 rocprofv3 --kernel-trace -d $O/w$wv -o t -- python3 $R/profiles/tools/flow_slices_alone.py ${SL:-1} ${PB:-512} 2 > $O/w$wv.log 2>&1 || { tail -5 $O/w$wv.log; exit 1; }
  tail -1 $O/w$wv.log
  python3 $R/profiles/tools/db_kernel_stats.py $(find $O/w$wv -name "*.db" | head -1) 3 4
  rm -rf $O/w$wv
  for c in FETCH_SIZE WRITE_SIZE; do
    WAVE=$wv timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p -- python3 $R/profiles/tools/flow_slices_alone.py ${SL:-1} ${PB:-512} 1 > $O/p_$c.log 2>&1 || { tail -5 $O/p_$c.log; exit 1; }
    python3 $R/profiles/tools/pmc_table.py k_sor_ $O/p | grep -E "wg=64|wg=512" ; rm -rf $O/p
  done
done
