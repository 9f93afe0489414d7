#!/bin/bash
# HBM-side counters of the solver kernel alone (one slice of PB pairs through the flow pyramid): FETCH_SIZE and WRITE_SIZE in separate passes, for k_sor_stream (WAVE=0) and k_sor_wave
set -e
R=$PWD; O=$R/gpurun_out/wave_pmc; mkdir -p $O; rm -f $O/table.txt
cd /tmp && export TMPDIR=/tmp
for wv in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    WAVE=$wv WAVE_ITEMS=${WAVE_ITEMS:-2048} timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p -- python3 $R/profiles/tools/flow_slices_alone.py 1 ${PB:-512} 1 > $O/p_$wv_$c.log 2>&1 || { tail -5 $O/p_$wv_$c.log; exit 1; }
    python3 $R/profiles/tools/pmc_table.py k_sor_ $O/p >> $O/table.txt; python3 $R/profiles/tools/pmc_table.py k_coef_lanes $O/p >> $O/table.txt; rm -rf $O/p
  done
done
cat $O/table.txt
