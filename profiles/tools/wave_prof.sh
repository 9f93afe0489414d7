#!/bin/bash
# kernel table of the flow pyramid alone (one slice of 512 pairs) with the streaming kernel and with the one-wave pipelines
set -e
R=$PWD; O=$R/gpurun_out/wave_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wv in 0 1; do
  WAVE=$wv WAVE_ITEMS=${WAVE_ITEMS:-2048} timeout -k 10 300 rocprofv3 --kernel-trace -d $O/w$wv -o t -- python3 $R/profiles/tools/flow_slices_alone.py ${SL:-1} ${PB:-512} 2 > $O/w$wv.log 2>&1 || { tail -5 $O/w$wv.log; exit 1; }
  tail -1 $O/w$wv.log
  python3 $R/profiles/tools/db_kernel_stats.py $(find $O/w$wv -name "*.db" | head -1) 3 14 | tee $O/w$wv_table.txt > $O/table_w$wv.txt
  cat $O/table_w$wv.txt
  rm -rf $O/w$wv
done
