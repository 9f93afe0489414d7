#!/bin/bash
# per-level launch durations of the solver kernel, pyramid alone
set -e
R=$PWD; O=$R/gpurun_out/wave_levels; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wv in ${WAVES:-4 0}; do
  WAVE=$wv WAVE_ITEMS=${WAVE_ITEMS:-0} timeout -k 10 300 rocprofv3 --kernel-trace -d $O/w$wv -o t -- python3 $R/profiles/tools/flow_slices_alone.py ${SL:-1} ${PB:-512} 1 > $O/w$wv.log 2>&1 || { tail -5 $O/w$wv.log; exit 1; }
  tail -1 $O/w$wv.log
  python3 $R/profiles/tools/db_kernel_by_grid.py $(find $O/w$wv -name "*.db" | head -1) k_sor_ | tee $O/levels_w$wv.txt
  rm -rf $O/w$wv
done
