#!/bin/bash
# is k_sor_wave bound by memory or by its instruction stream?  Same loads and stores with 10 / 5 / 1 of the ten half-sweeps computed (results wrong for < 10).
set -e
R=$PWD; O=$R/gpurun_out/wave_skip; mkdir -p $O
for sk in ${SKIPS:-5 1}; do
  cd $R/sindslam_amd/csrc && touch flow_wave.hip && make EXTRA_FILE="-fno-slp-vectorize -DSWV_SKIP=$sk ${XDEF}" build/flow_wave.o > /dev/null 2>&1 && make > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  WAVE=${WAVE:-3} timeout -k 10 300 rocprofv3 --kernel-trace -d $O/w -o t -- python3 $R/profiles/tools/flow_slices_alone.py 1 512 1 > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
  echo "half-sweeps computed: $sk"; python3 $R/profiles/tools/db_kernel_stats.py $(find $O/w -name "*.db" | head -1) 2 2 | tail -2
  python3 $R/profiles/tools/db_kernel_by_grid.py $(find $O/w -name "*.db" | head -1) k_sor_wave | head -4
  rm -rf $O/w
done
