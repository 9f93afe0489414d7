#!/bin/bash
# SQ counters of the streaming solver with the device filled the way the bench fills it: the bench runs three flow slices of ~170 pairs side by side, i.e. 3 x 510 workgroups for
# the 512 workgroup slots of the chip on the three-strip levels; the solver ALONE on 170 pairs (collect_r04.sh part 1b) leaves slots empty on every level that cuts into one or two
# strips.  Here: one launch sequence over B pairs (default 512 = a whole step) so that every level has at least as many workgroups as slots.
#   bash profiles/tools/sor_pmc_filled.sh <out.txt> [B]
set -o pipefail
out=${1:-gpurun_out/sq_counters_solver_filled.txt}; B=${2:-512}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/spf; mkdir -p $O; rm -f $R/$out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/sor_only.py $B 1 384 288 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set" >> $R/$out; tail -3 $O/pmc$i.log >> $R/$out; }
  python3 $R/profiles/tools/pmc_table.py k_sor_stream $O/pmc$i >> $R/$out 2>&1; rm -rf $O/pmc$i
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/profiles/tools/sor_only.py $B 2 384 288 > $O/trace.log 2>&1 && { grep "ms per batch" $O/trace.log >> $R/$out; python3 $R/profiles/sor_by_grid.py $O/trace >> $R/$out; rm -rf $O/trace; }
echo done
