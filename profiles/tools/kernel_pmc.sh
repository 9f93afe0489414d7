#!/bin/bash
# SQ counters of ONE kernel of the flow pyramid alone (sor_only.py, 170 pairs of 384 x 288), one --pmc pass per counter set (--kernel-trace only next to them):
#   bash profiles/tools/kernel_pmc.sh <kernel-substring> <out.txt>
set -o pipefail
k=${1:-k_coef}; out=${2:-gpurun_out/kernel_pmc.txt}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/kpmc; mkdir -p $O; rm -f $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/sor_only.py 170 1 384 288 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set" >> $R/$out; tail -3 $O/pmc$i.log >> $R/$out; }
  python3 $R/profiles/tools/pmc_table.py $k $O/pmc$i >> $R/$out 2>&1; rm -rf $O/pmc$i
done
echo done
