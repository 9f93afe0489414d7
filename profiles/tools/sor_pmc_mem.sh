#!/bin/bash
# memory-side PMC passes over the flow solver alone: bash profiles/tools/sor_pmc_mem.sh <tag> [env assignments...]   (e.g. SIND_SOR_DRY=1)
tag=${1:-sorpmcmem}; shift; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$tag; mkdir -p $O
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
i=0; rm -f $O/pmc_table.txt
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/sor_only.py 64 1 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 $O/pmc$i.log; }
  python3 $R/profiles/tools/pmc_table.py k_sor_fused $O/pmc$i >> $O/pmc_table.txt 2>&1; rm -rf $O/pmc$i
done
grep "wg=512" $O/pmc_table.txt | awk '{printf "%-44s %14s\n", $4, $8}'
