#!/bin/bash
# PMC passes over the flow solver alone (profiles/tools/sor_only.py): bash profiles/tools/sor_pmc.sh <tag> [env assignments...]
tag=${1:-sorpmc}; shift; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$tag; mkdir -p $O
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
cd $R && timeout -k 10 200 python3 profiles/tools/sor_only.py 64 2 > $O/plain.txt 2>&1 || { cat $O/plain.txt; exit 1; }
cat $O/plain.txt
cd /tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/profiles/tools/sor_only.py 64 2 > $O/trace.txt 2>&1 || exit 1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv; python3 $R/profiles/sor_by_grid.py $O/trace > $O/sor_by_level.txt; rm -rf $O/trace
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/sor_only.py 64 1 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 $O/pmc$i.log; }
  python3 $R/profiles/tools/pmc_table.py k_sor_fused $O/pmc$i >> $O/pmc_table.txt 2>&1; rm -rf $O/pmc$i
done
cat $O/pmc_table.txt
