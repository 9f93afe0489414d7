#!/bin/bash
# SQ counters of k_sor_wave (or k_sor_stream with WAVE=0) on the pyramid alone
set -e
R=$PWD; O=$R/gpurun_out/wave_sq; mkdir -p $O; rm -f $O/table.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  i=$((i+1))
  WAVE=${WAVE:-1} timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/flow_slices_alone.py 1 ${PB:-512} 1 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 $O/pmc$i.log; }
  python3 $R/profiles/tools/pmc_table.py k_sor_ $O/pmc$i 2>&1 | grep -E "wg=64 |wg=512 " >> $O/table.txt || true; rm -rf $O/pmc$i
done
cat $O/table.txt
