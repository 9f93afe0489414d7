#!/bin/bash
# memory-side counters of k_sor_wave on the pyramid alone (one slice of PB pairs)
R=$PWD; O=$R/gpurun_out/wave_pmc_mem; mkdir -p $O; rm -f $O/table.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/profiles/tools/flow_slices_alone.py 1 ${PB:-512} 1 > $O/pmc$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 $O/pmc$i.log; }
  python3 $R/profiles/tools/pmc_table.py k_sor_wave $O/pmc$i 2>&1 >> $O/table.txt; rm -rf $O/pmc$i
done
cat $O/table.txt
