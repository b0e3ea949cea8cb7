#!/bin/bash
# extra PMC passes of the first-ply search (stall / instruction-cache / TLB counters) -> gpurun_out/$1/
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_STALL" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_THRASHING_STALL TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_TA_BUSY"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o x --output-format csv -- python3 scratch/pmc_point.py > $out/p$i.log 2>&1
  echo "# $c" >> $out/summary.txt
  python scratch/pmc_summary2.py $out/p$i | grep -v "k_advance\|k_scan\|k_compact\|k_fold" >> $out/summary.txt
done
cat $out/summary.txt
