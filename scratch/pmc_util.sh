#!/bin/bash
# which unit is busy during the first-ply search? one small counter set per pass -> gpurun_out/$1/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o x --output-format csv -- python3 scratch/pmc_point.py > $out/p$i.log 2>&1
  echo "# $c" >> $out/summary.txt
  python scratch/pmc_summary2.py $out/p$i 2>&1 | grep -v "k_advance\|k_scan\|k_compact\|k_fold\|Traceback\|File\|for r in\|IndexError" >> $out/summary.txt
  grep -h "error code\|exceeds" $out/p$i.log | head -1 >> $out/summary.txt
done
cat $out/summary.txt
