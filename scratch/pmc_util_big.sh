#!/bin/bash
# which unit is busy during the first-ply search of BASELINE config 3 (Gobang 9x9, 512x8, 32768 games: k_search_big, two 64-game workgroups per CU)?
# one small counter set per pass -> gpurun_out/$1/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  CFG=3 timeout 200 rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o x --output-format csv -- python3 scratch/pmc_point.py > $out/p$i.log 2>&1
  echo "# $c" >> $out/summary.txt
  python scratch/pmc_summary2.py $out/p$i 2>&1 | grep -v "k_advance\|k_scan\|k_compact\|k_fold\|Traceback\|File\|for r in\|IndexError" >> $out/summary.txt
  grep -h "error code\|exceeds" $out/p$i.log | head -1 >> $out/summary.txt
  rm -rf $out/p$i
done
cat $out/summary.txt
