#!/bin/bash
# which unit is busy in the PERSISTENT self-play kernels?  One small counter set per pass over ONE refilled call (GENS x 32768 games on 32768 slots:
# scratch/pmc_refill.py) of the headline config (CFG=0: k_selfplay_small) and of BASELINE config 3 (CFG=3: k_selfplay_big) -> gpurun_out/$1/util_cfg*.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-0 3}; do
  i=0; sum=$out/util_cfg$cfg.txt
  echo "# rocprofv3 --pmc <set> --kernel-trace -- python3 scratch/pmc_refill.py  (CFG=$cfg GENS=1: one agz_selfplay call of 32768 games on 32768 slots + the refilled tail; counters summed over the launch, per kernel)" > $sum
  for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    CFG=$cfg GENS=2 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $out/u$i -o x --output-format csv -- python3 scratch/pmc_refill.py > $out/u$i.log 2>&1
    echo "# $c" >> $sum
    python scratch/pmc_summary2.py $out/u$i 2>&1 | grep -E "k_selfplay|k_search" >> $sum
    grep -h "error code\|exceeds" $out/u$i.log | head -1 >> $sum
    rm -rf $out/u$i $out/u$i.log
  done
done
tail -12 $out/util_cfg*.txt
