#!/bin/bash
# memory-side PMC passes of the first-ply search -> gpurun_out/$1/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_THRASHING_STALL TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_TA_BUSY" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o x --output-format csv -- python3 scratch/pmc_point.py > $out/p$i.log 2>&1
  echo "# $c" >> $out/summary.txt
  python scratch/pmc_summary2.py $out/p$i 2>&1 | grep -v "k_advance\|k_scan\|k_compact\|k_fold" >> $out/summary.txt
done
cat $out/summary.txt
