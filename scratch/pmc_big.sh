#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o x --output-format csv -- python3 scratch/pmc_point_big.py > $out/p$i.log 2>&1
  echo "# rocprofv3 --pmc $c --kernel-trace -- python3 scratch/pmc_point_big.py" >> $out/summary.txt
  python scratch/pmc_summary2.py $out/p$i | grep -v "k_advance\|k_scan\|k_compact\|k_fold" >> $out/summary.txt
done
cat $out/summary.txt
