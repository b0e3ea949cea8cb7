#!/bin/bash
# quick PMC passes of the first-ply search (32768 games x 64 rollouts): instruction mix, HBM traffic -> gpurun_out/$1/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o x --output-format csv -- python3 scratch/pmc_point.py > $out/p$i.log 2>&1
  echo "# $c" >> $out/summary.txt
  python scratch/pmc_summary2.py $out/p$i 2>&1 | grep -v "k_advance\|k_scan\|k_compact\|k_fold" >> $out/summary.txt
done
cat $out/summary.txt
