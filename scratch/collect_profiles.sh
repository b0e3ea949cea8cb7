#!/bin/bash
# Runs on the GPU box (gpurun): the round's measurement set -> gpurun_out/$1/ (copied into profiles/ afterwards by scratch/install_profiles.py)
#   bench lines of the headline config and of BASELINE configs 2-5, rocprofv3 kernel stats of the headline bench command,
#   PMC passes of the first-ply search (instruction mix, FETCH_SIZE, WRITE_SIZE, activity), per-ply search times by batch size
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  n=$(echo $c | cut -c1-12 | tr " " _)
  timeout 150 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$n -o x --output-format csv -- python3 scratch/pmc_point.py > $out/pmc_$n.log 2>&1
  echo "# rocprofv3 --pmc $c --kernel-trace -- python3 scratch/pmc_point.py" >> $out/pmc_summary.txt
  python scratch/pmc_summary2.py $out/pmc_$n 2>&1 | grep -v "k_advance\|k_scan\|k_compact\|k_fold" >> $out/pmc_summary.txt
  grep -h sum_p $out/pmc_$n.log | tail -1 >> $out/pmc_summary.txt
done
python bench.py --steps 3 --warmup 1 > $out/bench_headline.json 2> $out/bench_headline.err
for c in 2 3 4 5; do timeout 900 python bench.py --config $c --steps 2 --warmup 1 > $out/bench_config$c.json 2> $out/bench_config$c.err; done
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery > $out/bench_under_rocprof.json 2> $out/stats.log
# per-ply search time by batch size (default dispatch), 128x6 and 512x8
for L in 256 1024 2048 4096 8192 16384 24576 32768; do echo "128x6 L=$L $(python scratch/prof_search.py 64 $L 3 | tail -1)" >> $out/per_ply_by_batch.txt; done
for L in 256 1024 4096 8192 16384 32768; do echo "512x8 L=$L $(NH=512 NT=8 python scratch/prof_search.py 64 $L 3 | tail -1)" >> $out/per_ply_by_batch.txt; done
ls $out
