#!/bin/bash
# parity subset, then same-box A/B of the library in the tree against OLD=<library before a change> -> gpurun_out/$1/
out=gpurun_out/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_persistent.py -x -q -m gpu > $out/tests.txt 2>&1; tail -2 $out/tests.txt
{
CFG=0 bash scratch/ab_env_cfg.sh "AGZ_LIB_PATH=$OLD" "AGZ_X=0" "AGZ_LIB_PATH=$OLD" "AGZ_X=0" "AGZ_LIB_PATH=$OLD" "AGZ_X=0"
CFG=2 STEPS=10 bash scratch/ab_env_cfg.sh "AGZ_LIB_PATH=$OLD" "AGZ_X=0" "AGZ_LIB_PATH=$OLD" "AGZ_X=0"
CFG=3 STEPS=4 bash scratch/ab_env_cfg.sh "AGZ_LIB_PATH=$OLD" "AGZ_X=0"
} > $out/ab.txt 2>&1
cat $out/ab.txt
