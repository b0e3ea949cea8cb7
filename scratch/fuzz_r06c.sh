#!/bin/bash
# exact (fp32) mode on the library of round 6: lock-step and refilled generations against the oracle's fp32 forward -> gpurun_out/$1/fuzz.txt
out=gpurun_out/$1; mkdir -p $out
export FUZZ_EXACT=1
{
echo "## FUZZ_EXACT=1 default dispatch (lock-step), default set"
FUZZ_SEED_OFFSET=700 timeout 700 python scratch/fuzz_generation.py
echo "## FUZZ_EXACT=1 AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=3, refilled, default set"
AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=3 FUZZ_SEED_OFFSET=800 timeout 700 python scratch/fuzz_generation.py
echo "## FUZZ_EXACT=1 AGZ_PERSIST=1 FUZZ_SLOT_DIV=4 FUZZ_SET=3 (thousands of games, cheap searches; sparse waves)"
AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=4 FUZZ_SET=3 FUZZ_SEED_OFFSET=900 timeout 500 python scratch/fuzz_generation.py
} > $out/fuzz.txt 2>&1
grep -c IDENTICAL $out/fuzz.txt; grep -v "IDENTICAL\|^   " $out/fuzz.txt | tail -8
