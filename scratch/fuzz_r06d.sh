#!/bin/bash
# the final library of round 6 (descent loop over the table's 16-bit words): the benchmarked form once more against the oracle -> gpurun_out/$1/fuzz.txt
out=gpurun_out/$1; mkdir -p $out
export AGZ_PERSIST=1 AGZ_AGE_CLASS=block
{
echo "## final library (descent loop over the table's words): AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=4 FUZZ_SET=3 (thousands of games, cheap searches; sparse waves), refilled"
FUZZ_SLOT_DIV=4 FUZZ_SET=3 FUZZ_SEED_OFFSET=1100 timeout 330 python scratch/fuzz_generation.py
echo "## final library: AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=3, refilled, default set"
FUZZ_SLOT_DIV=3 FUZZ_SEED_OFFSET=1000 timeout 1000 python scratch/fuzz_generation.py
} > $out/fuzz.txt 2>&1
grep -c IDENTICAL $out/fuzz.txt; grep -v "IDENTICAL\|^   " $out/fuzz.txt | tail -6
