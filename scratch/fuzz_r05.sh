#!/bin/bash
# round 5 fuzz set on the GPU box: whole generations against the oracle through the PERSISTENT self-play kernels (refilled slots, chains,
# age classes by workgroup parity), bf16 and exact; -> gpurun_out/$1/fuzz.txt
out=gpurun_out/$1; mkdir -p $out
export AGZ_PERSIST=1 AGZ_AGE_CLASS=block
{
echo "## AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=3 (persistent kernels, a third of the games in flight, slots refilled), default set, bf16"
FUZZ_SLOT_DIV=3 timeout 1500 python scratch/fuzz_generation.py
echo "## ... FUZZ_CHAIN=1 (chains of three calls), default set, other seeds"
FUZZ_SLOT_DIV=3 FUZZ_CHAIN=1 FUZZ_SEED_OFFSET=200 timeout 1500 python scratch/fuzz_generation.py
echo "## ... FUZZ_SET=3 (thousands of games, cheap searches), refilled"
FUZZ_SLOT_DIV=4 FUZZ_SET=3 timeout 900 python scratch/fuzz_generation.py
} > $out/fuzz.txt 2>&1
tail -3 $out/fuzz.txt
