#!/bin/bash
# round 6 fuzz set on the GPU box (final library: next words in LDS, block-local count of the sampled action, sparse 4-lane waves): whole generations
# against the oracle — default dispatch (lock-step: k_search_* per ply), persistent kernels (refilled, chains, age classes), other shapes -> gpurun_out/$1/fuzz.txt
out=gpurun_out/$1; mkdir -p $out
{
echo "## default dispatch (lock-step generations: one launch per ply), default set, bf16"
timeout 1500 python scratch/fuzz_generation.py
echo "## FUZZ_SET=2 (V = 128, 256-wide trunks, 13x13, Hex 11x11, wide Connect4), default dispatch"
FUZZ_SET=2 timeout 1500 python scratch/fuzz_generation.py
export AGZ_PERSIST=1 AGZ_AGE_CLASS=block
echo "## AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=3 (persistent kernels, a third of the games in flight, slots refilled), default set, other seeds"
FUZZ_SLOT_DIV=3 FUZZ_SEED_OFFSET=300 timeout 1500 python scratch/fuzz_generation.py
echo "## ... FUZZ_CHAIN=1 (chains of three calls), default set, other seeds"
FUZZ_SLOT_DIV=3 FUZZ_CHAIN=1 FUZZ_SEED_OFFSET=400 timeout 1500 python scratch/fuzz_generation.py
echo "## ... FUZZ_SET=3 (thousands of games, cheap searches), refilled"
FUZZ_SLOT_DIV=4 FUZZ_SET=3 FUZZ_SEED_OFFSET=500 timeout 900 python scratch/fuzz_generation.py
echo "## ... FUZZ_SET=2, refilled"
FUZZ_SLOT_DIV=3 FUZZ_SET=2 FUZZ_SEED_OFFSET=600 timeout 1500 python scratch/fuzz_generation.py
} > $out/fuzz.txt 2>&1
grep -c IDENTICAL $out/fuzz.txt; grep -v IDENTICAL $out/fuzz.txt | tail -12
