#!/bin/bash
# the two blocks of scratch/fuzz_r06.sh that its 50-minute call did not reach -> gpurun_out/$1/fuzz.txt
out=gpurun_out/$1; mkdir -p $out
export AGZ_PERSIST=1 AGZ_AGE_CLASS=block
{
echo "## AGZ_PERSIST=1 AGZ_AGE_CLASS=block FUZZ_SLOT_DIV=4 FUZZ_SET=3 (thousands of games, cheap searches), refilled: the case the first call did not reach"
FUZZ_SLOT_DIV=4 FUZZ_SET=3 FUZZ_SEED_OFFSET=500 timeout 900 python scratch/fuzz_generation.py reversi6
echo "## ... FUZZ_SET=2 (V = 128, 256-wide trunks, 13x13, Hex 11x11, wide Connect4), refilled"
FUZZ_SLOT_DIV=3 FUZZ_SET=2 FUZZ_SEED_OFFSET=600 timeout 1500 python scratch/fuzz_generation.py
unset AGZ_PERSIST AGZ_AGE_CLASS
echo "## duels (scratch/fuzz_duel.py), default dispatch"
timeout 900 python scratch/fuzz_duel.py
} > $out/fuzz.txt 2>&1
grep -c IDENTICAL $out/fuzz.txt; grep -v "IDENTICAL\|^   " $out/fuzz.txt | tail -8
