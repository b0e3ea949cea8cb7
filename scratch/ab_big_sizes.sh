#!/bin/bash
# first-ply search time of the 512x8 trunk by batch size: the library in the tree against scratch/libagz_old.so (the library before k_selfplay_big4 /
# the epilogue, head and planes changes of mlp_big_body)
for L in 2048 4096 8192 12288 16384 24576 32768; do
  a=$(NH=512 NT=8 python scratch/prof_search.py 64 $L 3 2>/dev/null | tail -1 | sed 's/.*tree \([0-9.]*\) ms.*/\1/')
  b=$(NH=512 NT=8 python scratch/prof_lib.py libagz_old.so 64 $L 3 2>/dev/null | tail -1 | sed 's/.*tree \([0-9.]*\) ms.*/\1/')
  echo "L=$L  now $a ms  before $b ms"
done
