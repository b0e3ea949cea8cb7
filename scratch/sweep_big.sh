#!/bin/bash
# first-ply search time of the 512x8 trunk at 32768 games for sub-batch chain counts and leaf tiles per network workgroup
for ch in 1 2 3 4; do for mt in 0 2 4 8; do
  if [ $mt = 0 ]; then unset AGZ_BIG_MT; else export AGZ_BIG_MT=$mt; fi
  echo "chains=$ch mt=$mt $(AGZ_CHAINS=$ch NH=512 NT=8 NOPROF=1 python scratch/prof_search.py ${VV:-64} ${LL:-32768} 3 | tail -1)"
done; done
