#!/bin/bash
# Diagnostic: VI pass time of a few tile-kernel shapes for two paddings of the l tile (MIMO_LS_PAD)
for pad in 2 1 3; do
  echo "LS_PAD $pad"
  for sh in "12 64" "16 32" "16 48" "8 256" "9 128" "8 96" "16 16" "13 64" "10 33"; do
    MIMO_LS_PAD=$pad python tools/quick_time.py 4e6 $sh 2>&1 | grep -v amdgpu.ids
  done
done
