#!/bin/bash
# Lanes a heavy frame gets in a mixed batch (AMVHIP_SPLIT), swept on one box: 160 000 frames of 160x120, every 16th white noise.
# Round 6: entropy stage 3.41 - 3.52 / 2.97 / 2.73 - 2.74 / 2.66 - 2.70 / 2.82 - 2.84 ms with 4 / 8 / 16 (the default) / 32 / 64.
out=gpurun_out/split_sweep.txt
: > $out
for sp in default 4 8 16 32 64; do
  if [ $sp = default ]; then unset AMVHIP_SPLIT; else export AMVHIP_SPLIT=$sp; fi
  for rep in 1 2; do
  r=$(python3 tools/time_kernels.py --frames 160000 --mixed 16 --steps 6 2>/dev/null | tail -n 1)
  echo "$sp $r" >> $out
  done
done
echo done
