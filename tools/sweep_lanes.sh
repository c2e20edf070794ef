#!/bin/bash
# lanes-per-frame sweep of the entropy stage by batch size, on one box (what huffman_sync_lanes' table is held against):
# usage: tools/sweep_lanes.sh <width> <height> "<frame counts>" "<lane counts>"  -> gpurun_out/lanes_sweep_<w>x<h>.txt
w=${1:-160}; h=${2:-120}
counts=${3:-"1250 2500 5000 10000 20000 30000 40000 50000 60000 80000"}
lanes_list=${4:-"default 1 2 4 8 16 32 64"}
out=gpurun_out/lanes_sweep_${w}x${h}.txt
: > $out
for n in $counts; do
  for lanes in $lanes_list; do
    if [ $lanes = default ]; then unset AMVHIP_SYNC_LANES; else export AMVHIP_SYNC_LANES=$lanes; fi
    r=$(python3 tools/time_kernels.py --width $w --height $h --frames $n --steps 6 2>/dev/null | tail -n 1)
    echo "$n $lanes $r" >> $out
  done
done
echo sweep done
