#!/bin/bash
# A/B timing of library builds on ONE box: tools/ab.sh libA.so libB.so ...   (each twice, interleaved)
for rep in 1 2; do
  for l in "$@"; do
    echo -n "$l: "
    WN_HIP_LIB=$PWD/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o 'ms_per_step[^,]*\|avg_launch_ms[^,]*' | head -3 | tr '\n' ' '
    echo
  done
done
