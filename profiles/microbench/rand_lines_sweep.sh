#!/bin/bash
# (the binary: mkdir -p profiles/microbench/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rand_lines.hip -o profiles/microbench/_bin/rand_lines)
[ -x profiles/microbench/_bin/rand_lines ] || { mkdir -p profiles/microbench/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rand_lines.hip -o profiles/microbench/_bin/rand_lines; }
# Runs on the GPU box: the random-line rate over a 222 GiB span (configs[3]'s map) and over 23 GiB (the 10^9-key map), whole-span against slices.
# usage: bash profiles/microbench/rand_lines_sweep.sh > gpurun_out/r04/rand_lines.jsonl
B=profiles/microbench/_bin/rand_lines
for SPAN in 222 23; do
  for SL in 0 16384 4096 1024 256 64; do
    timeout -s KILL 120 $B $SPAN 4000 $SL 4 4 16
  done
  for WPS in 2 8; do timeout -s KILL 120 $B $SPAN 4000 0 $WPS 4 16; done
  for U in 1 2 8 16; do timeout -s KILL 120 $B $SPAN 4000 0 4 $U 16; done
  timeout -s KILL 120 $B $SPAN 4000 0 8 8 16
  timeout -s KILL 120 $B $SPAN 4000 1024 8 8 16
done
