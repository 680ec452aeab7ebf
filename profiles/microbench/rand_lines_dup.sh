#!/bin/bash
# (the binary: mkdir -p profiles/microbench/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rand_lines.hip -o profiles/microbench/_bin/rand_lines)
[ -x profiles/microbench/_bin/rand_lines ] || { mkdir -p profiles/microbench/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rand_lines.hip -o profiles/microbench/_bin/rand_lines; }
# Runs on the GPU box: what asking for the same line from several instructions of a lane (dup) and from neighbouring lanes (group) costs.
# "gathers" count (lane, line) pairs: with group g the distinct lines are gathers / g.
B=profiles/microbench/_bin/rand_lines
for SPAN in 23 0.025; do
  for DG in "1 1" "2 1" "4 1" "1 2" "4 2" "1 8" "4 8"; do
    timeout -s KILL 120 $B $SPAN 2000 0 4 4 16 $DG
    timeout -s KILL 120 $B $SPAN 2000 0 4 8 16 $DG
  done
done
