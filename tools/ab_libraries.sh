#!/bin/bash
# Runs ON THE GPU BOX: the same sweeps with two builds of libcorrfield (CORRFIELD_LIBRARY), interleaved.
# Usage: bash tools/ab_libraries.sh <before.so> <after.so>
A=$1; B=$2
for round in 1 2; do
  for lib in $A $B; do
    echo "== round $round  $(basename $lib)  one reference point, 256^3"
    CORRFIELD_LIBRARY=$lib python tools/sweep_members.py --measures spearman kendall mi_binned --members 32 48 64 96 128 2>&1 | grep -v amdgpu | tail -5
  done
done
for lib in $A $B; do
  echo "== $(basename $lib)  two-field symmetric mode, 256^3"
  CORRFIELD_LIBRARY=$lib python tools/sweep_members.py --symmetric --measures spearman kendall mi_binned --members 64 100 128 --iters 3 2>&1 | grep -v amdgpu | tail -3
  echo "== $(basename $lib)  sibling reductions"
  CORRFIELD_LIBRARY=$lib python tools/measure_stats.py 2>&1 | grep -v amdgpu | tail -8
done
