#!/bin/bash
# same-box A/B of the whole path between the in-tree library and an alternative build (tools/build_variant.sh): three alternating pairs
# usage (GPU box, repo root): tools/ab_lib.sh build/<variant>.so [bench flags]
alt=$PWD/$1; shift
B="python3 $PWD/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 8 $*"
for i in 1 2 3; do
  echo -n "in-tree library: "; $B 2>/dev/null | tail -1
  echo -n "$(basename $alt): "; DP_HIP_LIB=$alt $B 2>/dev/null | tail -1
done
