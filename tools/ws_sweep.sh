#!/bin/bash
# A/B of the weight-stationary 3x3 kernels against the ring kernels on the layer shapes they serve (GPU box, repo root)
for shape in "8 128 100 168 128 3" "8 256 50 84 256 3" "8 256 100 168 256 3" "8 256 200 336 256 3" "8 256 25 42 256 3" "1 256 200 336 256 3" "1 256 50 84 256 3"; do
  for m in 0 1; do
    echo -n "DP_CONV_WS=$m  "
    DP_CONV_WS=$m python3 tools/conv_micro.py $shape 20 2>&1 | grep conv
  done
done
