#!/bin/bash
# A/B on one box: res2.0's projection shortcut inside the fused bottleneck tail (DP_FUSE_SC_TAIL=1, default) against the shortcut tensor
# written by its own launch and read back as the residual (=0). usage (GPU box, repo root): tools/ab_sc_tail.sh [config]
root=$PWD
B="python3 $root/bench.py --no-cpu-baseline --no-extras --no-roofline --config ${1:-densepose_rcnn_R_50_FPN_s1x}"
for i in 1 2 3; do
  for m in 0 1; do echo -n "DP_FUSE_SC_TAIL=$m: "; DP_FUSE_SC_TAIL=$m $B --steps 60 --warmup 8 2>/dev/null | tail -1; done
done
