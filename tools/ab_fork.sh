#!/bin/bash
# RPN levels on forked streams (EngineOptions.fork_levels = 2, DP_FORK) against in line (0): headline, other batch sizes and configurations, alternating
B="python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 --warmup 10"
for flags in "" "--batch 1" "--batch 2" "--batch 4" "--batch 16" "--config densepose_rcnn_R_101_FPN_s1x" "--config densepose_rcnn_R_50_FPN_DL_s1x" "--config densepose_rcnn_R_50_FPN_s1x_legacy" "--dtype fp16" "--config densepose_rcnn_R_101_FPN_DL_s1x --dtype fp16 --height 1080 --width 1920 --batch 16"; do
  for rep in 1 2; do
    for f in 2 0; do
      echo -n "DP_FORK=$f [$flags] "; DP_FORK=$f $B $flags 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
    done
  done
done
