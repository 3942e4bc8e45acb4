#!/bin/bash
# images/s of the other BASELINE.json configurations and batch sizes on the current build (GPU box, repo root)
B="python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | tail -1; }
run
run --config densepose_rcnn_R_101_FPN_s1x
run --config densepose_rcnn_R_50_FPN_DL_s1x
run --config densepose_rcnn_R_101_FPN_DL_s1x --dtype fp16 --height 1080 --width 1920 --batch 16
run --config densepose_rcnn_R_50_FPN_s1x_legacy
run --dtype fp16
for b in 1 2 4 16 64; do run --batch $b; done
