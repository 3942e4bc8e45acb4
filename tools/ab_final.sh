#!/bin/bash
# A/B of the decoder's fork point on one box + the two timelines (GPU box, repo root)
root=$PWD; out=$root/gpurun_out
B="python3 $root/bench.py --no-cpu-baseline --no-extras --no-roofline"
for i in 1 2 3; do
  for m in 0 1; do echo -n "DP_DEC_LATE=$m: "; DP_DEC_LATE=$m $B --steps 60 --warmup 8 2>/dev/null | tail -1; done
done
for m in 0 1; do echo -n "batch 1 DP_DEC_LATE=$m: "; DP_DEC_LATE=$m $B --steps 60 --warmup 8 --batch 1 2>/dev/null | tail -1; done
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export DP_DEC_LATE=$m
  rm -rf $out/tl_late$m
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tl_late$m -- $B --steps 6 --warmup 4 --pipeline 1 > $out/tl_late$m.log 2>&1
  python3 $root/tools/timeline.py $out/tl_late$m > $out/timeline_late$m.txt
done
