#!/bin/bash
# same-box A/B of the 32-pixel row kernel for the DensePose head (DP_CONV_ROWS2=0: the head on the LDS-ring kernel); usage: ab_rows2.sh [bench flags]
B="python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps ${STEPS:-60} --warmup 10 $*"
for i in 1 2 3; do
  echo -n "rows2 on : "; $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo -n "rows2 off: "; DP_CONV_ROWS2=0 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
