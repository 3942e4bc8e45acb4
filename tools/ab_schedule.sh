#!/bin/bash
# schedule switches of the headline workload, alternating, N rounds: tools/ab_schedule.sh [rounds]
B="python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 --warmup 10"
run() { echo -n "$1 [$2] "; env $1 $B $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in $(seq 1 ${1:-3}); do
  run X=1 ""
  run X=1 "--no-overlap"
  run DP_DEC_LATE=0 ""
  run X=1 "--pipeline 1"
  run X=1 "--pipeline 3"
  run DP_FORK=1 ""
done
