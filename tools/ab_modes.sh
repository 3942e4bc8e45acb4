#!/bin/bash
# images/s of the headline workload under env-variable A/B settings and schedules: tools/ab_modes.sh "VAR=a VAR=b" ["bench flags" ...]
vars=$1; shift
for flags in "" "$@"; do
  for v in $vars; do
    echo -n "$v  [$flags]  "
    env $v python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps ${STEPS:-40} --warmup 10 $flags 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
