#!/bin/bash
# diagnostic builds of the row kernels (tools/build_rows_variant.sh <n>), one layer each: timing only
for v in "$@"; do echo "== DP_ROWS_EXP=$v"; MODES=${MODES:-rows32} DP_HIP_LIB=build/rows_exp_$v.so timeout -k 10 100 python tools/rows_micro.py 64 2>&1 | grep median; done
