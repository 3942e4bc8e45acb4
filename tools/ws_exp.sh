#!/bin/bash
# experiment builds (build/exp_N.so, -DDP_EXP=N) of the weight-stationary kernel on one layer + SQ counter passes of the real build
shape=${1:-"8 256 200 336 256 3"}
echo "real build:"; python3 tools/conv_micro.py $shape 20 2>&1 | grep conv
for e in 1 4 8 16; do echo -n "DP_EXP=$e "; DP_HIP_LIB=$PWD/build/exp_$e.so python3 tools/conv_micro.py $shape 20 2>&1 | grep conv; done
PMC_MAX=4 tools/pmc_passes.sh wsr wsr -- python3 $PWD/tools/conv_micro.py $shape 5
