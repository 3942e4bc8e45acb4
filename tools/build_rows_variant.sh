#!/bin/bash
# usage: tools/build_rows_variant.sh <DP_ROWS_EXP value> [more -D flags]  -> build/rows_exp_<value>.so (the other translation units come from build/obj/*.o,
# built once with the Makefile's flags; only dp_conv_rows.hip is recompiled). Diagnostic builds of the row kernels: timing only.
set -e
cd "$(dirname "$0")/../densepose_torchscript_amd/csrc"
B=../../build
mkdir -p $B/obj
for f in dp_conv dp_conv_ws dp_bottleneck dp_stem dp_ops dp_detect dp_extra; do
  [ $B/obj/$f.o -nt $f.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c -o $B/obj/$f.o $f.hip
done
[ $B/obj/dp_pack.o -nt dp_pack.cpp ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -c -o $B/obj/dp_pack.o dp_pack.cpp
v=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DDP_ROWS_EXP=$v "$@" -c -o $B/obj/rows_$v.o dp_conv_rows.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $B/rows_exp_$v.so $B/obj/rows_$v.o $B/obj/dp_conv.o $B/obj/dp_conv_ws.o $B/obj/dp_bottleneck.o $B/obj/dp_stem.o $B/obj/dp_ops.o $B/obj/dp_detect.o $B/obj/dp_extra.o $B/obj/dp_pack.o
python3 ../lib.py stamp $(realpath $B/rows_exp_$v.so)
