#!/bin/bash
# usage: tools/build_variant.sh <translation unit without .hip> <tag> [-D flags]  ->  build/<unit>_<tag>.so
# Diagnostic / experiment builds of ONE translation unit of csrc/ (the others come from build/obj/*.o, compiled once with the Makefile's
# flags): e.g.  tools/build_variant.sh dp_conv_pw stamps -DDP_PWS_EXP=16 ;  then  DP_HIP_LIB=build/dp_conv_pw_stamps.so python tools/pws_micro.py
set -e
cd "$(dirname "$0")/../densepose_torchscript_amd/csrc"
B=../../build
mkdir -p $B/obj
unit=$1; tag=$2; shift 2
ALL="dp_conv dp_conv_ws dp_conv_wq dp_conv_rows dp_conv_pw dp_bottleneck dp_pair dp_pair256 dp_stem dp_ops dp_detect dp_extra"
objs=""
for f in $ALL; do
  if [ $f != $unit ]; then
    [ $B/obj/$f.o -nt $f.hip ] && [ $B/obj/$f.o -nt dp_common.h ] && [ $B/obj/$f.o -nt dp_policy.h ] && [ $B/obj/$f.o -nt ../../include/densepose_hip.h ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c -o $B/obj/$f.o $f.hip
    objs="$objs $B/obj/$f.o"
  fi
done
for f in dp_pack dp_policy; do
  [ $B/obj/$f.o -nt $f.cpp ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -c -o $B/obj/$f.o $f.cpp
  objs="$objs $B/obj/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c -o $B/obj/${unit}_$tag.o $unit.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $B/${unit}_$tag.so $B/obj/${unit}_$tag.o $objs
python3 ../lib.py stamp $(realpath $B/${unit}_$tag.so)
echo built build/${unit}_$tag.so
