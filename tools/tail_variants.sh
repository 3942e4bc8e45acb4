#!/bin/bash
# the fused res2 tail (bottleneck_strip64_kernel) under the experiment builds of tools/build_variant.sh dp_bottleneck <tag> ...: same box, two rounds
# usage: tools/tail_variants.sh "tag1 tag2 ..."     ("" = the product library)
for rep in 1 2; do
  for tag in default $1; do
    lib=""; [ $tag != default ] && lib=build/dp_bottleneck_$tag.so
    for nxt in 1 0; do
      DP_SKIP_STAMP_CHECK=1 DP_HIP_LIB=$lib python3 tools/tail_micro.py 8 200 336 40 $nxt 2>&1 | grep "^tail"
    done
  done
done
