# diagnostic builds of dp_conv_wq.hip (tools/build_variant.sh dp_conv_wq <tag> -DDP_WSQ_EXP=...) through tools/wsq_micro.py on one box
export CHECK=0 SHAPES=${SHAPES:-200x336}
for v in ${VARIANTS:-stamps nofetch}; do
  echo "== $v"; DP_SKIP_STAMP_CHECK=1 DP_HIP_LIB=build/dp_conv_wq_$v.so timeout -k 10 200 python tools/wsq_micro.py 8 bf16 2>&1 | grep -v amdgpu.ids | cut -c1-400
done
echo "== product"; CHECK=${PCHECK:-0} SHAPES=200x336,100x168,50x84,25x42,13x21 timeout -k 10 300 python tools/wsq_micro.py 8 bf16 2>&1 | grep -v amdgpu.ids | cut -c1-300
