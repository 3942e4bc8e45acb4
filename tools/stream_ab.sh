for rep in 1 2; do
for lib in build/dp_conv_r5stream.so ""; do
  for shape in "8 256 200 336 256 1" "8 128 100 168 512 1" "8 512 100 168 128 1" "8 64 200 336 64 1"; do
    echo -n "${lib:-new} | "; DP_SKIP_STAMP_CHECK=1 DP_HIP_LIB=$lib python tools/conv_micro.py $shape 60 2>&1 | grep conv
    echo -n "${lib:-new} +res | "; DP_SKIP_STAMP_CHECK=1 DP_HIP_LIB=$lib python tools/conv_micro.py $shape 60 res 2>&1 | grep conv
  done
done; done
