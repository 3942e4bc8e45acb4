mkdir -p gpurun_out
export LAYERS="res4 conv1,res5 conv3,res5 conv1"
DP_PWS_SKEW=0 DP_SKIP_STAMP_CHECK=1 DP_HIP_LIB=build/dp_conv_pw_xdma.so timeout -k 10 200 python tools/pws_micro.py 8 > gpurun_out/pws_stamps.log 2>&1
grep "pws<" gpurun_out/pws_stamps.log | grep "wave [04]" | grep lockstep
unset LAYERS
DP_SKIP_STAMP_CHECK=1 DP_HIP_LIB=build/dp_conv_pw_xdma_plain.so timeout -k 10 400 python tools/pws_micro.py 8 > gpurun_out/pws_xdma.log 2>&1; grep " pws-lock" gpurun_out/pws_xdma.log
