mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "weight_stationary_pointwise or stream_kernel" > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log; tail -3 gpurun_out/t1.log
grep -q "rc=0" gpurun_out/t1.log || exit 1
LAYERS="res4 conv3,res4 conv1" timeout -k 10 300 python tools/pws_micro.py 8 2>&1 | grep "pws\|before"
for v in 1 0; do
  DP_CONV_PWS=$v timeout -k 10 300 python tools/prof_layers.py bf16 8 > gpurun_out/layers_pws$v.txt 2>&1
done
grep -E "res4.[1-5].conv3 " gpurun_out/layers_pws1.txt | cut -c1-200 | head -3
grep -E "res4.[1-5].conv3 " gpurun_out/layers_pws0.txt | cut -c1-200 | head -3
STEPS=40 bash tools/ab_modes.sh "DP_CONV_PWS=1 DP_CONV_PWS=0 DP_CONV_PWS=1 DP_CONV_PWS=0"
STEPS=20 bash tools/ab_modes.sh "DP_CONV_PWS=1 DP_CONV_PWS=0 DP_CONV_PWS=1 DP_CONV_PWS=0" "--config densepose_rcnn_R_101_FPN_s1x"
