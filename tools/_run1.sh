mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "deconv or conv2d_ring or random_shapes or two_sources or fused_rpn" > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log; tail -3 gpurun_out/t1.log
grep -q "rc=0" gpurun_out/t1.log || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "fp32_matches_reference_golden or batch_equals_single" > gpurun_out/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t2.log; tail -3 gpurun_out/t2.log
grep -q "rc=0" gpurun_out/t2.log || exit 1
STEPS=40 bash tools/ab_modes.sh "DP_GROUP_DECONV=1 DP_GROUP_DECONV=0 DP_GROUP_DECONV=1 DP_GROUP_DECONV=0" "--dets 100"
