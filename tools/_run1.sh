mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bottleneck" > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log; tail -3 gpurun_out/t1.log
grep -q "rc=0" gpurun_out/t1.log || { grep -E "^E |Error|FAILED" gpurun_out/t1.log | head -20; exit 1; }
for v in 1 0; do
  DP_FUSE_PAIR=$v timeout -k 10 300 python tools/prof_layers.py bf16 8 > gpurun_out/layers_pair$v.txt 2>&1
done
grep -E "res3\.[0-3]\.conv[13]" gpurun_out/layers_pair1.txt | cut -c1-200
echo ---
grep -E "res3\.[0-3]\.conv[13]" gpurun_out/layers_pair0.txt | cut -c1-200
STEPS=40 bash tools/ab_modes.sh "DP_FUSE_PAIR=1 DP_FUSE_PAIR=0 DP_FUSE_PAIR=1 DP_FUSE_PAIR=0"
