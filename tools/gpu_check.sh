#!/bin/bash
# usage (GPU box, repo root): tools/gpu_check.sh   -> gpurun_out/gpu_tests.log (the whole -m gpu suite) and, when green, gpurun_out/bench.json (the default bench line)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log; tail -5 gpurun_out/gpu_tests.log
grep -q "rc=0" gpurun_out/gpu_tests.log || { grep -E "^E |FAILED" gpurun_out/gpu_tests.log | head -20; exit 1; }
timeout -k 10 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench.json
