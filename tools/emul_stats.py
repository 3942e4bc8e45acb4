"""16-bit engine against the storage-emulating oracle (oracle/ref_storage.py): per-stage deviation statistics for every case the
tests gate (tests/test_gpu_e2e.py::test_16bit_mode_matches_the_storage_emulating_oracle takes its bounds from this output)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden_case_inputs, load_golden
from emul_common import run_pair, stage_stats, label_stats

cases = [("full_r50_s1x_800x1333", "bf16"), ("full_r50_s1x_800x1333", "fp16"), ("full_r101_s1x_small", "bf16"), ("full_r50_dl_p28", "bf16"),
         ("tiny_r101_dl_p28_video", "bf16"), ("tiny_r101_dl_p28_video", "fp16"), ("tiny_r50_legacy", "bf16"), ("tiny_r50_s1x_a", "bf16")]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, dt in cases:
    r = run_pair(name, dt)
    print("== %s %s: R = %d, decoder_fold %s, fused shortcuts %d" % (name, dt, r["R"], r["fold"], len(r["fused"])))
    for k, (a, b) in r["stages"].items():
        print("   %-14s %s" % (k, stage_stats(a, b, dt)))
    print("   labels: %s" % (label_stats(r),))
