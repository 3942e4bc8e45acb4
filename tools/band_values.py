"""The numbers tests/test_gpu_e2e.py::test_bf16_mode_stays_in_its_measured_band holds: detections matched and the IUV deviation of the matched
ones relative to the map's largest fp32 value, per case, bf16 mode (the kernels are deterministic: the same on every box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_e2e import _label_agreement, _match_to_reference, _run  # noqa: E402

for name in ["tiny_r50_s1x_a", "full_r50_s1x_small", "full_r50_s1x_800x1333"]:
    meta, z, cfg, pred, out = _run(name, "bf16")
    hits, iuv = _match_to_reference(out, z, meta["iuv_stride"], 1.5, 0.05)
    agree, npx = _label_agreement(out, z, 1.5)
    print("%s: %d of %d detections within 1.5 px / 0.05, IUV deviation %.4f of the map's range, labels %.4f over %d px" % (
        name, hits, z["out/scores"].shape[0], iuv, agree, npx))
