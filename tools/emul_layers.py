"""Layer-by-layer, teacher-forced: every convolution of the 16-bit engine against the storage-emulating oracle computed FROM THE ENGINE'S
OWN INPUT of that layer (oracle/ref_storage.py, `force`). usage: emul_layers.py [case] [dtype]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from emul_common import run_forced
name = sys.argv[1] if len(sys.argv) > 1 else "full_r50_s1x_small"
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
stats, iuv, lab = run_forced(name, dt)
for k, st in stats.items():
    print("%-58s n %9d  differ %8d (%.4f %%)  max %.2f units in the last place, %.5f of the tensor's top" % (k[-58:], st["n"], st["differ"], 100.0 * st["differ"] / st["n"], st["max_ulps"], st["max_rel_to_top"]))
for k, st in iuv.items():
    print("%-58s %s" % (k, st))
print("labels (pixels, differing, largest decision margin among the differing):", lab)
