"""Driven by tests/test_asan.py inside a process that preloads the AddressSanitizer runtime: loads the ASAN build of the
library (host side instrumented, csrc/Makefile target `asan`) through plain ctypes - no torch, no GPU - and exercises the
host code that touches caller memory: argument validation of every params struct, the workspace-size queries and the
weight packer writing into exactly-sized buffers. Any out-of-bounds access aborts the process with an ASAN report."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from densepose_torchscript_amd import lib as L  # noqa: E402  (ctypes + os only)

lib = L.load()
assert os.path.basename(L.LIB_PATH).endswith("_asan.so"), L.LIB_PATH
# argument validation: zeroed / inconsistent structs must be rejected before anything is launched
for cls, fn in ((L.ConvParams, lib.dp_conv2d_nhwc), (L.NmsParams, lib.dp_batched_nms), (L.RoiAlignParams, lib.dp_roi_align_nhwc),
                (L.BoxDecodeParams, lib.dp_box_decode_score), (L.PostprocessParams, lib.dp_postprocess_boxes),
                (L.IuvParams, lib.dp_iuv_upsample_split), (L.GroupNormParams, lib.dp_groupnorm_relu_nhwc),
                (L.PreprocessParams, lib.dp_preprocess_u8), (L.BottleneckParams, lib.dp_bottleneck_tail_nhwc),
                (L.ResizeParams, lib.dp_resize_u8_bilinear), (L.RpnLevelParams, lib.dp_rpn_topk_decode)):
    p = cls()
    rc = fn(C.byref(p), None)
    assert rc in (0, -1, -2), (cls.__name__, rc)      # 0: an empty problem (R = 0 detections) is legal and launches nothing
    assert rc == 0 or lib.dp_last_error()
p = L.ConvParams()
p.N, p.H, p.W, p.Ho, p.Wo, p.Cin, p.Cout = 1, 4, 4, 4, 4, 7, 8
assert lib.dp_conv2d_nhwc(C.byref(p), None) == -1
assert lib.dp_conv2d_kernel_class(C.byref(p)) >= 0 and lib.dp_conv2d_tile_rows(C.byref(p)) > 0
assert lib.dp_nms_workspace_bytes(2, 1000) > 0 and lib.dp_rpn_topk_workspace_bytes(8, 200, 336, 3) > 0
# ABI 6: the policy table (string keys, out parameter), the pair kernel's and the grouped launch's argument validation
v = C.c_int64(0)
for i in range(lib.dp_policy_num_keys()):
    key = lib.dp_policy_key(i)
    assert lib.dp_get_policy(key, C.byref(v)) == 0 and lib.dp_set_policy(key, v.value) == 0
assert lib.dp_policy_key(lib.dp_policy_num_keys()) is None and lib.dp_policy_key(-1) is None
assert lib.dp_set_policy(b"no_such_key", 1) == -1 and lib.dp_set_policy(None, 1) == -1 and lib.dp_get_policy(b"conv_ws", None) == -1
assert lib.dp_set_policy(b"x" * 4096, 1) == -1
lib.dp_reset_policy()
q = L.PairParams()
assert lib.dp_bottleneck_pair_supported(C.byref(q)) == 0 and lib.dp_bottleneck_pair_nhwc(C.byref(q), None) == -2
q.Cmid, q.Cout, q.Cmid_next, q.Kpad3, q.Kpad1n, q.dtype, q.M = 128, 512, 128, 128, 512, L.DP_BF16, 64
assert lib.dp_bottleneck_pair_supported(C.byref(q)) == 1 and lib.dp_bottleneck_pair_nhwc(C.byref(q), None) == -1     # null pointers
q.M = 1 << 22
assert lib.dp_bottleneck_pair_supported(C.byref(q)) == 0                                                               # beyond 32-bit offsets
# ABI 7: the first block of res2 with its projection shortcut inside the fused tail (dp_bottleneck_params.sc_in)
t = L.BottleneckParams()
t.N, t.H, t.W, t.Cmid, t.Cout, t.Kpad2, t.Kpad3, t.ntaps2, t.k_order2, t.dtype = 1, 8, 8, 64, 256, 576, 128, 9, 1, L.DP_BF16
t.hi_off2 = t.wi_off2 = -1
t.Csc, t.sc_in = 64, 4096
assert lib.dp_bottleneck_tail_supported(C.byref(t)) == 1 and lib.dp_bottleneck_tail_nhwc(C.byref(t), None) == -1      # null pointers
t.residual = 4096
assert lib.dp_bottleneck_tail_supported(C.byref(t)) == 0 and lib.dp_bottleneck_tail_nhwc(C.byref(t), None) == -2 and b"shortcut form" in lib.dp_last_error()
t.residual, t.Csc = None, 32
assert lib.dp_bottleneck_tail_supported(C.byref(t)) == 0
g = L.ConvParams()
g.N, g.H, g.W, g.Ho, g.Wo, g.Cin, g.Cout, g.Cout_w, g.Kpad, g.stride, g.ntaps, g.dtype = 2, 28, 28, 28, 28, 512, 80, 128, 2048, 1, 4, L.DP_BF16
g.in_, g.bias, g.n_groups = 4096, 4096, 5
assert lib.dp_conv2d_nhwc(C.byref(g), None) == -1 and b"n_groups" in lib.dp_last_error()
g.n_groups = 4
assert lib.dp_conv2d_nhwc(C.byref(g), None) == -1 and b"group 0" in lib.dp_last_error()                                # null per-group pointers
# the packer into exactly-sized buffers, every dtype, shapes with ragged padding
rng = np.random.default_rng(0)
for dtype, wdt in ((L.DP_F32, np.float32), (L.DP_BF16, np.uint16), (L.DP_F16, np.uint16)):
    for co, nt, ci, ca, tm in ((64, 49, 3, 8, 0), (64, 9, 64, 64, 1), (130, 1, 16, 16, 0), (15, 9, 24, 24, 0), (256, 9, 256, 256, 0)):
        q, info = L.PackParams(), L.PackInfo()
        q.Cout, q.ntaps, q.Cin, q.cin_alloc, q.dtype, q.tap_major = co, nt, ci, ca, dtype, tm
        assert lib.dp_pack_conv_info(C.byref(q), C.byref(info)) == 0
        wmat = rng.standard_normal((co, nt, ci)).astype(np.float32)
        taps = rng.integers(-3, 4, (nt, 2)).astype(np.int32)
        bias = rng.standard_normal((co,)).astype(np.float32)
        w_out = np.empty((info.cout_w, info.kpad), dtype=wdt)
        ktab = np.empty((info.n_ktab, 4), dtype=np.int32)
        b_out = np.empty((info.cout_w,), dtype=np.float32)
        assert lib.dp_pack_conv_weights(C.byref(q), wmat.ctypes.data, taps.ctypes.data, bias.ctypes.data, w_out.ctypes.data,
                                        ktab.ctypes.data, b_out.ctypes.data) == 0
        assert lib.dp_pack_conv_weights(C.byref(q), wmat.ctypes.data, taps.ctypes.data, None, w_out.ctypes.data,
                                        ktab.ctypes.data, b_out.ctypes.data) == 0
w = rng.standard_normal((5, 27)).astype(np.float32)
v4 = [rng.uniform(0.5, 2.0, (5,)).astype(np.float32) for _ in range(4)]
sh = np.empty((5,), dtype=np.float32)
assert lib.dp_fold_frozen_bn(w.ctypes.data, 5, 27, v4[0].ctypes.data, v4[1].ctypes.data, v4[2].ctypes.data, v4[3].ctypes.data, 1e-5,
                             w.ctypes.data, sh.ctypes.data) == 0
taps = np.empty((9, 2), dtype=np.int32)
pos = np.empty((9,), dtype=np.int32)
assert lib.dp_conv_taps(3, 3, 56, 56, 1, 28, 28, taps.ctypes.data, pos.ctypes.data) == 1
print("asan driver ok")
