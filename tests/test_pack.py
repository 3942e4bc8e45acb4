"""The C ABI's host-side weight packer (dp_fold_frozen_bn, dp_conv_taps, dp_pack_conv_weights: csrc/dp_pack.cpp) against an
independent numpy restatement of the layout the kernels expect (dp_conv.hip header, store_tile). CPU-only: these entry
points never touch the GPU."""
import ctypes as C

import numpy as np
import pytest
import torch


def _row_perm():
    """physical row i*16 + q*4 + e of a 64-cout block carries logical cout (i>>1)*32 + q*8 + (i&1)*4 + e"""
    perm = np.zeros(64, dtype=np.int64)
    for i in range(4):
        for q in range(4):
            for e in range(4):
                perm[i * 16 + q * 4 + e] = (i >> 1) * 32 + q * 8 + (i & 1) * 4 + e
    return perm


def numpy_pack(wmat, taps, bias, cin_alloc, dtype, tap_major):
    """-> (weight: cout_w * kpad elements of raw storage in the tiled layout, ktab [n, 4], bias [cout_w], plane_major)"""
    es = 4 if dtype == "fp32" else 2
    ch, pe = 16 // es, 64 // es
    co, nt, ci = wmat.shape
    cout_w = (co + 127) // 128 * 128
    k = nt * cin_alloc
    kpad = (k + 128 // es - 1) // (128 // es) * (128 // es)
    full = np.zeros((cout_w, nt, cin_alloc), dtype=np.float32)
    full[:co, :, :ci] = wmat
    plane_major = (not tap_major) and nt > 1 and cin_alloc % pe == 0
    if plane_major:
        full = full.reshape(cout_w, nt, cin_alloc // pe, pe).transpose(0, 2, 1, 3)
    flat = np.zeros((cout_w, kpad), dtype=np.float32)
    flat[:, :k] = full.reshape(cout_w, k)
    flat = flat[_row_perm()[None, :] + 64 * np.arange(cout_w // 64)[:, None]].reshape(cout_w, kpad)
    t = torch.from_numpy(flat)
    if dtype == "bf16":
        raw = t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    elif dtype == "fp16":
        raw = t.to(torch.float16).numpy().view(np.uint16)
    else:
        raw = flat.view(np.uint32)
    # stored in 1 KiB tiles of 16 rows x 64 bytes of K (dp_common.h dp_wtile_off): tile (row group, plane) row-major
    raw = raw.reshape(cout_w // 16, 16, kpad // pe, pe).transpose(0, 2, 1, 3).reshape(cout_w, kpad).copy()
    ktab = np.zeros((kpad // ch, 4), dtype=np.int32)
    for kc in range(kpad // ch):
        k0 = kc * ch
        if k0 >= k:
            continue
        if plane_major:
            plane, within = divmod(k0, pe)
            cb, tap = divmod(plane, nt)
            c0 = cb * pe + within
        else:
            tap, c0 = divmod(k0, cin_alloc)
        ktab[kc] = (taps[tap][0], taps[tap][1], c0, 1 | (tap << 8))
    b = np.zeros((cout_w,), dtype=np.float32)
    if bias is not None:
        b[:co] = bias
    return raw, ktab, b, plane_major


@pytest.fixture(scope="module")
def L():
    from densepose_torchscript_amd import lib
    lib.build_library()
    lib.load()
    return lib


CASES = [
    # Cout, Cin, k, cin_alloc, pad, dil, in_hw, tap_major
    (64, 3, 7, 8, 3, 1, None, False),       # stem-like: Cin padded to 8, 49 taps, not a whole plane per tap
    (64, 64, 3, 64, 1, 1, None, True),      # res2 conv2: tap-major for the fused tail
    (64, 64, 3, 64, 1, 1, None, False),
    (256, 64, 1, 64, 0, 1, None, False),
    (256, 256, 3, 256, 1, 1, None, False),  # channel-block-major, several blocks
    (15, 256, 1, 256, 0, 1, None, False),   # Cout neither a multiple of 8 nor of 128
    (40, 24, 3, 24, 2, 2, None, False),     # dilated, Cin = 24: not whole planes in 16-bit storage
    (32, 32, 3, 32, 6, 6, (28, 28), False),   # dilated ASPP branch: all 9 taps live on 28x28
    (32, 32, 3, 32, 56, 56, (28, 28), False),  # dilation 56: only the centre tap survives
    (130, 16, 1, 16, 0, 1, None, False),    # two 128-row blocks, the second almost empty
]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", CASES)
def test_c_packer_equals_numpy_restatement(L, dtype, case):
    co, ci, ks, ca, pad, dil, in_hw, tap_major = case
    rng = np.random.default_rng(co * 1000 + ci + ks)
    w = (rng.standard_normal((co, ci, ks, ks)) * 0.3).astype(np.float32)
    w.flat[::7] *= 1e-3        # small and large magnitudes, exact ties of the 16-bit rounding included below
    w.flat[::11] = np.float32(1.00390625)    # exactly half way between two bf16 values: round to even
    bias = rng.standard_normal((co,)).astype(np.float32)
    # taps through the ABI, and by hand
    taps = np.empty((ks * ks, 2), dtype=np.int32)
    pos = np.empty((ks * ks,), dtype=np.int32)
    H, W = in_hw or (0, 0)
    nt = L.load().dp_conv_taps(ks, ks, pad, dil, 1, H, W, taps.ctypes.data, pos.ctypes.data)
    want_taps = [(r * dil, s * dil, r * ks + s) for r in range(ks) for s in range(ks)
                 if in_hw is None or not (r * dil - pad >= H or r * dil - pad <= -H or s * dil - pad >= W or s * dil - pad <= -W)]
    assert nt == len(want_taps) and [tuple(t) for t in taps[:nt].tolist()] == [t[:2] for t in want_taps]
    assert pos[:nt].tolist() == [t[2] for t in want_taps]
    wmat = np.ascontiguousarray(np.stack([w[:, :, q // ks, q % ks] for q in pos[:nt]], axis=1))
    p = L.PackParams()
    p.Cout, p.ntaps, p.Cin, p.cin_alloc, p.dtype, p.tap_major = co, nt, ci, ca, L.DTYPES[dtype], int(tap_major)
    info = L.PackInfo()
    assert L.load().dp_pack_conv_info(C.byref(p), C.byref(info)) == 0
    raw_ref, ktab_ref, b_ref, pm = numpy_pack(wmat, [tuple(t) for t in taps[:nt].tolist()], bias, ca, dtype, tap_major)
    assert (info.cout, info.cout_w, info.kpad, info.n_ktab, info.plane_major) == (
        (co + 7) // 8 * 8, raw_ref.shape[0], raw_ref.shape[1], ktab_ref.shape[0], int(pm))
    w_out = np.zeros(raw_ref.shape, dtype=raw_ref.dtype)
    ktab = np.full(ktab_ref.shape, -1, dtype=np.int32)
    b_out = np.full(b_ref.shape, np.nan, dtype=np.float32)
    tp = np.ascontiguousarray(taps[:nt])
    assert L.load().dp_pack_conv_weights(C.byref(p), wmat.ctypes.data, tp.ctypes.data, bias.ctypes.data, w_out.ctypes.data,
                                         ktab.ctypes.data, b_out.ctypes.data) == 0
    assert np.array_equal(w_out, raw_ref)
    assert np.array_equal(ktab, ktab_ref)
    assert np.array_equal(b_out, b_ref)


def test_frozen_bn_fold_is_bit_exact_with_numpy(L):
    """batch_norm.py:31,54-62 folded: scale = gamma * (1 / sqrt(var + 1e-5)), shift = beta - mean * scale, all fp32."""
    rng = np.random.default_rng(7)
    co, per = 37, 64 * 9
    w = rng.standard_normal((co, per)).astype(np.float32)
    g, b, m = (rng.standard_normal((co,)).astype(np.float32) for _ in range(3))
    v = rng.uniform(0.01, 4.0, (co,)).astype(np.float32)
    w_out, shift = np.empty_like(w), np.empty((co,), dtype=np.float32)
    assert L.load().dp_fold_frozen_bn(w.ctypes.data, co, per, g.ctypes.data, b.ctypes.data, m.ctypes.data, v.ctypes.data, 1e-5,
                                      w_out.ctypes.data, shift.ctypes.data) == 0
    scale = g * (np.float32(1.0) / np.sqrt(v + np.float32(1e-5)))
    assert np.array_equal(shift, b - m * scale)
    assert np.array_equal(w_out, w * scale[:, None])


def test_packer_rejects_bad_arguments(L):
    lib = L.load()
    p, info = L.PackParams(), L.PackInfo()
    p.Cout, p.ntaps, p.Cin, p.cin_alloc, p.dtype = 8, 1, 7, 7, L.DP_BF16
    assert lib.dp_pack_conv_info(C.byref(p), C.byref(info)) == -1 and b"multiple of 8" in lib.dp_last_error()
    p.cin_alloc, p.Cout, p.ntaps, p.tap_major = 8, 256, 9, 1
    assert lib.dp_pack_conv_info(C.byref(p), C.byref(info)) == -1 and b"tap-major" in lib.dp_last_error()
    p.tap_major = 0
    assert lib.dp_pack_conv_info(C.byref(p), C.byref(info)) == 0
    assert lib.dp_pack_conv_weights(C.byref(p), None, None, None, None, None, None) == -1
    assert lib.dp_conv_taps(0, 3, 1, 1, 1, 0, 0, None, None) < 0


def test_packed_model_layers_use_the_c_packer(L):
    """pack.PackedConv (what the engine uploads) == the numpy restatement for a layer of each kind of the tiny model -
    on the CPU device, no GPU needed."""
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
    from densepose_torchscript_amd.pack import PackedModel
    cfg = get_config("densepose_rcnn_R_50_FPN_DL_s1x", TINY_OPTS)
    model = PackedModel(cfg, make_synthetic_state(cfg, 0), L.DP_BF16, torch.device("cpu"))
    for name in ("stem", "backbone.bottom_up.res2.0.conv2", "fpn_output3", "rpn_head", "fc1", "aspp3", "dp_fcn2"):
        l = model.layers[name]
        assert l.weight.dtype == torch.bfloat16 and tuple(l.weight.shape) == (l.cout_w, l.kpad)
        assert l.ktab.shape == (l.kpad // 8, 4) and l.bias.shape == (l.cout_w,)
    # pool 28 (densepose/config.py:177): dilation 6 and 12 keep their 9 taps, dilation 56 only the centre one (deeplab.py:33)
    assert (model.layers["aspp1"].ntaps, model.layers["aspp2"].ntaps, model.layers["aspp3"].ntaps) == (9, 9, 1)
    assert model.layers["fpn_output3"].plane_major and not model.layers["backbone.bottom_up.res2.0.conv2"].plane_major
    # (3x3 over 32 channels = one 64-byte plane per tap: channel-block major; over 8 channels a tap is a quarter plane: tap order)
