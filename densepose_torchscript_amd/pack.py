"""Weight pipeline: canonical state dict -> device-resident packed layers for the HIP kernels.

* FrozenBN folded into the conv (scale into the weights, shift into the bias):
  y = (x - mean) * rsqrt(var + 1e-5) * gamma + beta   (/root/reference/detectron2/layers/batch_norm.py:31,54-62)
* OIHW -> [Cout_pad128][K] with K = (tap, channel) contiguous ("KRSC"), zero padded to the 128-byte K step,
  plus the per-16-byte-chunk tap table (ktab) consumed by dp_conv2d_nhwc. Both steps (and the BN fold) are host-side entry
  points of the C ABI (dp_fold_frozen_bn, dp_conv_taps, dp_pack_conv_weights: csrc/dp_pack.cpp); this module decides WHAT to
  pack (layer list, fc1 / deconv / stem re-arrangements) and uploads the result.
* fc1's K axis is permuted from the reference's NCHW flatten (c, y, x) (box_head.py:70-71) to NHWC (y, x, c).
* ConvTranspose2d(k4, s2, p1) (chart.py:45-59) is split into its four 2x2 sub-pixel convolutions.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .lib import DP_BF16, DP_F16, DP_F32
from .weights import decoder_layout, resnet_blocks

BN_EPS = 1e-5


def round_up(x, m):
    return (x + m - 1) // m * m


class PackedConv:
    """One dp_conv2d_nhwc layer resident on the device. The packing itself (row permutation, K order, tap table, dtype
    conversion) is the C ABI's host-side packer (dp_pack_conv_weights, csrc/dp_pack.cpp); tests/test_pack.py checks it
    against an independent numpy restatement."""

    def __init__(self, name, wmat, taps, bias, cin_alloc, cout, stride, hi_off, wi_off, dtype, device, plane_major=None):
        # wmat: float32 [Cout, ntaps, Cin] ; taps: list of (dy, dx)
        # plane_major=False (opt-in, PackedModel uses it for the res2 conv2 layers) keeps K TAP major - K = tap * Cin +
        # channel - which is what the fused bottleneck tail (dp_bottleneck_tail_nhwc) consumes: the two 64-byte halves of a
        # pixel's 128-byte line are then read by consecutive loads. Only for layers that run on the table-driven generic
        # kernel (Cout <= 64): the LDS-ring kernels enumerate the taps as planes 0..ntaps-1 of the channel-block-major order.
        lib = L.load()
        self.name = name
        self.dtype = dtype
        wmat = np.ascontiguousarray(wmat, dtype=np.float32)
        co, nt, ci = wmat.shape
        assert nt == len(taps) and ci <= cin_alloc and cin_alloc % 8 == 0 and co == cout
        p = L.PackParams()
        p.Cout, p.ntaps, p.Cin, p.cin_alloc, p.dtype = co, nt, ci, cin_alloc, dtype
        p.tap_major = 1 if plane_major is False else 0
        info = L.PackInfo()
        L.check(lib.dp_pack_conv_info(C.byref(p), C.byref(info)), "dp_pack_conv_info[%s]" % name)
        self.cin = cin_alloc
        self.cout, self.cout_w, self.kpad = info.cout, info.cout_w, info.kpad
        self.split_k = 0      # > 1: dp_conv_params.split_k of this layer (PackedModel sets it for the long-K layers, see set_split_k)
        self.plane_major = bool(info.plane_major)
        wdt = {DP_F32: np.float32, DP_BF16: np.uint16, DP_F16: np.float16}[dtype]
        w_out = np.empty((self.cout_w, self.kpad), dtype=wdt)
        ktab = np.empty((info.n_ktab, 4), dtype=np.int32)
        b_out = np.empty((self.cout_w,), dtype=np.float32)
        taps_a = np.ascontiguousarray(np.asarray(taps, dtype=np.int32).reshape(nt, 2))
        bias_a = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        assert bias_a is None or bias_a.shape == (co,)
        L.check(lib.dp_pack_conv_weights(C.byref(p), wmat.ctypes.data, taps_a.ctypes.data, None if bias_a is None else bias_a.ctypes.data,
                                         w_out.ctypes.data, ktab.ctypes.data, b_out.ctypes.data), "dp_pack_conv_weights[%s]" % name)
        t = torch.from_numpy(w_out)
        if dtype == DP_BF16:
            t = t.view(torch.bfloat16)
        self.weight = t.to(device).contiguous()
        self.ktab = torch.from_numpy(ktab).to(device)
        self.bias = torch.from_numpy(b_out).to(device)
        self.ntaps = nt
        self.stride_w = 0           # horizontal stride when it differs from `stride` (paired-pixel stem), 0 = same
        self.stride = stride
        self.hi_off = hi_off
        self.wi_off = wi_off
        self.macs_per_pixel = co * nt * ci  # algorithmic MACs per output pixel (un-padded)

    def nbytes(self):
        return self.weight.numel() * self.weight.element_size() + self.ktab.numel() * 4 + self.bias.numel() * 4


def _fold_bn(w, st, norm_name):
    """FrozenBN folded into the conv (batch_norm.py:31,54-62) by dp_fold_frozen_bn: (scaled weights, per-cout shift)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    g, b, m, v = (np.ascontiguousarray(st[norm_name + s], dtype=np.float32) for s in (".weight", ".bias", ".running_mean", ".running_var"))
    co = w.shape[0]
    w_out, shift = np.empty_like(w), np.empty((co,), dtype=np.float32)
    L.check(L.load().dp_fold_frozen_bn(w.ctypes.data, co, w.size // co, g.ctypes.data, b.ctypes.data, m.ctypes.data, v.ctypes.data,
                                       BN_EPS, w_out.ctypes.data, shift.ctypes.data), "dp_fold_frozen_bn[%s]" % norm_name)
    return w_out, shift


def conv_from_oihw(name, w, bias, cin_alloc, stride, pad, dil, dtype, device, in_hw=None, plane_major=None):
    co, ci, R, S = w.shape
    taps = np.empty((R * S, 2), dtype=np.int32)
    pos = np.empty((R * S,), dtype=np.int32)
    # taps that can never land inside the map for ANY output pixel contribute exactly 0 and are dropped
    # (deeplab.py:33 dilation 56 on a 28x28 ROI map: only the centre tap survives)
    H, W = in_hw if in_hw is not None else (0, 0)
    nt = L.load().dp_conv_taps(R, S, pad, dil, stride, H, W, taps.ctypes.data, pos.ctypes.data)
    if nt <= 0:
        raise L.DensePoseHipError("dp_conv_taps[%s] failed (%d)" % (name, nt))
    wmat = np.stack([w[:, :, int(q) // S, int(q) % S] for q in pos[:nt]], axis=1).astype(np.float32)  # [co, ntaps, ci]
    return PackedConv(name, wmat, [tuple(t) for t in taps[:nt].tolist()], bias, cin_alloc, co, stride, -pad, -pad, dtype, device,
                      plane_major=plane_major)


def dual_source_pointwise(name, w1, b1, cin1_alloc, w2, b2, cin2_alloc, stride2, dtype, device):
    """out = W1 x1 + W2 x2[::stride2, ::stride2] + (b1 + b2) as ONE pointwise layer whose K axis is the cin1 channels of the first
    source followed by the cin2 channels of the second (dp_conv_params.in2): conv3 of the first bottleneck block of a ResNet stage
    with the block's projection shortcut as extra K planes (resnet.py:189-205). w1: [Cout, cin1, 1, 1], w2: [Cout, cin2, 1, 1]
    (FrozenBN already folded). Both channel counts must be multiples of 32 (a 64-byte K plane belongs to one source)."""
    co, c1 = w1.shape[0], w1.shape[1]
    c2 = w2.shape[1]
    es = 4 if dtype == DP_F32 else 2
    assert w2.shape[0] == co and (cin1_alloc * es) % 64 == 0 and (cin2_alloc * es) % 64 == 0 and c1 <= cin1_alloc and c2 <= cin2_alloc
    assert ((cin1_alloc + cin2_alloc) * es) % 128 == 0, "the K axis is padded to 128 bytes: the two sources together must fill it"
    wcat = np.zeros((co, 1, cin1_alloc + cin2_alloc), dtype=np.float32)
    wcat[:, 0, :c1] = w1.reshape(co, c1)
    wcat[:, 0, cin1_alloc:cin1_alloc + c2] = w2.reshape(co, c2)
    layer = PackedConv(name, wcat, [(0, 0)], (np.asarray(b1, np.float32) + np.asarray(b2, np.float32)), cin1_alloc + cin2_alloc, co,
                       1, 0, 0, dtype, device)
    assert layer.kpad == cin1_alloc + cin2_alloc
    layer.cin1, layer.cin2, layer.stride2 = cin1_alloc, cin2_alloc, stride2
    layer.macs_per_pixel = co * (c1 + c2)
    return layer


def stem_paired_conv(name, w, bias, dtype, device):
    """The 7x7 stride-2 pad-3 stem (resnet.py:350-353) over the PAIRED image layout of dp_preprocess_u8 (paired=1): cell j of a
    row = the 4-channel pixels 2j - 3 and 2j - 2 (3 real channels + 1 zero each). Output column wo reads pixels 2wo - 3 ..
    2wo + 3 = cells wo .. wo + 3, so the layer is a 7 x 4-tap convolution with stride (2, 1) over 8-channel cells: tap
    (dy, dxp), element e * 4 + c  <->  kernel position (dy, dx = 2 dxp + e), channel c (dx = 7 and c = 3 carry zero weights).
    Same K order as the plain form (dy, dx, c ascending), 224 K elements instead of 392."""
    co, ci, R, S = w.shape
    assert (R, S) == (7, 7) and ci <= 3
    taps, cols = [], []
    for dy in range(7):
        for dxp in range(4):
            cell = np.zeros((co, 8), dtype=np.float32)
            for e in range(2):
                dx = 2 * dxp + e
                if dx < 7:
                    cell[:, e * 4:e * 4 + ci] = w[:, :, dy, dx]
            taps.append((dy, dxp))
            cols.append(cell)
    layer = PackedConv(name, np.stack(cols, axis=1), taps, bias, 8, co, 2, -3, 0, dtype, device)
    layer.stride_w = 1
    layer.macs_per_pixel = co * 49 * ci     # algorithmic MACs (the zero weights of the paired form are not work)
    return layer


def linear_as_conv(name, w, bias, cin_alloc, dtype, device):
    co, ci = w.shape
    return PackedConv(name, w.reshape(co, 1, ci).astype(np.float32), [(0, 0)], bias, cin_alloc, co, 1, 0, 0, dtype, device)


def deconv_parity_convs(name, w_list, b_list, cin_alloc, dtype, device):
    """w_list: ConvTranspose2d weights [Cin, Cout_i, 4, 4] concatenated along Cout. Returns {(a, b): PackedConv}.
    out[2i+a, 2j+b] = sum over input rows iy with ky = (2i+a) + 1 - 2*iy in [0,3]:
       a=0: (dy=0, ky=1), (dy=-1, ky=3) ;  a=1: (dy=+1, ky=0), (dy=0, ky=2)   (same for columns)."""
    w = np.concatenate(w_list, axis=1).astype(np.float32)  # [Cin, Ctot, 4, 4]
    bias = np.concatenate(b_list).astype(np.float32)
    sel = {0: [(0, 1), (-1, 3)], 1: [(1, 0), (0, 2)]}
    out = {}
    for a in (0, 1):
        for b in (0, 1):
            taps, cols = [], []
            for dy, ky in sel[a]:
                for dx, kx in sel[b]:
                    taps.append((dy, dx))
                    cols.append(w[:, :, ky, kx].T)  # [Ctot, Cin]
            wmat = np.stack(cols, axis=1)
            out[(a, b)] = PackedConv("%s[%d%d]" % (name, a, b), wmat, taps, bias, cin_alloc, w.shape[1], 1, 0, 0, dtype, device)
    return out


def set_split_k(layer, segments):
    """Mark a layer for split-K (dp_conv_params.split_k) when the kernels behind it take the layer: 16-bit storage, 64-byte K planes
    inside one tap, Cout in 128-slices, at least two planes per segment. The value is fixed per LAYER - never per batch - so a
    row's summation order does not depend on what else is in the batch."""
    es = 4 if layer.dtype == DP_F32 else 2
    planes = layer.kpad * es // 64
    if layer.dtype != DP_F32 and (layer.cin * es) % 64 == 0 and layer.cout % 128 == 0 and planes >= 2 * segments:
        layer.split_k = segments


class PackedModel:
    def __init__(self, cfg, state, dtype, device):
        self.cfg = cfg
        self.dtype = dtype
        self.device = device
        L = {}
        st = state
        bu = "backbone.bottom_up."

        def bnconv(name, cin_alloc, stride, pad, plane_major=None):
            w, shift = _fold_bn(st[name + ".weight"].astype(np.float32), st, name + ".norm")
            return conv_from_oihw(name, w, shift, cin_alloc, stride, pad, 1, dtype, device, plane_major=plane_major)

        def bconv(name, cin_alloc, stride=1, pad=0, dil=1, in_hw=None, bias=True):
            return conv_from_oihw(name, st[name + ".weight"].astype(np.float32),
                                  st[name + ".bias"].astype(np.float32) if bias else None, cin_alloc, stride, pad, dil, dtype, device, in_hw)

        A8 = lambda c: round_up(c, 8)  # noqa: E731
        w, shift = _fold_bn(st[bu + "stem.conv1.weight"].astype(np.float32), st, bu + "stem.conv1.norm")
        L["stem"] = stem_paired_conv(bu + "stem.conv1", w, shift, dtype, device)
        for stage, b, cin, cmid, cout, stride, sc in resnet_blocks(cfg):
            p = "%s%s.%d." % (bu, stage, b)
            if sc:
                L[p + "shortcut"] = bnconv(p + "shortcut", A8(cin), stride, 0)
            L[p + "conv1"] = bnconv(p + "conv1", A8(cin), stride, 0)
            # 64 -> 64 3x3 in 16-bit storage (the res2 blocks): tap-major K for the fused bottleneck tail
            L[p + "conv2"] = bnconv(p + "conv2", A8(cmid), 1, 1, plane_major=False if (cmid == 64 and dtype != DP_F32) else None)
            L[p + "conv3"] = bnconv(p + "conv3", A8(cmid), 1, 0)
            # (res5's 3x3 - 25 x 42 pixels per frame against K = 4608 - ran split-K in three segments in round 3; in the 16-bit modes it
            # now runs on the row-streaming weight-stationary kernel, dp_conv_rows.hip, whatever the batch)
            # (res5's conv1 - 132 tiles of 128 x 256 at batch 8, each walking K = 2048 - was tried with two split-K segments in round 4:
            # 40 -> 50 us, the 128 x 128 split instance plus the reduction pass cost more than the idle half of the chip; fpn_lateral5
            # with four: 34 -> 37 us. Not set.)
            # first block of res3 / res4 / res5 in the 16-bit modes: the projection shortcut as extra K planes of conv3 (one launch,
            # the shortcut tensor is never written or read back; fp32 parity mode keeps the reference's two convolutions + add)
            # (res2.0, stride 1: the same matrix feeds the fused bottleneck tail, dp_bottleneck_params.sc_in - round 5)
            if sc and dtype != DP_F32 and A8(cmid) % 32 == 0 and A8(cin) % 32 == 0 and (A8(cmid) + A8(cin)) % 64 == 0:
                w3, s3 = _fold_bn(st[p + "conv3.weight"].astype(np.float32), st, p + "conv3.norm")
                ws, ss = _fold_bn(st[p + "shortcut.weight"].astype(np.float32), st, p + "shortcut.norm")
                L[p + "conv3+shortcut"] = dual_source_pointwise(p + "conv3+shortcut", w3, s3, A8(cmid), ws, ss, A8(cin), stride, dtype, device)
        c = cfg.res2_out
        F = A8(cfg.fpn_out)
        for lvl in (2, 3, 4, 5):
            L["fpn_lateral%d" % lvl] = bconv("backbone.fpn_lateral%d" % lvl, A8(c), 1, 0)
            L["fpn_output%d" % lvl] = bconv("backbone.fpn_output%d" % lvl, F, 1, 1)
            c *= 2
        pg = "proposal_generator.rpn_head."
        L["rpn_conv"] = bconv(pg + "conv", F, 1, 1)
        A = len(cfg.anchor_ratios)
        wo = st[pg + "objectness_logits.weight"].astype(np.float32)
        wd = st[pg + "anchor_deltas.weight"].astype(np.float32)
        w = np.concatenate([wo, wd], axis=0)  # channel a | A + 4a + c   (rpn.py:331: a*4 + coord)
        bcat = np.concatenate([st[pg + "objectness_logits.bias"], st[pg + "anchor_deltas.bias"]]).astype(np.float32)
        L["rpn_head"] = conv_from_oihw("rpn_head", w, bcat, F, 1, 0, 1, dtype, device)
        self.rpn_head_c = L["rpn_head"].cout
        # the same head as a plain [16][F] matrix (+ bias) for the fused form: dp_conv2d_nhwc's head_w / head_b, applied inside
        # the 3x3 conv's epilogue where the 256-cout ring kernel runs the level (engine.rpn)
        self.rpn_head_plain = None
        if self.rpn_head_c == 16 and dtype != DP_F32:
            wp = np.zeros((16, F), dtype=np.float32)
            wp[: w.shape[0], : w.shape[1]] = w.reshape(w.shape[0], -1)
            bp = np.zeros((16,), dtype=np.float32)
            bp[: bcat.shape[0]] = bcat
            tdt = torch.bfloat16 if dtype == DP_BF16 else torch.float16
            self.rpn_head_plain = (torch.from_numpy(wp).to(tdt).to(device).contiguous(), torch.from_numpy(bp).to(device))
        # box head: fc1 K permuted (c, y, x) -> (y, x, c) over the ALLOCATED channel count
        P = cfg.box_pool
        w1 = st["roi_heads.box_head.fc1.weight"].astype(np.float32)
        o = w1.shape[0]
        w1 = w1.reshape(o, cfg.fpn_out, P, P).transpose(0, 2, 3, 1)  # [o, y, x, c]
        w1p = np.zeros((o, P, P, F), dtype=np.float32)
        w1p[..., : cfg.fpn_out] = w1
        L["fc1"] = linear_as_conv("fc1", w1p.reshape(o, P * P * F), st["roi_heads.box_head.fc1.bias"], P * P * F, dtype, device)
        # K = 12544 (392 planes of 64 B) in two segments: batch 8 then fills the chip once with 256 x 256 tiles (0.199 -> 0.192 ms), a
        # single frame's 1000 rows run 128 instead of 64 workgroups (0.151 -> 0.086 ms); four segments: 0.216 / 0.054 ms
        set_split_k(L["fc1"], 2)
        fin = o
        for i in range(1, cfg.box_num_fc):
            n = "roi_heads.box_head.fc%d" % (i + 1)
            L["fc%d" % (i + 1)] = linear_as_conv(n, st[n + ".weight"].astype(np.float32), st[n + ".bias"], A8(fin), dtype, device)
            fin = st[n + ".weight"].shape[0]
        wcls = st["roi_heads.box_predictor.cls_score.weight"].astype(np.float32)
        wbox = st["roi_heads.box_predictor.bbox_pred.weight"].astype(np.float32)
        L["box_out"] = linear_as_conv("box_out", np.concatenate([wcls, wbox], axis=0),
                                      np.concatenate([st["roi_heads.box_predictor.cls_score.bias"],
                                                      st["roi_heads.box_predictor.bbox_pred.bias"]]).astype(np.float32),
                                      A8(fin), dtype, device)
        if cfg.dp_decoder_on:
            D = A8(cfg.dp_decoder_dims)
            for lvl, n in decoder_layout(cfg):
                for k in range(n):
                    nm = "roi_heads.decoder.%s.%d" % (lvl, 2 * k)
                    L[nm] = bconv(nm, F if k == 0 else D, 1, 1)
            L["decoder_predictor"] = bconv("roi_heads.decoder.predictor", D, 1, 0)
        hd = "roi_heads.densepose_head."
        Pd = cfg.dp_pool
        cin = F
        if cfg.is_deeplab:
            a = hd + "ASPP."
            L["aspp0"] = bconv(a + "convs.0.0", F, 1, 0, bias=False)
            for i, d in ((1, 6), (2, 12), (3, 56)):
                L["aspp%d" % i] = bconv(a + "convs.%d.0" % i, F, 1, d, d, in_hw=(Pd, Pd), bias=False)
            L["aspp4"] = bconv(a + "convs.4.1", F, 1, 0, bias=False)
            L["aspp_project"] = bconv(a + "project.0", 5 * F, 1, 0, bias=False)
            self.gn = {}
            for i, nm in ((0, "convs.0.1"), (1, "convs.1.1"), (2, "convs.2.1"), (3, "convs.3.1"), (4, "convs.4.2")):
                self.gn["aspp%d" % i] = (torch.from_numpy(st[a + nm + ".weight"].astype(np.float32)).to(device),
                                         torch.from_numpy(st[a + nm + ".bias"].astype(np.float32)).to(device))
        for i in range(cfg.dp_num_convs):
            n = hd + "body_conv_fcn%d" % (i + 1)
            L["dp_fcn%d" % (i + 1)] = bconv(n, cin, 1, 1, bias=not cfg.is_deeplab)
            if cfg.is_deeplab:
                self.gn["dp_fcn%d" % (i + 1)] = (torch.from_numpy(st[n + ".norm.weight"].astype(np.float32)).to(device),
                                                 torch.from_numpy(st[n + ".norm.bias"].astype(np.float32)).to(device))
            cin = A8(cfg.dp_head_dim)
        pr = "roi_heads.densepose_predictor."
        names = ("ann_index_lowres", "index_uv_lowres", "u_lowres", "v_lowres")
        self.deconv = deconv_parity_convs("dp_predictor", [st[pr + n + ".weight"] for n in names],
                                          [st[pr + n + ".bias"] for n in names], cin, dtype, device)
        self.iuv_c = self.deconv[(0, 0)].cout
        self.layers = L

    def nbytes(self):
        n = sum(l.nbytes() for l in self.layers.values()) + sum(l.nbytes() for l in self.deconv.values())
        return n

    def parameter_tensors(self):
        """Every device tensor of the packed model, in a fixed order (RCCL broadcast of the weights)."""
        out = []
        for k in sorted(self.layers):
            l = self.layers[k]
            out += [l.weight, l.ktab, l.bias]
        for k in sorted(self.deconv):
            l = self.deconv[k]
            out += [l.weight, l.ktab, l.bias]
        for k in sorted(getattr(self, "gn", {})):
            out += list(self.gn[k])
        if self.rpn_head_plain is not None:
            out += list(self.rpn_head_plain)
        return out
