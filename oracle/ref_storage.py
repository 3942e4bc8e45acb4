"""ORACLE for the 16-bit modes - the CPU restatement of oracle/ref_cpu.py with every tensor rounded to the storage type exactly where
the HIP engine stores it. TEST INFRASTRUCTURE (imported by tests/ only; the product package never imports oracle/).

oracle/ref_cpu.py is the reference's arithmetic in fp32 and is pinned bit for bit against the reference's own outputs
(tests/golden). The throughput modes of the engine (bf16, fp16: every BASELINE.json config that names a dtype) cannot be compared
with it tighter than a band - they round ~60 times on the way. This class emulates those roundings, so that the engine's 16-bit
result can be checked against a CPU computation up to the ORDER of the fp32 accumulation only:
  * weights: FrozenBN folded into the convolution in fp32 (batch_norm.py:31,54-62: scale = gamma * rsqrt(var + 1e-5) into the
    weights, beta - mean * scale into an fp32 bias), THEN rounded to the storage type; biases stay fp32 (pack.py),
  * every layer output is rounded where the engine writes it to memory; sums the engine forms in its fp32 epilogues (residual add,
    FPN top-down add, the fused projection shortcut, the decoder's level sum folded into the convolutions that produce its terms)
    are formed in fp32 here too and rounded once,
  * outputs the engine keeps in fp32 (RPN head, box logits, the predictor's four maps) are not rounded; boxes / anchors / NMS /
    softmax are fp32 in both (box_regression.py:84, nms.py:20).
Which per-LAYER choices the engine made (projection shortcut fused into conv3, decoder sum folded) is passed in explicitly.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .ref_cpu import OracleModel

_EPS = 1e-5


class StorageOracle(OracleModel):
    def __init__(self, cfg, state, storage, fused_shortcuts=(), decoder_fold=True, nms_trick_max_numel=4000):
        super().__init__(cfg, state, nms_trick_max_numel)
        self.sdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[storage]
        self.fused_shortcuts = set(fused_shortcuts)      # block prefixes ("backbone.bottom_up.res3.0.") whose shortcut rides in conv3
        self.decoder_fold = bool(decoder_fold)
        self._folded = {}
        # "teacher forcing" (tests): force[name] = list of the engine's outputs of layer `name`, in call order. Each layer is then
        # compared with the engine's tensor (self.forced_stats[name]) and CONTINUES from the engine's tensor: every layer starts from
        # bit-identical inputs, so what is measured per layer is the order of the fp32 accumulation alone, not 60 layers of it.
        self.force = None
        self.forced_stats = {}

    # ---------------------------------------------------------------- rounding helpers
    def q(self, t):
        """round to the storage type (RNE like the kernels' v_cvt_pk / the packer), back to fp32"""
        return t.to(self.sdt).to(torch.float32)

    def wq(self, name):
        return self.q(self.w[name])

    def folded(self, name):
        """conv `name` followed by FrozenBN `name`.norm -> (rounded folded weights, fp32 shift); dp_fold_frozen_bn's arithmetic"""
        if name not in self._folded:
            w = self.w
            scale = w[name + ".norm.weight"] * (1.0 / torch.sqrt(w[name + ".norm.running_var"] + np.float32(_EPS)))
            shift = w[name + ".norm.bias"] - w[name + ".norm.running_mean"] * scale
            self._folded[name] = (self.q(w[name + ".weight"] * scale.view(-1, 1, 1, 1)), shift)
        return self._folded[name]

    def _forced(self, name, y):
        if self.force is None or not self.force.get(name):
            return y
        # (the engine may run the calls of one layer - the RPN's five levels - in another order: first recorded tensor of this geometry)
        lst = self.force[name]
        i = next((k for k, t in enumerate(lst) if tuple(t.shape[2:]) == tuple(y.shape[2:]) and t.shape[0] == y.shape[0]), None)
        assert i is not None, (name, [tuple(t.shape) for t in lst], tuple(y.shape))
        g = lst.pop(i)[:, : y.shape[1]]
        d = (g.double() - y.double()).abs()
        top = float(y.abs().max()) or 1.0
        st = self.forced_stats.setdefault(name, {"n": 0, "differ": 0, "max_rel_to_top": 0.0, "max_ulps": 0.0})
        # units in the last place of the storage type at the element's own magnitude (2^-7 / 2^-10 relative spacing)
        spacing = (2.0 ** -7 if self.sdt == torch.bfloat16 else 2.0 ** -10) * torch.clamp(torch.maximum(g.abs(), y.abs()).double(), min=1e-30)
        st["n"] += d.numel()
        st["differ"] += int((d > 0).sum())
        st["max_rel_to_top"] = max(st["max_rel_to_top"], float(d.max()) / top)
        st["max_ulps"] = max(st["max_ulps"], float((d / spacing).max()))
        return g

    def bnconv(self, x, name, stride=1, padding=0, relu=True, add=None):
        """round(act(conv(x) + shift (+ add)))"""
        wt, shift = self.folded(name)
        y = F.conv2d(x, wt, None, stride=stride, padding=padding) + shift.view(1, -1, 1, 1)
        if add is not None:
            y = y + add
        return self._forced(name, self.q(F.relu(y) if relu else y))

    def bconv(self, x, name, padding=0, dilation=1, relu=False, add=None, out_f32=False):
        """conv with its own bias (or none): round(act(conv(x) + bias (+ add)))"""
        b = self.w.get(name + ".bias")
        y = F.conv2d(x, self.wq(name + ".weight"), None, padding=padding, dilation=dilation)
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
        if add is not None:
            y = y + add
        if relu:
            y = F.relu(y)
        return self._forced(name, y if out_f32 else self.q(y))

    # ---------------------------------------------------------------- rcnn.py:156-181 (stored in the storage type)
    def preprocess(self, image_u8):
        x, padding = super().preprocess(image_u8)
        return self.q(x), padding

    # ---------------------------------------------------------------- resnet.py
    def resnet(self, x):
        from oracle.structure import resnet_blocks
        bu = "backbone.bottom_up."
        x = self.bnconv(x, bu + "stem.conv1", stride=2, padding=3)           # the conv rows are rounded before the pool reads them
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        outs = {}
        for stage, b, cin, cmid, cout, stride, sc in resnet_blocks(self.cfg):
            p = "%s%s.%d." % (bu, stage, b)
            t = self.bnconv(x, p + "conv1", stride=stride)
            t = self.bnconv(t, p + "conv2", padding=1)
            if sc and p in self.fused_shortcuts:
                # out = relu(W3 t2 + Ws x[::s, ::s] + (b3 + bs)), one rounding (pack.dual_source_pointwise)
                w3, s3 = self.folded(p + "conv3")
                ws, ss = self.folded(p + "shortcut")
                y = F.conv2d(t, w3) + F.conv2d(x, ws, stride=stride) + (s3 + ss).view(1, -1, 1, 1)
                x = self._forced(p + "conv3+shortcut", self.q(F.relu(y)))
            else:
                shortcut = self.bnconv(x, p + "shortcut", stride=stride, relu=False) if sc else x
                x = self.bnconv(t, p + "conv3", add=shortcut)
            outs[stage] = x
        return outs

    # ---------------------------------------------------------------- fpn.py:125-166
    def fpn(self, c):
        results = {}
        prev = self.bconv(c["res5"], "backbone.fpn_lateral5")
        results["p5"] = self.bconv(prev, "backbone.fpn_output5", padding=1)
        for lvl in (4, 3, 2):
            top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
            prev = self.bconv(c["res%d" % lvl], "backbone.fpn_lateral%d" % lvl, add=top_down)
            results["p%d" % lvl] = self.bconv(prev, "backbone.fpn_output%d" % lvl, padding=1)
        results["p6"] = F.max_pool2d(results["p5"], kernel_size=1, stride=2, padding=0)
        return results

    # ---------------------------------------------------------------- rpn.py:153-172
    def rpn_head(self, feats):
        logits, deltas = [], []
        pg = "proposal_generator.rpn_head."
        for f in feats:
            t = self.bconv(f, pg + "conv", padding=1, relu=True)             # the hidden tensor, rounded as if it had been stored
            logits.append(self.bconv(t, pg + "objectness_logits", out_f32=True))
            deltas.append(self.bconv(t, pg + "anchor_deltas", out_f32=True))
        return logits, deltas

    # ---------------------------------------------------------------- poolers.py (pooled tensors are stored in the storage type)
    def roi_pool(self, feats, scales, boxes, out_size, sampling):
        return self.q(super().roi_pool(feats, scales, boxes, out_size, sampling))

    # ---------------------------------------------------------------- box_head.py:95-98, fast_rcnn.py
    def box_branch(self, features, proposals, want_all=False):
        # fc1 / fc2 round their ReLU outputs, the predictor's logits stay fp32; everything else as in the fp32 oracle
        cfg = self.cfg
        feats = [features[k] for k in ("p2", "p3", "p4", "p5")]
        scales = [1.0 / s for s in self.strides[:4]]
        pb = proposals["proposal_boxes"]
        x = torch.flatten(self.roi_pool(feats, scales, pb, cfg.box_pool, cfg.box_sampling), start_dim=1)
        for i in range(cfg.box_num_fc):
            n = "roi_heads.box_head.fc%d" % (i + 1)
            x = self.q(F.relu(F.linear(x, self.wq(n + ".weight"), self.w[n + ".bias"])))
        scores = F.linear(x, self.wq("roi_heads.box_predictor.cls_score.weight"), self.w["roi_heads.box_predictor.cls_score.bias"])
        deltas = F.linear(x, self.wq("roi_heads.box_predictor.bbox_pred.weight"), self.w["roi_heads.box_predictor.bbox_pred.bias"])
        return self._select_detections(scores, deltas, proposals)

    def _select_detections(self, scores, deltas, proposals):
        """fast_rcnn.py:257-326, 86-140 on given logits (fp32, as in ref_cpu.OracleModel.box_branch)"""
        from . import ops_ref
        cfg = self.cfg
        pb = proposals["proposal_boxes"]
        boxes = self.apply_deltas(deltas, pb, cfg.bbox_reg_weights)
        probs = F.softmax(scores, dim=-1)
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
        if not bool(valid.all()):
            boxes, probs = boxes[valid], probs[valid]
        sc = probs[:, :-1]
        boxes3 = boxes.reshape(-1, 4).view(-1, 1, 4)
        mask = sc > cfg.score_thresh
        inds = mask.nonzero()
        bsel = boxes3[inds[:, 0], 0]
        ssel = sc[mask]
        keep = ops_ref.batched_nms(bsel.float(), ssel, inds[:, 1], cfg.nms_thresh, self.nms_trick_max_numel)
        if cfg.dets_per_image >= 0:
            keep = keep[: cfg.dets_per_image]
        return {"image_size": proposals["image_size"], "pred_boxes": bsel[keep], "scores": ssel[keep], "pred_classes": inds[keep][:, 1]}

    # ---------------------------------------------------------------- roi_head.py:71-79
    def decoder(self, features):
        from oracle.structure import decoder_layout
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=False)   # noqa: E731
        name = lambda lvl, k: "roi_heads.decoder.%s.%d" % (lvl, 2 * k)                               # noqa: E731
        layout = decoder_layout(self.cfg)
        if self.decoder_fold:
            # x = head(p2) + up(h3) + up(h4) + up(h5) = head(p2) + up(h3 + h4 + h5): the last convolution of every low head adds the
            # running sum of the heads before it AFTER its ReLU, the p2 head adds up(sum) after its ReLU - one rounding each
            low = None
            for lvl, n in layout:
                if lvl == "p2":
                    continue
                t = features[lvl]
                for k in range(n):
                    last = k == n - 1
                    y = F.relu(F.conv2d(t, self.wq(name(lvl, k) + ".weight"), None, padding=1) + self.w[name(lvl, k) + ".bias"].view(1, -1, 1, 1))
                    if last and low is not None:
                        y = y + low
                    t = self._forced(name(lvl, k), self.q(y))
                    if not last:
                        t = self.q(up(t))
                low = t
            y = F.relu(F.conv2d(features["p2"], self.wq(name("p2", 0) + ".weight"), None, padding=1) + self.w[name("p2", 0) + ".bias"].view(1, -1, 1, 1))
            base = self._forced(name("p2", 0), self.q(y + up(low)))
        else:
            # head by head, every intermediate stored; the merge pass sums base + up(h3) + up(h4) + up(h5) in fp32, one rounding
            lows, base = [], None
            for lvl, n in layout:
                t = features[lvl]
                for k in range(n):
                    t = self.bconv(t, name(lvl, k), padding=1, relu=True)
                    if lvl != "p2" and k < n - 1:
                        t = self.q(up(t))
                if lvl == "p2":
                    base = t
                else:
                    lows.append(t)
            y = base
            for t in lows:
                y = y + up(t)
            base = self.q(y)
        return self.bconv(base, "roi_heads.decoder.predictor")

    # ---------------------------------------------------------------- v1convx.py:44-59 / deeplab.py:64-74,105-144
    def _gn_relu(self, t, wname, bname):
        return self._forced("gn:" + wname, self.q(F.relu(F.group_norm(t, 32, self.w[wname], self.w[bname], _EPS))))

    def dp_head(self, x):
        cfg = self.cfg
        hd = "roi_heads.densepose_head."
        if cfg.is_deeplab:
            a = hd + "ASPP."
            res = [self._gn_relu(self.bconv(x, a + "convs.0.0"), a + "convs.0.1.weight", a + "convs.0.1.bias")]
            for i, d in ((1, 6), (2, 12), (3, 56)):
                res.append(self._gn_relu(self.bconv(x, a + "convs.%d.0" % i, padding=d, dilation=d), a + "convs.%d.1.weight" % i, a + "convs.%d.1.bias" % i))
            size = x.shape[-2:]
            t = self.q(F.adaptive_avg_pool2d(x, 1))
            t = self._gn_relu(self.bconv(t, a + "convs.4.1"), a + "convs.4.2.weight", a + "convs.4.2.bias")
            res.append(F.interpolate(t, size=size, mode="bilinear", align_corners=False))    # a 1x1 map: a broadcast, exact
            x = self.bconv(torch.cat(res, dim=1), a + "project.0", relu=True)
        for i in range(cfg.dp_num_convs):
            n = hd + "body_conv_fcn%d" % (i + 1)
            if cfg.is_deeplab:
                x = self._gn_relu(self.bconv(x, n, padding=1), n + ".norm.weight", n + ".norm.bias")
            else:
                x = self.bconv(x, n, padding=1, relu=True)
        return x

    # ---------------------------------------------------------------- chart.py:62-90 (fp32 outputs)
    def dp_predictor(self, x):
        pr = "roi_heads.densepose_predictor."
        outs = []
        for nm in ("ann_index_lowres", "index_uv_lowres", "u_lowres", "v_lowres"):
            t = F.conv_transpose2d(x, self.wq(pr + nm + ".weight"), self.w[pr + nm + ".bias"], stride=2, padding=1)
            outs.append(F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=False))
        return outs
