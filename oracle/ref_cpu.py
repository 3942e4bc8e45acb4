"""ORACLE - CPU restatement (torch fp32) of the reference's DensePose inference path.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg. The product package (densepose_torchscript_amd/) never imports it.

The reference is pure Python over torch ATen + torchvision ops; this file restates the forward
path ``DefaultPredictor.forward`` -> ``GeneralizedRCNN.inference`` function by function (citations
are relative to /root/reference), including the bug-compatible quirks Q1-Q6 of SURVEY App. A.
torchvision's two ops are restated in oracle/ops_ref.py.

PINNING: the reference ships no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, run in the build container by
oracle/make_goldens.py (which imports the unmodified /root/reference) and committed under
tests/golden/*.npz; tests/test_oracle_golden.py replays them, tests/test_oracle_vs_reference.py
re-checks live whenever /root/reference is present.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import ops_ref

_SCALE_CLAMP = math.log(1000.0 / 16)  # box_regression.py:11


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class OracleModel:
    def __init__(self, cfg, state, nms_trick_max_numel=4000):
        self.cfg = cfg
        # torchvision batched_nms strategy switch: 4000 box elements where the reference runs on the CPU (default; what the
        # goldens were recorded with), 20000 restates its CUDA mode (run.py:22-29)
        self.nms_trick_max_numel = nms_trick_max_numel
        self.w = OrderedDict((k, _t(v)) for k, v in state.items())
        self.pixel_mean = torch.tensor(cfg.pixel_mean, dtype=torch.float32).view(-1, 1, 1)
        self.pixel_std = torch.tensor(cfg.pixel_std, dtype=torch.float32).view(-1, 1, 1)
        self.strides = (4, 8, 16, 32, 64)
        # anchor_generator.py:181-216 (python float64 math, then float32 tensor)
        self.cell_anchors = []
        for size in cfg.anchor_sizes:
            area = size ** 2.0
            rows = []
            for r in cfg.anchor_ratios:
                w = math.sqrt(area / r)
                h = r * w
                rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
            self.cell_anchors.append(torch.tensor(rows))

    # ---------------------------------------------------------------- layers
    def conv(self, x, name, stride=1, padding=0, dilation=1):
        w = self.w
        return F.conv2d(x, w[name + ".weight"], w.get(name + ".bias"), stride=stride, padding=padding, dilation=dilation)

    def frozen_bn(self, x, name):
        # batch_norm.py:41-63 (eps 1e-5, F.batch_norm eval)
        w = self.w
        return F.batch_norm(x, w[name + ".running_mean"], w[name + ".running_var"], w[name + ".weight"],
                            w[name + ".bias"], training=False, eps=1e-5)

    # ---------------------------------------------------------------- defaults.py:65-97
    def resize(self, img):
        """uint8 [H,W,3] (or [3,H,W]) -> uint8 [3,h,w]; CPU uint8 bilinear, floor size (Q4,Q5)."""
        if img.shape[2] == 3:
            img = img.permute(2, 0, 1)
        else:
            assert img.shape[0] == 3
        height, width = int(img.shape[1]), int(img.shape[2])
        k = min(self.cfg.min_size / min(height, width), self.cfg.max_size / max(height, width))
        image = F.interpolate(img[None], scale_factor=k, mode="bilinear", align_corners=False)[0]
        return image, height, width

    # ---------------------------------------------------------------- rcnn.py:156-181
    def preprocess(self, image_u8):
        x = (image_u8 - self.pixel_mean) / self.pixel_std
        h, w = int(x.shape[1]), int(x.shape[2])
        Hp = (h + 31) // 32 * 32
        Wp = (w + 31) // 32 * 32
        padding = (0, Wp - w, 0, Hp - h)
        return F.pad(x, padding, value=0.0).unsqueeze(0), padding

    # ---------------------------------------------------------------- resnet.py:350-354,189-205,430-453
    def resnet(self, x):
        from oracle.structure import resnet_blocks
        bu = "backbone.bottom_up."
        x = self.conv(x, bu + "stem.conv1", stride=2, padding=3)
        x = F.relu_(self.frozen_bn(x, bu + "stem.conv1.norm"))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        outs = {}
        for stage, b, cin, cmid, cout, stride, sc in resnet_blocks(self.cfg):
            p = "%s%s.%d." % (bu, stage, b)
            out = F.relu_(self.frozen_bn(self.conv(x, p + "conv1", stride=stride), p + "conv1.norm"))  # stride in 1x1
            out = F.relu_(self.frozen_bn(self.conv(out, p + "conv2", padding=1), p + "conv2.norm"))
            out = self.frozen_bn(self.conv(out, p + "conv3"), p + "conv3.norm")
            if sc:
                shortcut = self.frozen_bn(self.conv(x, p + "shortcut", stride=stride), p + "shortcut.norm")
            else:
                shortcut = x
            out = out + shortcut
            x = F.relu_(out)
            outs[stage] = x
        return outs

    # ---------------------------------------------------------------- fpn.py:125-166,198-199
    def fpn(self, c):
        results = {}
        prev = self.conv(c["res5"], "backbone.fpn_lateral5")
        results["p5"] = self.conv(prev, "backbone.fpn_output5", padding=1)
        for lvl in (4, 3, 2):
            top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
            lateral = self.conv(c["res%d" % lvl], "backbone.fpn_lateral%d" % lvl)
            prev = lateral + top_down
            results["p%d" % lvl] = self.conv(prev, "backbone.fpn_output%d" % lvl, padding=1)
        results["p6"] = F.max_pool2d(results["p5"], kernel_size=1, stride=2, padding=0)
        return results

    def backbone(self, images):
        return self.fpn(self.resnet(images))

    # ---------------------------------------------------------------- anchor_generator.py:165-179
    def anchors(self, feats):
        out = []
        for f, stride, base in zip(feats, self.strides, self.cell_anchors):
            gh, gw = int(f.shape[-2]), int(f.shape[-1])
            sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32)
            sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            xx = xx.reshape(-1)
            yy = yy.reshape(-1)
            shifts = torch.stack((xx, yy, xx, yy), dim=1)
            out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
        return out

    # ---------------------------------------------------------------- box_regression.py:74-112
    @staticmethod
    def apply_deltas(deltas, boxes, weights):
        deltas = deltas.float()
        boxes = boxes.to(deltas.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = weights
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = deltas[:, 2::4] / ww
        dh = deltas[:, 3::4] / wh
        dw = torch.clamp(dw, max=_SCALE_CLAMP)
        dh = torch.clamp(dh, max=_SCALE_CLAMP)
        pcx = dx * widths[:, None] + ctr_x[:, None]
        pcy = dy * heights[:, None] + ctr_y[:, None]
        pw = torch.exp(dw) * widths[:, None]
        ph = torch.exp(dh) * heights[:, None]
        x1 = pcx - 0.5 * pw
        y1 = pcy - 0.5 * ph
        x2 = pcx + 0.5 * pw
        y2 = pcy + 0.5 * ph
        return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)

    @staticmethod
    def clip_boxes(boxes, img_size):
        # structures.py:107-112: x clamped to img_size[1], y to img_size[0]
        x1 = boxes[:, 0].clamp(min=0, max=img_size[1])
        y1 = boxes[:, 1].clamp(min=0, max=img_size[0])
        x2 = boxes[:, 2].clamp(min=0, max=img_size[1])
        y2 = boxes[:, 3].clamp(min=0, max=img_size[0])
        return torch.stack((x1, y1, x2, y2), dim=-1)

    @staticmethod
    def nonempty(boxes, thr=0.0):
        return ((boxes[:, 2] - boxes[:, 0]) >= thr) & ((boxes[:, 3] - boxes[:, 1]) >= thr)

    # ---------------------------------------------------------------- rpn.py:153-172,300-394 + proposal_utils.py:19-134
    def rpn_head(self, feats):
        logits, deltas = [], []
        pg = "proposal_generator.rpn_head."
        for f in feats:
            t = F.relu(self.conv(f, pg + "conv", padding=1))
            logits.append(self.conv(t, pg + "objectness_logits"))
            deltas.append(self.conv(t, pg + "anchor_deltas"))
        return logits, deltas

    def rpn(self, images, features, want_all=False):
        cfg = self.cfg
        feats = [features[k] for k in ("p2", "p3", "p4", "p5", "p6")]
        anchors = self.anchors(feats)
        logits, deltas = self.rpn_head(feats)
        flat_logits = [s.permute(0, 2, 3, 1).flatten(1) for s in logits]
        flat_deltas = [x.view(x.shape[0], -1, 4, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2)
                       for x in deltas]
        # Q1: image_sizes = [W_pad, H_pad] (rpn.py:339) but clip_boxes reads it as (h, w)
        image_size = torch.tensor([images.shape[3], images.shape[2]], dtype=torch.int64)
        props = [self.apply_deltas(d.reshape(-1, 4), a, (1.0, 1.0, 1.0, 1.0)).view(1, -1, 4)
                 for a, d in zip(anchors, flat_deltas)]
        tk_scores, tk_props, lvl_ids = [], [], []
        for lvl, (p, l) in enumerate(zip(props, flat_logits)):
            n = min(int(l.shape[1]), cfg.rpn_pre_topk)
            sc, idx = l.topk(n, dim=1)
            tk_scores.append(sc)
            tk_props.append(p[0][idx[0]][None])
            lvl_ids.append(torch.full((n,), lvl, dtype=torch.int64))
        scores = torch.cat(tk_scores, dim=1)[0]
        boxes = torch.cat(tk_props, dim=1)[0]
        lvl = torch.cat(lvl_ids, dim=0)
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
        if not bool(valid.all()):
            boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
        boxes = self.clip_boxes(boxes, image_size)
        keep = self.nonempty(boxes, 0.0)
        if int(keep.sum()) != len(boxes):
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = ops_ref.batched_nms(boxes.float(), scores, lvl, cfg.rpn_nms_thresh, self.nms_trick_max_numel)
        keep = keep[: cfg.rpn_post_topk]
        res = {"image_size": image_size, "proposal_boxes": boxes[keep], "objectness_logits": scores[keep]}
        if want_all:
            res["rpn_logits"] = flat_logits
            res["rpn_deltas"] = flat_deltas
            res["pre_nms_boxes"] = boxes
            res["pre_nms_scores"] = scores
            res["pre_nms_levels"] = lvl
        return res

    # ---------------------------------------------------------------- poolers.py:15-51,187-227
    @staticmethod
    def assign_levels(boxes, min_level, max_level, canonical_box_size=224, canonical_level=4):
        area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        sizes = torch.sqrt(area)
        lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
        lv = torch.clamp(lv, min=min_level, max=max_level)
        return lv.to(torch.int64) - min_level

    def roi_pool(self, feats, scales, boxes, out_size, sampling):
        rois = torch.cat([torch.zeros((boxes.shape[0], 1), dtype=boxes.dtype), boxes], dim=1)
        if len(feats) == 1:
            return ops_ref.roi_align(feats[0], rois.to(feats[0].dtype), out_size, scales[0], sampling, False)
        min_level = int(-math.log2(scales[0]))
        max_level = int(-math.log2(scales[-1]))
        la = self.assign_levels(boxes, min_level, max_level)
        C = feats[0].shape[1]
        out = torch.zeros((boxes.shape[0], C, out_size, out_size), dtype=feats[0].dtype)
        for lvl in range(len(feats)):
            inds = torch.nonzero(la == lvl)[:, 0]
            out[inds] = ops_ref.roi_align(feats[lvl], rois[inds].to(feats[lvl].dtype), out_size, scales[lvl], sampling, False)
        return out

    # ---------------------------------------------------------------- box_head.py:95-98, fast_rcnn.py:238-326,86-140
    def box_branch(self, features, proposals, want_all=False):
        cfg = self.cfg
        feats = [features[k] for k in ("p2", "p3", "p4", "p5")]
        scales = [1.0 / s for s in self.strides[:4]]
        pb = proposals["proposal_boxes"]
        x = self.roi_pool(feats, scales, pb, cfg.box_pool, cfg.box_sampling)
        pooled = x
        x = torch.flatten(x, start_dim=1)
        for i in range(cfg.box_num_fc):
            n = "roi_heads.box_head.fc%d" % (i + 1)
            x = F.relu(F.linear(x, self.w[n + ".weight"], self.w[n + ".bias"]))
        scores = F.linear(x, self.w["roi_heads.box_predictor.cls_score.weight"], self.w["roi_heads.box_predictor.cls_score.bias"])
        deltas = F.linear(x, self.w["roi_heads.box_predictor.bbox_pred.weight"], self.w["roi_heads.box_predictor.bbox_pred.bias"])
        boxes = self.apply_deltas(deltas, pb, cfg.bbox_reg_weights)
        probs = F.softmax(scores, dim=-1)
        # fast_rcnn_inference_single_image
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
        if not bool(valid.all()):
            boxes, probs = boxes[valid], probs[valid]
        sc = probs[:, :-1]
        boxes3 = boxes.reshape(-1, 4).view(-1, 1, 4)  # Q2: clip result discarded -> no clip
        mask = sc > cfg.score_thresh
        inds = mask.nonzero()
        bsel = boxes3[inds[:, 0], 0]
        ssel = sc[mask]
        keep = ops_ref.batched_nms(bsel.float(), ssel, inds[:, 1], cfg.nms_thresh, self.nms_trick_max_numel)
        if cfg.dets_per_image >= 0:
            keep = keep[: cfg.dets_per_image]
        res = {"image_size": proposals["image_size"], "pred_boxes": bsel[keep], "scores": ssel[keep],
               "pred_classes": inds[keep][:, 1]}
        if want_all:
            res["box_pooled"] = pooled
            res["box_logits"] = scores
            res["box_deltas"] = deltas
            res["box_decoded"] = boxes
            res["box_probs"] = probs
        return res

    # ---------------------------------------------------------------- roi_head.py:71-79 (Decoder)
    def decoder(self, features):
        from oracle.structure import decoder_layout
        x = None
        for lvl, n in decoder_layout(self.cfg):
            t = features[lvl]
            for k in range(n):
                t = F.relu(self.conv(t, "roi_heads.decoder.%s.%d" % (lvl, 2 * k), padding=1))
                if lvl != "p2":
                    t = F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=False)
            x = t if x is None else x + t
        return self.conv(x, "roi_heads.decoder.predictor")

    # ---------------------------------------------------------------- v1convx.py:44-59 / deeplab.py:64-74,105-144
    def dp_head(self, x):
        cfg = self.cfg
        hd = "roi_heads.densepose_head."
        if cfg.is_deeplab:
            a = hd + "ASPP."
            res = []
            t = self.conv(x, a + "convs.0.0")
            res.append(F.relu(F.group_norm(t, 32, self.w[a + "convs.0.1.weight"], self.w[a + "convs.0.1.bias"], 1e-5)))
            for i, d in ((1, 6), (2, 12), (3, 56)):
                t = self.conv(x, a + "convs.%d.0" % i, padding=d, dilation=d)
                res.append(F.relu(F.group_norm(t, 32, self.w[a + "convs.%d.1.weight" % i], self.w[a + "convs.%d.1.bias" % i], 1e-5)))
            size = x.shape[-2:]
            t = F.adaptive_avg_pool2d(x, 1)
            t = self.conv(t, a + "convs.4.1")
            t = F.relu(F.group_norm(t, 32, self.w[a + "convs.4.2.weight"], self.w[a + "convs.4.2.bias"], 1e-5))
            res.append(F.interpolate(t, size=size, mode="bilinear", align_corners=False))
            x = F.relu(self.conv(torch.cat(res, dim=1), a + "project.0"))
        for i in range(cfg.dp_num_convs):
            n = hd + "body_conv_fcn%d" % (i + 1)
            x = self.conv(x, n, padding=1)
            if cfg.is_deeplab:
                x = F.group_norm(x, 32, self.w[n + ".norm.weight"], self.w[n + ".norm.bias"], 1e-5)
            x = F.relu(x)
        return x

    # ---------------------------------------------------------------- chart.py:62-90
    def dp_predictor(self, x):
        pr = "roi_heads.densepose_predictor."
        outs = []
        for nm in ("ann_index_lowres", "index_uv_lowres", "u_lowres", "v_lowres"):
            t = F.conv_transpose2d(x, self.w[pr + nm + ".weight"], self.w[pr + nm + ".bias"], stride=2, padding=1)
            outs.append(F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=False))
        return outs

    def densepose_branch(self, features, pred_boxes, want_all=False):
        cfg = self.cfg
        extra = {}
        if cfg.dp_decoder_on:
            dec = self.decoder(features)
            feats, scales = [dec], [1.0 / 4]
            extra["decoder_out"] = dec
        else:
            feats = [features[k] for k in ("p2", "p3", "p4", "p5")]
            scales = [1.0 / s for s in self.strides[:4]]
        pooled = self.roi_pool(feats, scales, pred_boxes, cfg.dp_pool, cfg.dp_sampling)
        head = self.dp_head(pooled)
        outs = self.dp_predictor(head)
        if want_all:
            extra["dp_pooled"] = pooled
            extra["dp_head_out"] = head
        return outs, extra

    # ---------------------------------------------------------------- postprocessing.py:11-61
    def postprocess(self, res, dp, height, width, padding):
        image_size = res["image_size"]
        ow = torch.tensor(width, dtype=torch.int64)
        oh = torch.tensor(height, dtype=torch.int64)
        scale_x = ow.float() / (image_size[0] - padding[0] - padding[1])
        scale_y = oh.float() / (image_size[1] - padding[2] - padding[3])
        boxes = res["pred_boxes"].clone()
        boxes[:, 0] *= scale_x
        boxes[:, 1] *= scale_y
        boxes[:, 2] *= scale_x
        boxes[:, 3] *= scale_y
        keep = self.nonempty(boxes)
        new_size = torch.stack([oh, ow])
        return {
            "image_size": new_size,
            "pred_boxes": self.clip_boxes(boxes[keep], new_size),
            "scores": res["scores"][keep],
            "pred_classes": res["pred_classes"][keep],
            "pred_densepose_coarse_segm": dp[0][keep],
            "pred_densepose_fine_segm": dp[1][keep],
            "pred_densepose_u": dp[2][keep],
            "pred_densepose_v": dp[3][keep],
        }

    # ---------------------------------------------------------------- whole path
    @torch.no_grad()
    def forward(self, original_image, want_all=False):
        image, height, width = self.resize(original_image)
        images, padding = self.preprocess(image)
        features = self.backbone(images)
        proposals = self.rpn(images, features, want_all)
        dets = self.box_branch(features, proposals, want_all)
        dp, extra = self.densepose_branch(features, dets["pred_boxes"], want_all)
        out = self.postprocess(dets, dp, height, width, padding)
        if want_all:
            inter = {"resized": image, "images": images, "padding": padding}
            inter.update(features)
            inter.update({k: v for k, v in proposals.items()})
            inter.update({("det_" + k if k in ("image_size", "pred_boxes", "scores", "pred_classes") else k): v
                          for k, v in dets.items()})
            inter.update(extra)
            inter.update({"dp_raw_%d" % i: t for i, t in enumerate(dp)})
            return out, inter
        return out

    __call__ = forward

    @torch.no_grad()
    def densepose_given_boxes(self, features, boxes):
        """Stage oracle: IUV maps for injected boxes (roi_head.py:160-184 forward_with_given_boxes)."""
        return self.densepose_branch(features, boxes)[0]


# ---------------------------------------------------------------- visualizer.py:10-56 (the "part-index argmax")
def resample_fine(coarse, fine, w, h):
    coarse_b = F.interpolate(coarse, (h, w), mode="bilinear", align_corners=False).argmax(dim=1)
    labels = F.interpolate(fine, (h, w), mode="bilinear", align_corners=False).argmax(dim=1) * (coarse_b > 0).long()
    return labels


def resample_uv(u, v, labels, w, h):
    u_b = F.interpolate(u, (h, w), mode="bilinear", align_corners=False)
    v_b = F.interpolate(v, (h, w), mode="bilinear", align_corners=False)
    uv = torch.zeros([2, h, w], dtype=torch.float32)
    for part_id in range(1, u_b.size(1)):
        uv[0][labels == part_id] = u_b[0, part_id][labels == part_id]
        uv[1][labels == part_id] = v_b[0, part_id][labels == part_id]
    return uv


def extract_iuv(out):
    """-> list of (labels int64 [h,w], uv f32 [2,h,w]) per detection (visualizer.py:33-56)."""
    boxes = out["pred_boxes"].clone()
    boxes[:, 2] -= boxes[:, 0]
    boxes[:, 3] -= boxes[:, 1]
    res = []
    for i in range(len(boxes)):
        x, y, w, h = [int(t) for t in boxes[i].long().tolist()]
        w = max(w, 1)
        h = max(h, 1)
        labels = resample_fine(out["pred_densepose_coarse_segm"][i:i + 1], out["pred_densepose_fine_segm"][i:i + 1], w, h)[0]
        uv = resample_uv(out["pred_densepose_u"][i:i + 1], out["pred_densepose_v"][i:i + 1], labels, w, h)
        res.append((labels, uv))
    return res
