"""The stages of ``GeneralizedRCNN.inference`` (/root/reference/detectron2/modeling/meta_arch/rcnn.py:110-154) as sequences of C-ABI launches:
preprocess, ResNet + FPN, RPN + proposal selection, box branch, DensePose decoder / head / chart predictor. Mixed into engine.Engine; the
launch glue is engine_ops.LayerOps, streams / graphs / the two phases of a batch step are engine.py."""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .engine_ops import Act
from .weights import decoder_layout, resnet_blocks

FPN_STRIDES = (4, 8, 16, 32, 64)
# torchvision 0.16.2 batched_nms switches from the coordinate-offset trick to the per-class loop above this many box
# ELEMENTS: 4000 where the reference runs on the CPU (what the goldens were recorded with), 20000 in its CUDA mode
# (run.py:22-29). At 800x1333 the RPN feeds 4 x 4819 = 19276 elements: per-level loop on the CPU, trick on CUDA; the two
# differ only where an IoU sits within rounding of the threshold. Engine.nms_reference picks the one to reproduce.
NMS_TRICK_MAX_NUMEL = {"cpu": 4000, "cuda": 20000}


class Stages:
    # ------------------------------------------------------------------ stages
    def preprocess(self, images_u8, Hp, Wp, hwc=False):
        """-> the normalised, zero-padded image in the PAIRED layout the stem consumes ([n, Hp, Wp / 2 + 3, 8]: two 4-channel
        pixels per cell, shifted right by 3 pixels; dp_preprocess_u8 paired=1, pack.stem_paired_conv). hwc: images_u8 is
        [n, h, w, 3] (frames that already have the test size, read as handed over) instead of the resize's planar [n, 3, h, w]."""
        if hwc:
            n, h, w, _ = images_u8.shape
        else:
            n, _, h, w = images_u8.shape
        Wq = Wp // 2 + 3
        out = self._empty((n, Hp, Wq, 8))
        p = L.PreprocessParams()
        p.src, p.dst = images_u8.data_ptr(), out.data_ptr()
        p.paired = 1
        p.src_hwc = 1 if hwc else 0
        p.n_img, p.h, p.w, p.Hp, p.Wp, p.dtype = n, h, w, Hp, Wp, self.dt
        for i in range(3):
            p.mean[i] = self.cfg.pixel_mean[i]
            p.std[i] = self.cfg.pixel_std[i]
        L.check(self.lib.dp_preprocess_u8(C.byref(p), self._stream()), "dp_preprocess_u8")
        return Act(out, n, Hp, Wq, 8)

    def preprocess_frames(self, frames, hwc, x):
        """The same for n <= 64 frames of the test size that live in separate allocations, written into the given paired-layout
        tensor x [n, Hp, Wq, 8] (dp_preprocess_u8_frames): no stacked uint8 copy of the batch in front of the graph."""
        n, Hp, Wq = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        f0 = frames[0]
        h, w = (int(f0.shape[0]), int(f0.shape[1])) if hwc else (int(f0.shape[1]), int(f0.shape[2]))
        assert len(frames) == n and all(f.shape == f0.shape and f.dtype == torch.uint8 and f.is_cuda and f.is_contiguous() for f in frames)
        p = L.PreprocessParams()
        p.src, p.dst, p.paired, p.src_hwc = None, x.data_ptr(), 1, 1 if hwc else 0
        p.n_img, p.h, p.w, p.Hp, p.Wp, p.dtype = n, h, w, Hp, 2 * (Wq - 3), self.dt
        for i in range(3):
            p.mean[i] = self.cfg.pixel_mean[i]
            p.std[i] = self.cfg.pixel_std[i]
        srcs = (C.c_void_p * n)(*[f.data_ptr() for f in frames])
        L.check(self.lib.dp_preprocess_u8_frames(C.byref(p), srcs, n, self._stream()), "dp_preprocess_u8_frames")

    def backbone(self, x):
        Ls = self.model.layers
        cfg = self.cfg
        bu = "backbone.bottom_up."
        with self._stage("backbone.stem"):
            fused = self.stem_pool(Ls["stem"], x) if self.fuse_stem_pool else None
            if fused is not None:
                x = fused       # conv + FrozenBN + ReLU + max-pool in one launch: the conv output is never written
            else:
                x = self.conv(Ls["stem"], x, relu=True, out_hw=(x.H // 2, x.W - 3))   # paired cells in, Hp/2 x Wp/2 pixels out
                Ho, Wo = (x.H + 2 - 3) // 2 + 1, (x.W + 2 - 3) // 2 + 1
                pooled = self._empty((x.N, Ho, Wo, x.C))
                L.check(self.lib.dp_maxpool3x3s2_nhwc(x.t.data_ptr(), pooled.data_ptr(), x.N, x.H, x.W, x.C, self.dt, self._stream()), "maxpool")
                x = Act(pooled, x.N, Ho, Wo, x.C)
        res = {}
        blocks = list(resnet_blocks(cfg))
        t_next = None   # conv1 output of the coming block, when the previous block's fused tail already produced it
        for bi, (stage, b, cin, cmid, cout, stride, sc) in enumerate(blocks):
            p = "%s%s.%d." % (bu, stage, b)
            with self._stage("backbone." + stage):
                fused_sc = Ls.get(p + "conv3+shortcut") if (sc and self.fuse_shortcut and (stride != 1 or self.fuse_sc_tail)) else None
                shortcut = x if (not sc or fused_sc is not None) else self.conv(Ls[p + "shortcut"], x)
                t = t_next if t_next is not None else self.conv(Ls[p + "conv1"], x, relu=True)
                t_next = None
                # conv1 of the next block of the SAME stage (stride 1, reads this block's output) rides in the fused tail
                nxt = blocks[bi + 1] if bi + 1 < len(blocks) and blocks[bi + 1][0] == stage else None
                l1n = Ls["%s%s.%d.conv1" % (bu, nxt[0], nxt[1])] if nxt is not None else None
                fused = None
                if self.fuse_bottleneck and fused_sc is None:
                    fused = self.bottleneck_tail(Ls[p + "conv2"], Ls[p + "conv3"], l1n, t, shortcut)
                elif self.fuse_bottleneck and stride == 1:
                    # first block of res2: conv2 -> conv3 with the projection shortcut as two more K planes, one launch (no next-conv1 stage)
                    fused = self.bottleneck_tail(Ls[p + "conv2"], fused_sc, None, t, None, sc_in=x)
                if fused is not None:
                    x, t_next = fused
                elif fused_sc is not None:
                    # out = relu(W3 t2 + Ws x[::s, ::s] + b3 + bs): the block's input is the second source of conv3's K axis
                    t = self.conv(Ls[p + "conv2"], t, relu=True)
                    x = self.conv(fused_sc, t, relu=True, in2=x)
                else:
                    t = self.conv(Ls[p + "conv2"], t, relu=True)
                    pair = self.bottleneck_pair(Ls[p + "conv3"], l1n, t, shortcut) if not sc else None
                    if pair is not None:       # conv3 + residual + ReLU -> the next block's conv1, the block output written once
                        x, t_next = pair
                    else:
                        x = self.conv(Ls[p + "conv3"], t, relu=True, residual=shortcut)
            res[stage] = x
        feats = {}
        with self._stage("backbone.fpn"):
            # the top-down chain (lateral5 -> lateral4 + up -> ...) is sequential; each level's 3x3 output conv only needs its own
            # merged map, so the small ones (p5, p4, p3) run on forked streams beside the rest of the chain
            prev = self.conv(Ls["fpn_lateral5"], res["res5"])
            outs = []
            for bi, lvl in enumerate((5, 4, 3, 2)):
                if lvl != 5:
                    prev = self.conv(Ls["fpn_lateral%d" % lvl], res["res%d" % lvl], residual=prev, rshift=1)  # + nearest x2 of top-down
                if lvl == 2:
                    feats["p2"] = self.conv(Ls["fpn_output2"], prev)
                    continue
                with self._branch(bi, 1):
                    # the branch READS `prev` on a side stream while the loop rebinds the name: without this the caching allocator
                    # could hand the block to a later main-stream allocation before the side-stream conv has read it
                    prev.t.record_stream(torch.cuda.current_stream(self.device))
                    feats["p%d" % lvl] = self.conv(Ls["fpn_output%d" % lvl], prev)
                    outs.append(feats["p%d" % lvl].t)
                    if lvl == 5:
                        p5 = feats["p5"]
                        H6, W6 = (p5.H - 1) // 2 + 1, (p5.W - 1) // 2 + 1
                        p6 = self._empty((p5.N, H6, W6, p5.C))
                        L.check(self.lib.dp_subsample2_nhwc(p5.t.data_ptr(), p6.data_ptr(), p5.N, p5.H, p5.W, p5.C, self.dt, self._stream()),
                                "subsample2")
                        feats["p6"] = Act(p6, p5.N, H6, W6, p5.C)
                        outs.append(p6)
            self._join(outs)
        return feats

    def rpn(self, feats, Hp, Wp, after_heads=None):
        cfg = self.cfg
        Ls = self.model.layers
        n = feats["p2"].N
        kmax = cfg.rpn_pre_topk
        nl = 5
        slots = nl * kmax
        cand_boxes = self._empty((n, slots, 4), torch.float32)
        cand_scores = self._empty((n, slots), torch.float32)
        cand_level = self._empty((n, slots), torch.int32)
        cand_valid = self._empty((n, slots), torch.int32)
        A = len(cfg.anchor_ratios)
        levels = (L.RpnLevelParams * nl)()
        heads, wss = [], []
        def level_head(f):
            hp = self.model.rpn_head_plain
            if self.rpn_split_min_hw and f.H * f.W >= self.rpn_split_min_hw and self.kernel_class(Ls["rpn_conv"], f) == 10:
                # a large level: class 10 runs the hidden layer 20 % faster than the ring kernel does conv + heads, and the heads read the
                # hidden tensor back in a few microseconds. Chosen by the level's size per image alone (class 10 never looks at the batch)
                t = self.conv(Ls["rpn_conv"], f, relu=True)
                return self.conv(Ls["rpn_head"], t, out_f32=True)
            if hp is not None and self.head_fusable(Ls["rpn_conv"], f):
                # 3x3 conv + ReLU + the two 1x1 heads in one launch: the 256-channel hidden tensor is never written (rpn.py:168-171)
                return self.conv(Ls["rpn_conv"], f, relu=True, head=(hp[0], hp[1], Ls["rpn_head"].macs_per_pixel))
            # (a launch too small for the fused form: the same K order as the fused one - which of the two runs depends on the batch)
            t = self.conv(Ls["rpn_conv"], f, relu=True, ring_order=True)
            return self.conv(Ls["rpn_head"], t, out_f32=True)

        # the five levels are independent (same weights, rpn.py:160-172): p2 on this stream, the small ones beside it
        level_heads = {}
        with self._branch(0, 2):
            level_heads["p3"] = level_head(feats["p3"])
        with self._branch(1, 2):
            for k in ("p4", "p5", "p6"):
                level_heads[k] = level_head(feats[k])
        level_heads["p2"] = level_head(feats["p2"])
        self._join([h.t for h in level_heads.values()])
        for li, k in enumerate(("p2", "p3", "p4", "p5", "p6")):
            f = feats[k]
            head = level_heads[k]
            heads.append(head)
            ws = self._empty((self.lib.dp_rpn_topk_workspace_bytes(n, f.H, f.W, A),), torch.uint8)
            wss.append(ws)
            p = levels[li]
            p.head = head.t.data_ptr()
            p.n_img, p.Hi, p.Wi, p.A, p.head_c = n, f.H, f.W, A, head.C
            p.stride_px = FPN_STRIDES[li]
            for a in range(A):
                for c in range(4):
                    p.cell_anchors[a][c] = self.cell_anchors[li][a][c]
            p.level, p.kmax, p.slot_off, p.slots_per_img = li, kmax, li * kmax, slots
            p.clip_x, p.clip_y = float(Hp), float(Wp)  # Q1: x clipped to the padded HEIGHT, y to the padded WIDTH
            p.cand_boxes, p.cand_scores = cand_boxes.data_ptr(), cand_scores.data_ptr()
            p.cand_level, p.cand_valid = cand_level.data_ptr(), cand_valid.data_ptr()
            p.workspace = ws.data_ptr()
        if after_heads is not None:
            after_heads()       # the caller's side-stream work that is to run beside the selection chain below (see _phase_a)
        # top-k + decode of all five levels in one select launch (one workgroup per image and level)
        L.check(self.lib.dp_rpn_topk_decode_levels(levels, nl, self._stream()), "dp_rpn_topk_decode_levels")
        post = cfg.rpn_post_topk
        props, scores, _, counts = self.nms(cand_boxes, cand_scores, cand_level, cand_valid, n, slots, cfg.rpn_nms_thresh, post)
        if self.keep_intermediates:
            self.inter["rpn_heads"] = heads
            self.inter["cand"] = (cand_boxes, cand_scores, cand_level, cand_valid)
        return props, scores, counts

    def nms(self, boxes, scores, group, valid, n, slots, thr, max_out):
        out_boxes = self._empty((n, max_out, 4), torch.float32)
        out_scores = self._empty((n, max_out), torch.float32)
        out_index = self._empty((n, max_out), torch.int32)
        out_count = self._empty((n,), torch.int32)
        ws = self._empty((self.lib.dp_nms_workspace_bytes(n, slots),), torch.uint8)
        p = L.NmsParams()
        p.boxes, p.scores, p.group, p.valid = boxes.data_ptr(), scores.data_ptr(), group.data_ptr(), valid.data_ptr()
        # torchvision's CPU kernel compares the float IoU with the DOUBLE threshold (nms_kernel.cpp); the float compare of dp_batched_nms is
        # the same predicate when it gets the largest float not above the configured value (0.7 -> 0.699999988; float(0.3) would round up)
        thr32 = np.float32(thr)
        if float(thr32) > float(thr):
            thr32 = np.nextafter(thr32, np.float32(-np.inf))
        p.n_img, p.n_slots, p.iou_thr, p.max_out, p.trick_max_numel = n, slots, float(thr32), max_out, NMS_TRICK_MAX_NUMEL[self.nms_reference]
        p.out_boxes, p.out_scores, p.out_index, p.out_count = out_boxes.data_ptr(), out_scores.data_ptr(), out_index.data_ptr(), out_count.data_ptr()
        p.workspace = ws.data_ptr()
        L.check(self.lib.dp_batched_nms(C.byref(p), self._stream()), "dp_batched_nms")
        return out_boxes, out_scores, out_index, out_count

    def roi_align(self, maps, scales, boxes, counts, n, max_rois, P, sampling, out, compact=False, offsets=None):
        p = L.RoiAlignParams()
        for i, m in enumerate(maps):
            p.feat[i] = m.t.data_ptr()
            p.Hl[i], p.Wl[i], p.scale[i] = m.H, m.W, scales[i]
        p.n_levels, p.min_level = len(maps), 2
        p.C, p.P, p.sampling = maps[0].C, P, sampling
        p.boxes, p.counts, p.n_img, p.max_rois = boxes.data_ptr(), counts.data_ptr(), n, max_rois
        p.out, p.dtype = out.data_ptr(), self.dt
        p.compact = 1 if compact else 0
        p.roi_offsets = offsets.data_ptr() if offsets is not None else None
        L.check(self.lib.dp_roi_align_nhwc(C.byref(p), self._stream()), "dp_roi_align_nhwc")

    def box_branch(self, feats, props, counts):
        cfg = self.cfg
        Ls = self.model.layers
        n = feats["p2"].N
        R = cfg.rpn_post_topk
        P = cfg.box_pool
        maps = [feats[k] for k in ("p2", "p3", "p4", "p5")]
        Cc = maps[0].C
        pooled = self._empty((n * R, P, P, Cc))  # rows >= count are zero-filled by the kernel (finite GEMM input)
        self.roi_align(maps, [1.0 / s for s in FPN_STRIDES[:4]], props, counts, n, R, P, cfg.box_sampling, pooled)
        x = Act(pooled.view(n * R, 1, 1, P * P * Cc), n * R, 1, 1, P * P * Cc)
        x = self.conv(Ls["fc1"], x, relu=True)
        for i in range(1, cfg.box_num_fc):
            x = self.conv(Ls["fc%d" % (i + 1)], x, relu=True)
        logits = self.conv(Ls["box_out"], x, out_f32=True)
        cand_boxes = self._empty((n, R, 4), torch.float32)
        cand_scores = self._empty((n, R), torch.float32)
        cand_group = self._empty((n, R), torch.int32)
        cand_valid = self._empty((n, R), torch.int32)
        p = L.BoxDecodeParams()
        p.logits, p.ld = logits.t.data_ptr(), logits.C
        p.prop_boxes, p.prop_counts, p.n_img, p.max_rois = props.data_ptr(), counts.data_ptr(), n, R
        p.wx, p.wy, p.ww, p.wh = cfg.bbox_reg_weights
        p.score_thresh = cfg.score_thresh
        p.cand_boxes, p.cand_scores, p.cand_group, p.cand_valid = (cand_boxes.data_ptr(), cand_scores.data_ptr(),
                                                                   cand_group.data_ptr(), cand_valid.data_ptr())
        L.check(self.lib.dp_box_decode_score(C.byref(p), self._stream()), "dp_box_decode_score")
        D = max(cfg.dets_per_image, 1)
        det_boxes, det_scores, det_index, det_counts = self.nms(cand_boxes, cand_scores, cand_group, cand_valid, n, R, cfg.nms_thresh, D)
        if self.keep_intermediates:
            self.inter["box_pooled"] = pooled
            self.inter["box_logits"] = logits
        return det_boxes, det_scores, det_counts

    def decoder(self, feats):
        """roi_head.py:71-79: x = head(p2) + head(p3) + head(p4) + head(p5), each head ending in a bilinear x2 except p2's;
        the three final upsamples and the level sum run as ONE pass (dp_merge_upsample2x_nhwc, same fp32 summation order)."""
        Ls = self.model.layers
        layout = decoder_layout(self.cfg)
        # 16-bit modes: the level sum rides in the epilogues of the convolutions that produce its terms. Bilinear up-sampling
        # is linear, so  x = head2 + up(h3) + up(h4) + up(h5) = head2 + up(h3 + h4 + h5):  the last convolution of every low head adds
        # the running sum of the heads before it after its ReLU (post_mode 1), and the p2 head adds up(sum) after ITS ReLU
        # (post_mode 2, taps computed in the kernel). The 756 MB pass of dp_merge_upsample2x_nhwc and its launch disappear;
        # fp32 parity mode keeps the reference's order (head by head, roi_head.py:76-78) with the merge kernel below.
        fold = (self.decoder_fold
                and all(self.post_fusable(Ls["roi_heads.decoder.%s.%d" % (lvl, 2 * (nconv - 1))],
                                          # asked for ONE image: the decision must not depend on the batch size (a frame's
                                          # result is the same whatever else is in the batch), and what fits N = 1 fits any N
                                          Act(None, 1, feats["p2"].H // (1 if lvl == "p2" else 2), feats["p2"].W // (1 if lvl == "p2" else 2), feats["p2"].C),
                                          2 if lvl == "p2" else 1) for lvl, nconv in layout)
                and all(feats[lvl].H * (1 << (nconv - 1)) * 2 == feats["p2"].H and feats[lvl].W * (1 << (nconv - 1)) * 2 == feats["p2"].W
                        for lvl, nconv in layout if lvl != "p2"))
        if self.keep_intermediates:
            self.inter["decoder_fold"] = bool(fold)      # the per-layer choice a storage-emulating oracle has to mirror (tests)
        if fold:
            low_sum = None
            for lvl, nconv in layout:
                if lvl == "p2":
                    continue
                t = feats[lvl]
                for k in range(nconv):
                    last = k == nconv - 1
                    t = self.conv(Ls["roi_heads.decoder.%s.%d" % (lvl, 2 * k)], t, relu=True,
                                  post=low_sum if last else None, post_mode=1 if (last and low_sum is not None) else 0)
                    if not last:
                        up = self._empty((t.N, 2 * t.H, 2 * t.W, t.C))
                        L.check(self.lib.dp_upsample_bilinear2x_nhwc(t.t.data_ptr(), up.data_ptr(), t.N, t.H, t.W, t.C, 0, self.dt,
                                                                     self._stream()), "upsample")
                        t = Act(up, t.N, 2 * t.H, 2 * t.W, t.C)
                low_sum = t
            if self._shared_chip == 2:
                # by the time the p2 head starts (0.7 ms of low-level heads later) the selection chain on the other stream is over
                # and the box head's large launches are running there: no CUs held back any more (+ 0.6 % images/s in A/B)
                self._shared_chip = 1
            base = self.conv(Ls["roi_heads.decoder.p2.0"], feats["p2"], relu=True, post=low_sum, post_mode=2)
            return self.conv(Ls["decoder_predictor"], base)

        def scale_head(lvl, nconv):
            t = feats[lvl]
            for k in range(nconv):
                t = self.conv(Ls["roi_heads.decoder.%s.%d" % (lvl, 2 * k)], t, relu=True)
                if lvl != "p2" and k < nconv - 1:
                    up = self._empty((t.N, 2 * t.H, 2 * t.W, t.C))
                    L.check(self.lib.dp_upsample_bilinear2x_nhwc(t.t.data_ptr(), up.data_ptr(), t.N, t.H, t.W, t.C, 0, self.dt,
                                                                 self._stream()), "upsample")
                    t = Act(up, t.N, 2 * t.H, 2 * t.W, t.C)
            return t

        # the scale heads are independent until the level sum: the three small ones run beside the p2 head
        base, lows = None, []
        for bi, (lvl, nconv) in enumerate(l for l in layout if l[0] != "p2"):
            with self._branch(bi, 4):
                lows.append(scale_head(lvl, nconv))
        for lvl, nconv in layout:
            if lvl == "p2":
                base = scale_head(lvl, nconv)
        self._join([t.t for t in lows])
        for t in lows:
            assert 2 * t.H == base.H and 2 * t.W == base.W and t.C == base.C
        arr = (C.c_void_p * len(lows))(*[t.t.data_ptr() for t in lows])
        L.check(self.lib.dp_merge_upsample2x_nhwc(base.t.data_ptr(), arr, len(lows), base.t.data_ptr(), base.N, lows[0].H, lows[0].W,
                                                  base.C, self.dt, self._stream()), "dp_merge_upsample2x_nhwc")
        return self.conv(Ls["decoder_predictor"], base)

    def groupnorm(self, x_t, R, HW, Cc, c_stride, c_off, gn, relu=True, r_dev=None):
        p = L.GroupNormParams()
        p.r_dev = r_dev.data_ptr() if r_dev is not None else None
        p.x, p.R, p.HW, p.C, p.c_stride, p.c_off, p.groups = x_t.data_ptr(), R, HW, Cc, c_stride, c_off, 32
        p.gamma, p.beta, p.eps, p.relu, p.dtype = gn[0].data_ptr(), gn[1].data_ptr(), 1e-5, 1 if relu else 0, self.dt
        L.check(self.lib.dp_groupnorm_relu_nhwc(C.byref(p), self._stream()), "dp_groupnorm_relu_nhwc")

    def dp_head(self, x, r_dev=None):
        """x: [R slots, P, P, C]; r_dev: int32 device tensor [1] = how many of the slots hold a box (None: all)."""
        cfg = self.cfg
        Ls = self.model.layers
        R, P = x.N, x.H
        rp = r_dev.data_ptr() if r_dev is not None else None
        if cfg.is_deeplab:
            gn = self.model.gn
            Cc = x.C
            cat = self._empty((R, P, P, 5 * Cc))
            for i in range(4):
                self.conv(Ls["aspp%d" % i], x, out=cat, out_c_stride=5 * Cc, out_c_off=i * Cc, n_dev=r_dev)
                self.groupnorm(cat, R, P * P, Cc, 5 * Cc, i * Cc, gn["aspp%d" % i], r_dev=r_dev)
            pooled = self._empty((R, 1, 1, Cc))
            L.check(self.lib.dp_global_avgpool_nhwc(x.t.data_ptr(), pooled.data_ptr(), R, P * P, Cc, self.dt, rp, self._stream()), "gap")
            t = self.conv(Ls["aspp4"], Act(pooled, R, 1, 1, Cc), n_dev=r_dev)
            self.groupnorm(t.t, R, 1, Cc, Cc, 0, gn["aspp4"], r_dev=r_dev)
            L.check(self.lib.dp_broadcast_hw_nhwc(t.t.data_ptr(), cat.data_ptr(), R, P * P, Cc, 5 * Cc, 4 * Cc, self.dt, rp, self._stream()),
                    "broadcast")
            x = self.conv(Ls["aspp_project"], Act(cat, R, P, P, 5 * Cc), relu=True, n_dev=r_dev)
        for i in range(cfg.dp_num_convs):
            if cfg.is_deeplab:
                x = self.conv(Ls["dp_fcn%d" % (i + 1)], x, n_dev=r_dev)
                self.groupnorm(x.t, R, P * P, x.C, x.C, 0, self.model.gn["dp_fcn%d" % (i + 1)], r_dev=r_dev)
            else:
                x = self.conv(Ls["dp_fcn%d" % (i + 1)], x, relu=True, n_dev=r_dev)
        return x

    def dp_predictor(self, x, r_dev=None):
        cfg = self.cfg
        R, P = x.N, x.H
        Ci = self.model.iuv_c
        P2 = 2 * P
        low = self._empty((R, P2, P2, Ci), torch.float32)
        items = list(self.model.deconv.items())
        if self.groups_fusable(items[0][1], x, r_dev):
            # the four parity classes in ONE launch (dp_conv_params.n_groups): same kernel and K order as the four launches, same bits
            self.conv(items[0][1], x, out_f32=True, out=low, out_c_stride=Ci, out_geom=(P2 * P2 * Ci, 2 * P2 * Ci, 2 * Ci, 0), n_dev=r_dev,
                      groups=[(layer, (a * P2 + b) * Ci) for (a, b), layer in items])
            items = []
        for (a, b), layer in items:
            # sub-pixel scatter: output pixel (2i + a, 2j + b)
            self.conv(layer, x, out_f32=True, out=low, out_c_stride=Ci,
                      out_geom=(P2 * P2 * Ci, 2 * P2 * Ci, 2 * Ci, (a * P2 + b) * Ci), n_dev=r_dev)
        S = 2 * P2
        nc, nf = cfg.dp_coarse_ch, cfg.dp_patches + 1
        coarse = self._empty((R, nc, S, S), torch.float32)
        fine = self._empty((R, nf, S, S), torch.float32)
        u = self._empty((R, nf, S, S), torch.float32)
        v = self._empty((R, nf, S, S), torch.float32)
        p = L.IuvParams()
        p.in_, p.R, p.Hs, p.Ws, p.in_c, p.n_coarse, p.n_fine = low.data_ptr(), R, P2, P2, Ci, nc, nf
        p.coarse, p.fine, p.u, p.v = coarse.data_ptr(), fine.data_ptr(), u.data_ptr(), v.data_ptr()
        p.r_dev = r_dev.data_ptr() if r_dev is not None else None
        L.check(self.lib.dp_iuv_upsample_split(C.byref(p), self._stream()), "dp_iuv_upsample_split")
        return coarse, fine, u, v

    def densepose_branch(self, feats, det_boxes, det_counts_dev, dec=None, slots=None):
        """roi_head.py:126-158 for ALL detection slots of the batch, sized on the DEVICE: the launches cover n x D box slots and
        read the live count R = sum(det_counts) from device memory (dp_count_offsets -> dp_conv_params.n_dev / r_dev), so the host
        never waits for R in the middle of a step. Returns tensors with n x D rows (the first R live) + the offsets tensor."""
        cfg = self.cfg
        n = feats["p2"].N
        D = det_boxes.shape[1]
        # Slots: the branch is launched before the host knows R. Round 3 sized everything for n x DETECTIONS_PER_IMAGE slots - with the
        # default 100 per image that is 3.9 MB of fp32 IUV maps per slot, 3.1 GB per step at batch 8 however few boxes there are, and
        # one retained result view pins it all. Now: `slots` (the caller's high-water mark of recent steps, see _dp_slots); the
        # device caps the compact ROI list at that many rows (dp_count_offsets_limited), and the caller - who reads the true
        # counts after the step anyway - runs the branch again with more slots in the rare step that overflowed.
        Rmax = n * D if slots is None else max(1, min(int(slots), n * D))
        offsets = self._empty((n,), torch.int32)
        total = self._empty((1,), torch.int32)
        capped = self._empty((n,), torch.int32)
        L.check(self.lib.dp_count_offsets_limited(det_counts_dev.data_ptr(), n, Rmax, capped.data_ptr(), offsets.data_ptr(), total.data_ptr(),
                                                  self._stream()), "dp_count_offsets_limited")
        det_counts_dev = capped
        if cfg.dp_decoder_on:
            if dec is None:
                with self._stage("decoder"):
                    dec = self.decoder(feats)
            maps, scales = [dec], [1.0 / 4]
            if self.keep_intermediates:
                self.inter["decoder_out"] = dec
        else:
            maps, scales = [feats[k] for k in ("p2", "p3", "p4", "p5")], [1.0 / s for s in FPN_STRIDES[:4]]
        P = cfg.dp_pool
        Cc = maps[0].C
        pooled = self._empty((Rmax, P, P, Cc))
        with self._stage("dp_pool"):
            self.roi_align(maps, scales, det_boxes, det_counts_dev, n, D, P, cfg.dp_sampling, pooled, compact=True, offsets=offsets)
        x = Act(pooled, Rmax, P, P, Cc)
        with self._stage("dp_head"):
            head = self.dp_head(x, total)
        if self.keep_intermediates:
            self.inter["dp_pooled"] = x
            self.inter["dp_head_out"] = head
        with self._stage("dp_predictor"):
            coarse, fine, u, v = self.dp_predictor(head, total)
        return coarse, fine, u, v
