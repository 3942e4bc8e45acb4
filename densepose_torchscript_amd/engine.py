"""MI355X DensePose inference engine: orchestrates the C-ABI HIP kernels over a batch of equal-size frames.

Mirrors ``GeneralizedRCNN.inference`` (/root/reference/detectron2/modeling/meta_arch/rcnn.py:110-154) stage
by stage; every arithmetic step is a call into libdensepose_hip.so (no torch compute ops on the hot path -
torch only owns device memory and the stream). Batch semantics follow SURVEY Q6: a batch of N frames
equals N independent single-image calls (per-image top-k / NMS / padding).
"""
import contextlib
import ctypes as C
import math

import numpy as np
import torch

from . import lib as L
from .engine_ops import Act, LayerOps  # noqa: F401  (Act is part of this module's interface)
from .engine_stages import FPN_STRIDES, NMS_TRICK_MAX_NUMEL, Stages  # noqa: F401
from .options import EngineOptions
from .pack import PackedModel, round_up

MAX_GRAPHS = 8   # captured HIP graphs kept per engine (each pins the activations of its shape): least recently used is dropped;
                 # raised to the number of live (stream slot, pipeline lane) pairs when that is larger (Engine._graph_cap)

_engine_device = None   # the one device this process drives (one process per GPU: DESIGN.md §5)


def _ptr(t):
    return t.data_ptr() if t is not None else None


class Engine(LayerOps, Stages):
    def __init__(self, cfg, state, dtype="bf16", device="cuda:0", options=None):
        if not torch.cuda.is_available():
            raise L.DensePoseHipError("no GPU visible: the DensePose engine has no CPU fallback")
        self.lib = L.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # One process drives ONE GPU (frames shard over processes, parallel.py): the library caches per-kernel attributes and
        # the CU count per process, and events / graph capture follow the current device. A second engine on another device
        # in the same process would silently read stale detection counts - refuse it instead.
        global _engine_device
        if _engine_device is None:
            _engine_device = self.device
        elif _engine_device != self.device:
            raise L.DensePoseHipError("this process already drives %s; start one process per GPU (parallel.launch_local_ranks) "
                                      "instead of a second engine on %s" % (_engine_device, self.device))
        torch.cuda.set_device(self.device)
        if dtype not in L.DTYPES:
            raise ValueError("dtype must be one of %s" % sorted(L.DTYPES))
        self.dt = L.DTYPES[dtype]
        self.tdt = {L.DP_F32: torch.float32, L.DP_BF16: torch.bfloat16, L.DP_F16: torch.float16}[self.dt]
        self.model = PackedModel(cfg, state, self.dt, self.device)
        self.cell_anchors = []
        for size in cfg.anchor_sizes:  # anchor_generator.py:181-216 (python float64, then fp32)
            rows = []
            for r in cfg.anchor_ratios:
                w = math.sqrt(size ** 2.0 / r)
                h = r * w
                rows.append([np.float32(-w / 2.0), np.float32(-h / 2.0), np.float32(w / 2.0), np.float32(h / 2.0)])
            self.cell_anchors.append(rows)
        self.trace = None   # trace.StageTrace: optional per-stage event timers / roctx ranges (off by default)
        self.keep_intermediates = False
        self.inter = {}
        self.flops_last = 0
        self.prof = None  # list of (kernel class, algorithmic flops, start event, end event) when profiling
        self.use_graphs = False
        self.nms_reference = "cpu"    # "cpu" | "cuda": which torchvision batched_nms strategy switch to reproduce (see above)
        # independent per-level layers on forked streams, bit mask: 1 FPN output convs, 2 RPN levels, 4 decoder scale heads.
        # Round 3 (bench.py, 2 runs each, same box): none 889 / 892 img/s, FPN 896 / 897, RPN 915 / 914, FPN + RPN 915 / 903 - the RPN
        # levels were forked from then on. End of round 6 (two pipeline lanes, faster kernels): in line 1178 - 1187 against 1152 - 1165
        # forked, R_101 958 against 913 - 922 (profiles/r6_ab_fork.txt): the default is 0 again. The decoder's heads fork from a stream
        # that is itself a fork, which hipGraph capture does not survive (segfault in capture_end on ROCm 7.2): bit 4 only outside capture
        opt = options if options is not None else EngineOptions()     # the A/B switches (options.py); the product path reads no environment
        self.options = opt
        self.fork_levels = int(opt.fork_levels)
        self.frames_direct = bool(opt.frames_direct)
        self._forked = {}
        self.fuse_stem_pool = True    # stem conv + ReLU + max-pool in one launch (dp_stem_pool_nhwc)
        self.fuse_rpn_head = True     # RPN 3x3 conv + 1x1 heads in one launch where the 256-cout ring kernel runs the level
        self.fuse_bottleneck = True   # res2 blocks: conv2 -> conv3 -> next conv1 in one launch (bottleneck_tail)
        self.overlap_decoder = True   # decoder on a side stream beside the RPN / box branch (see _phase_a)
        self.fuse_shortcut = bool(opt.fuse_shortcut)   # block-0 projection shortcut as K planes of conv3 (16-bit modes)
        self.fuse_sc_tail = bool(opt.fuse_sc_tail)     # ... of res2.0 too (stride 1: inside the fused bottleneck tail)
        self.fuse_pair = bool(opt.fuse_pair)   # A/B knob: 0 = conv3 and the next block's conv1 of res3's plain blocks as two launches
        self.group_deconv = bool(opt.group_deconv)   # A/B knob: 0 = the predictor's four sub-pixel convolutions as four launches
        self.split_k_on = bool(opt.split_k_on)   # A/B knob: layers with PackedConv.split_k run unsplit
        self.decoder_fold = bool(opt.decoder_fold)      # 16-bit modes: the decoder's level sum in the conv epilogues (post_res) instead of a merge pass
        self.rpn_split_min_hw = int(opt.rpn_split_min_hw)   # RPN levels this large: hidden layer on class 10, heads as a second launch
        self._shared_chip = 0         # dp_conv_params.shared_chip of the launches being issued: 1 beside other large launches, 2 beside the top-k / NMS chain
        self.decoder_after_rpn_heads = bool(opt.decoder_after_rpn_heads)   # where the decoder's side stream forks (see _phase_a)
        self._side_streams = {}
        self._stream_handles = set()   # HIP streams in use by this engine and its predictor (new_stream)
        self._capture_stream = None
        self._graphs = {}
        self._graph_slots, self._graph_captures = set(), {}
        self._meta_cache = {}
        self._pinned = {}
        self._r_hwm = {}              # (frames, slots per frame) -> high-water mark of the detections of recent batches (_dp_slots)

    # ------------------------------------------------------------------ helpers
    def _stage(self, name):
        """bracket of one stage of the path for the optional tracer (trace.py); a no-op context when tracing is off"""
        return self.trace.stage(name, self) if self.trace is not None else contextlib.nullcontext()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def new_stream(self):
        """A HIP stream that is none of the streams this engine already uses and not the current one. torch.cuda.Stream() hands
        out 32 pooled streams round-robin, whoever asks: after enough predictors have lived in a process two logical streams of
        one engine - a fork stream and the stream being captured, say - can be the SAME HIP stream, the fork / join events then
        wait on their own stream, and hipGraphLaunch of the graph captured that way crashed inside the runtime
        (hip::Graph::UpdateStreams; tools/ history: seventh predictor of a pytest process)."""
        taken = self._stream_handles | {torch.cuda.current_stream(self.device).cuda_stream}
        for _ in range(64):
            s = torch.cuda.Stream(device=self.device)
            if s.cuda_stream not in taken:
                break
        self._stream_handles.add(s.cuda_stream)
        return s

    @contextlib.contextmanager
    def _branch(self, i, group=1):
        """Run the enclosed launches on side stream `i` of the current stream, forked from everything launched on it so far:
        independent small layers (FPN output convs, RPN levels, decoder scale heads) that cannot fill the chip alone run beside
        each other. The caller joins with _join() before anything reads their results; tensors a branch allocates and hands
        back must be passed to _join(outputs=...). With `fork_levels` off (or while profiling) the body runs in line."""
        if not (self.fork_levels & group) or self.prof is not None or self.trace is not None or (
                group == 4 and torch.cuda.is_current_stream_capturing()):
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        pool = self._side_streams.setdefault(("fork", cur.cuda_stream), [])
        while len(pool) <= i:
            pool.append(self.new_stream())
        s = pool[i]
        s.wait_stream(cur)
        self._forked.setdefault(cur.cuda_stream, set()).add(i)
        shared, self._shared_chip = self._shared_chip, (self._shared_chip or 1)    # launches of a branch run beside the main chain (dp_conv_params.shared_chip)
        try:
            with torch.cuda.stream(s):
                yield
        finally:
            self._shared_chip = shared

    def _join(self, outputs=()):
        cur = torch.cuda.current_stream(self.device)
        pool = self._side_streams.get(("fork", cur.cuda_stream), [])
        for i in sorted(self._forked.pop(cur.cuda_stream, ())):
            cur.wait_stream(pool[i])
        for t in outputs:
            if t is not None and t.is_cuda:
                t.record_stream(cur)


    # ------------------------------------------------------------------ whole path for a batch of equal-size frames
    def _phase_a(self, images_u8, given_boxes=None, hwc=False):
        """preprocess -> backbone -> RPN -> box head -> detection select (everything whose launch sizes are static)."""
        from .resize import FusedResize
        if isinstance(images_u8, tuple):
            # ("x", paired-layout tensor, h, w): already resized + preprocessed (dp_resize_preprocess_u8_batch wrote it)
            _, xt, h, w = images_u8
            n = int(xt.shape[0])
            Hp, Wp = round_up(h, 32), round_up(w, 32)
            x = Act(xt, n, Hp, Wp // 2 + 3, 8)
        elif isinstance(images_u8, FusedResize):
            n, _, h, w = images_u8.shape
            Hp, Wp = round_up(h, 32), round_up(w, 32)
            with self._stage("preprocess"):
                xt = self._empty((n, Hp, Wp // 2 + 3, 8))
                images_u8.run(self, xt)
            x = Act(xt, n, Hp, Wp // 2 + 3, 8)
        else:
            assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[3 if hwc else 1] == 3
            images_u8 = images_u8.contiguous()
            if hwc:
                n, h, w, _ = images_u8.shape
            else:
                n, _, h, w = images_u8.shape
            Hp, Wp = round_up(h, 32), round_up(w, 32)
            with self._stage("preprocess"):
                x = self.preprocess(images_u8, Hp, Wp, hwc)
        feats = self.backbone(x)
        if self.keep_intermediates:
            self.inter.update(feats)
        # The decoder (roi_head.py:42-79) reads only the FPN maps, not the detections: it runs on a side stream beside the
        # proposal top-k / NMS chain and the box branch. The fork point is AFTER the RPN's head convolutions: those fill the chip
        # on their own (beside the decoder both only get slower), the chain behind them is a string of one-workgroup-per-image
        # launches that leaves the chip empty. The decoder's persistent launches leave an eighth of the CUs to that chain
        # (shared_chip = 2): + 2.4 % images/s, - 7 % single-frame time against forking at the FPN's end (same-box A/B).
        dec, side = None, None
        launch_decoder = None
        if self.cfg.dp_decoder_on and self.overlap_decoder:
            cur = torch.cuda.current_stream(self.device)
            side = self._side_streams.get(cur.cuda_stream)
            if side is None:
                side = self._side_streams[cur.cuda_stream] = self.new_stream()
            late = self.decoder_after_rpn_heads and given_boxes is None

            def launch_decoder():
                nonlocal dec
                side.wait_stream(cur)
                self._shared_chip = 2 if late else 1
                with torch.cuda.stream(side), self._stage("decoder"):
                    dec = self.decoder(feats)
                self._shared_chip = 1     # from here to the join the launches of this stream share the chip with the decoder's
            if not late:
                launch_decoder()
                launch_decoder = None
        if given_boxes is None:
            with self._stage("rpn"):
                props, prop_scores, prop_counts = self.rpn(feats, Hp, Wp, after_heads=launch_decoder)
            if self.keep_intermediates:
                self.inter["proposals"] = (props, prop_scores, prop_counts)
            with self._stage("box_head"):
                det_boxes, det_scores, det_counts = self.box_branch(feats, props, prop_counts)
        else:
            det_boxes, det_scores, det_counts = given_boxes
        if side is not None:
            cur.wait_stream(side)
            dec.t.record_stream(cur)
            self._shared_chip = 0
        return dict(n=n, h=h, w=w, feats=feats, det_boxes=det_boxes, det_scores=det_scores, det_counts=det_counts, dec=dec)

    def _dp_slots(self, n, D, seen=None):
        """Slot count of the DensePose branch for a batch of n frames: a high-water mark of the box counts of recent batches of that
        size (+ 25 %, rounded up to 16, at least 16 per frame until a count has been seen), never more than n x D. With `seen`:
        record a batch's true count (the mark decays by 2 % per batch, so one crowded scene does not size the next hour)."""
        key = (n, D)
        hwm = self._r_hwm.get(key)
        if seen is not None:
            self._r_hwm[key] = float(seen) if hwm is None else max(float(seen), 0.98 * hwm)
            return None
        want = 16 * n if hwm is None else int(hwm * 1.25) + n
        return max(1, min(n * D, (want + 15) // 16 * 16))

    def _pinned_counts(self, key, n):
        buf = self._pinned.get(key)
        if buf is None or buf.numel() < n:
            buf = torch.empty((max(n, 8),), dtype=torch.int32, pin_memory=True)  # cudaHostAlloc is slow: allocate once
            self._pinned[key] = buf
        return buf[:n]

    def _phase_a_run(self, images_u8, slot, given_boxes=None, hwc=False):
        """Phase A + asynchronous read-back of the detection counts. With ``use_graphs`` the launch sequence of a given
        (sub-batch shape, stream slot) is captured once into a HIP graph and replayed: the ~200 kernel launches of the
        static part cost one graph launch on the host instead of ~200 x (ctypes call + hipLaunchKernel)."""
        from .resize import FusedResize
        frames = None
        fused = images_u8 if isinstance(images_u8, FusedResize) else None
        if fused is not None:
            shape = ("fused",) + tuple(fused.shape)
        elif isinstance(images_u8, (list, tuple)):      # separate same-size device frames: read where they are by the preprocess launch
            frames, shape = images_u8, (len(images_u8),) + tuple(images_u8[0].shape)
        else:
            shape = tuple(images_u8.shape)
        n = shape[1] if fused is not None else shape[0]
        graphable = self.use_graphs and given_boxes is None and not self.keep_intermediates and self.prof is None and self.trace is None
        frames_direct = graphable and frames is not None and len(frames) <= 64 and self.frames_direct and all(f.is_contiguous() for f in frames)
        if frames_direct:
            shape = ("frames",) + shape
        if not graphable:
            if frames is not None:
                images_u8 = torch.stack(frames)
            st = self._phase_a(images_u8, given_boxes, hwc)
            pinned = self._pinned_counts(("eager", slot), n)
            pinned.copy_(st["det_counts"], non_blocking=True)
        else:
            # every switch that changes the captured launch sequence is part of the key: toggling one on a live engine captures anew
            key = (shape, hwc, slot, self.overlap_decoder, self.decoder_after_rpn_heads, self.fuse_bottleneck, self.fuse_rpn_head,
                   self.fuse_stem_pool, self.fork_levels, self.nms_reference, self.decoder_fold, self.fuse_shortcut, self.fuse_sc_tail,
                   self.fuse_pair, self.group_deconv, self.split_k_on, self.rpn_split_min_hw)
            entry = self._graphs.pop(key, None)
            if entry is None:
                # every (stream slot / pipeline lane) of one geometry needs a graph of its own: never cap below the slots in use,
                # and say so when the same key keeps being re-captured (a recapture costs a device synchronise + an eager run)
                self._graph_slots.add(slot)
                ncap = self._graph_captures[key] = self._graph_captures.get(key, 0) + 1
                if ncap == 3:
                    import warnings
                    warnings.warn("HIP graph of %s captured %d times: more live (geometry, slot) pairs than the graph cache holds - "
                                  "run such a stream eagerly (use_graphs=False) or with fewer streams" % (key[:2], ncap))
                while len(self._graphs) >= max(MAX_GRAPHS, 2 * len(self._graph_slots)):      # drop the least recently used graph and its memory pool
                    torch.cuda.synchronize(self.device)     # (rare: a new input geometry) its last replay may still be running
                    old = self._graphs.pop(next(iter(self._graphs)))
                    self._pinned.pop(old[5], None)
                    del old
                if fused is not None:
                    # the graph starts BEHIND the fused resize + preprocess (the frame pointers change every step): its static input is
                    # the paired-layout tensor that launch writes
                    _, _, fh, fw = fused.shape
                    xs = self._empty((n, round_up(fh, 32), round_up(fw, 32) // 2 + 3, 8))
                    fused.run(self, xs)
                    static_in = ("x", xs, fh, fw)
                elif frames_direct:
                    # ... and behind the preprocess launch that reads the separate frames of a batch where they are (their addresses
                    # change every step; a stacked uint8 copy of the batch would cost 3 h w n bytes of traffic per step)
                    fh, fw = (shape[2], shape[3]) if hwc else (shape[3], shape[4])      # shape = ("frames", n) + one frame's shape
                    xs = self._empty((n, round_up(fh, 32), round_up(fw, 32) // 2 + 3, 8))
                    self.preprocess_frames(frames, hwc, xs)
                    static_in = ("x", xs, fh, fw)
                else:
                    static_in = torch.stack(frames) if frames is not None else images_u8.clone()
                self._phase_a(static_in, None, hwc)   # eager warm-up: one-time attribute / table initialisation outside capture
                torch.cuda.current_stream(self.device).synchronize()
                pinned = self._pinned_counts(key, n)
                graph = torch.cuda.CUDAGraph()
                flops0 = self.flops_last
                if self._capture_stream is None:
                    self._capture_stream = self.new_stream()    # not torch's process-wide default capture stream: see new_stream
                with torch.cuda.graph(graph, stream=self._capture_stream):
                    st = self._phase_a(static_in, None, hwc)
                    pinned.copy_(st["det_counts"], non_blocking=True)
                entry = (graph, static_in, st, pinned, self.flops_last - flops0, key)
                self.flops_last = flops0
            self._graphs[key] = entry                       # (re-)inserted last = most recently used
            graph, static_in, st, pinned, flops, _ = entry
            if fused is not None:
                fused.run(self, static_in[1])           # horizontal pass + (vertical pass, normalise, pad, layout) into the graph's input
            elif frames_direct:
                self.preprocess_frames(frames, hwc, static_in[1])
            elif frames is not None:
                torch.stack(frames, out=static_in)      # one gather kernel: the frames land in the graph's input directly
            else:
                static_in.copy_(images_u8, non_blocking=True)
            graph.replay()
            self.flops_last += flops
            st = dict(st)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        st["counts_pinned"], st["counts_event"] = pinned, ev
        return st

    def _phase_b(self, st, orig_hw):
        """DensePose branch (sized by the detection counts R - the one host read-back of the path) + postprocess."""
        n, h, w = st["n"], st["h"], st["w"]
        det_boxes, det_scores, det_counts = st["det_boxes"], st["det_scores"], st["det_counts"]
        if self.keep_intermediates:
            self.inter["detections"] = (det_boxes, det_scores.clone(), det_counts)    # network-input coordinates, before detector_postprocess
        flops0 = self.flops_last
        slots = self._dp_slots(n, det_boxes.shape[1])
        coarse, fine, u, v = self.densepose_branch(st["feats"], det_boxes, det_counts, st.get("dec"), slots=slots)
        flops_dp = self.flops_last - flops0
        # the returned `scores` are slices of this tensor: with graph replay st[...] lives in the graph's memory pool and is
        # overwritten by the next replay, so the results get their own copy (n x D floats). Enqueued BEHIND the DensePose branch: a
        # device-to-device copy is a blit launch with ~15 us of queue latency, which the branch's first kernel used to wait behind
        det_scores = det_scores.clone()
        # detector_postprocess (postprocessing.py:43-54); image_size there is [W_pad, H_pad] (Q1) minus the padding
        D = det_boxes.shape[1]
        # per-image scale factors and output sizes: a video / benchmark stream repeats the same geometry batch after batch, so the
        # two small device tensors are made once per geometry instead of one pageable upload + two slicing kernels per step
        mkey = (tuple(orig_hw), h, w, D)
        cached = self._meta_cache.get(mkey)
        if cached is None:
            meta = np.zeros((n, 4), dtype=np.float32)
            for i, (H0, W0) in enumerate(orig_hw):
                meta[i] = (np.float32(W0) / np.float32(w), np.float32(H0) / np.float32(h), H0, W0)
            meta_d = torch.from_numpy(meta).to(self.device)
            if len(self._meta_cache) >= 64:
                self._meta_cache.clear()
            cached = self._meta_cache[mkey] = (meta_d[:, :2].contiguous(), meta_d[:, 2:].contiguous())
        scale_d, hw_d = cached
        fin_boxes = self._empty((n, D, 4), torch.float32)
        keep = self._empty((n, D), torch.int32)
        p = L.PostprocessParams()
        p.boxes, p.counts, p.n_img, p.max_dets = det_boxes.data_ptr(), det_counts.data_ptr(), n, D
        p.scale_xy, p.out_hw, p.out_boxes, p.keep = scale_d.data_ptr(), hw_d.data_ptr(), fin_boxes.data_ptr(), keep.data_ptr()
        L.check(self.lib.dp_postprocess_boxes(C.byref(p), self._stream()), "dp_postprocess_boxes")
        # Everything of the step is enqueued; only now does the host need R - to cut the result views (postprocessing.py:52-61
        # returns [R, ...] tensors). The device does not wait for this read-back any more.
        st["counts_event"].synchronize()
        counts_host = st["counts_pinned"].numpy().astype(np.int64)
        offs = np.zeros((n,), dtype=np.int64)
        offs[1:] = np.cumsum(counts_host)[:-1]
        R = int(counts_host.sum())
        self._dp_slots(n, D, seen=R)
        if R > slots:
            # more boxes than the high-water mark allowed for (first step of a busier scene): once more with room for all of them
            coarse, fine, u, v = self.densepose_branch(st["feats"], det_boxes, det_counts, st.get("dec"), slots=self._dp_slots(n, D))
        self.flops_last = flops0 + (flops_dp * R) // max(slots, 1)     # the launches cover `slots` slots, R of them do work
        if self.keep_intermediates:
            for k in ("dp_pooled", "dp_head_out"):
                a = self.inter[k]
                self.inter[k] = Act(a.t[:R], R, a.H, a.W, a.C)
        results = []
        classes = torch.zeros((n, D), dtype=torch.int64, device=self.device)   # single class: person (fast_rcnn.py:128); the caller owns it
        for i in range(n):
            r = int(counts_host[i])
            o = int(offs[i])
            results.append({
                "image_size": torch.tensor([orig_hw[i][0], orig_hw[i][1]], dtype=torch.int64),
                "pred_boxes": fin_boxes[i, :r],
                "scores": det_scores[i, :r],
                "pred_classes": classes[i, :r],
                "pred_densepose_coarse_segm": coarse[o:o + r],
                "pred_densepose_fine_segm": fine[o:o + r],
                "pred_densepose_u": u[o:o + r],
                "pred_densepose_v": v[o:o + r],
            })
        return results, (keep, counts_host)

    @torch.no_grad()
    def forward_batch(self, images_u8, orig_hw, given_boxes=None, num_streams=1, slot=0, hwc=False):
        """images_u8: uint8 [n,3,h,w] on the device (already resized, defaults.py:89) - or, with hwc, [n,h,w,3] / a list of n
        [h,w,3] device frames that already have the test size (read as handed over). orig_hw: list of (H, W).
        Returns a list of n dicts with the reference's 8 keys (postprocessing.py:52-61).

        num_streams > 1 splits the batch into that many sub-batches, each running the whole path on its own HIP stream:
        frames are independent, so the sub-batches' kernels fill each other's partial waves / tails and the
        detection-count read-back of one overlaps the other's kernels. Results are identical to num_streams=1."""
        from .resize import FusedResize
        n = len(images_u8) if isinstance(images_u8, (list, tuple)) else (images_u8.n if isinstance(images_u8, FusedResize) else images_u8.shape[0])
        self.flops_last = 0
        self.inter = {}
        g = max(1, min(int(num_streams), n)) if given_boxes is None else 1
        if g == 1:
            st = self._phase_a_run(images_u8, ("lane", slot), given_boxes, hwc)   # one HIP graph instance per pipeline lane
            results, keep = self._phase_b(st, orig_hw)
            self._pending_keep = [(keep, 0)]
            return results
        if isinstance(images_u8, (list, tuple)):
            images_u8 = torch.stack(images_u8)
        if not hasattr(self, "_streams") or len(self._streams) < g:
            self._streams = [self.new_stream() for _ in range(g)]
        main = torch.cuda.current_stream(self.device)
        bounds = [(i * n) // g for i in range(g + 1)]
        states = []
        for k in range(g):
            s = self._streams[k]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                states.append(self._phase_a_run(images_u8[bounds[k]:bounds[k + 1]], k + 1, None, hwc))
        results, self._pending_keep = [], []
        for k in range(g):
            with torch.cuda.stream(self._streams[k]):
                res, keep = self._phase_b(states[k], orig_hw[bounds[k]:bounds[k + 1]])
            for r in res:
                for t in r.values():
                    if t.is_cuda:
                        t.record_stream(main)
            self._pending_keep.append((keep, bounds[k]))
            results.extend(res)
        for k in range(g):
            main.wait_stream(self._streams[k])
        return results

    def apply_keep_filter(self, results):
        """Drops detections whose rescaled box has negative extent (cannot happen for finite decoded boxes; kept for
        bug-compatibility with postprocessing.py:51). Costs one extra sync, so callers may skip it."""
        for (keep, counts_host), first in self._pending_keep:
            kh = keep.cpu().numpy()
            for i in range(len(counts_host)):
                res = results[first + i]
                r = int(counts_host[i])
                k = kh[i, :r].astype(bool)
                if not k.all():
                    idx = torch.from_numpy(np.nonzero(k)[0]).to(self.device)
                    for key in list(res):
                        if key != "image_size":
                            res[key] = res[key][idx]
        return results
