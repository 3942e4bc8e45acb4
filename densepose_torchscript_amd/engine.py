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
from .pack import PackedModel, round_up
from .weights import decoder_layout, resnet_blocks

FPN_STRIDES = (4, 8, 16, 32, 64)
# torchvision 0.16.2 batched_nms switches from the coordinate-offset trick to the per-class loop above this many box
# ELEMENTS: 4000 where the reference runs on the CPU (what the goldens were recorded with), 20000 in its CUDA mode
# (run.py:22-29). At 800x1333 the RPN feeds 4 x 4819 = 19276 elements: per-level loop on the CPU, trick on CUDA; the two
# differ only where an IoU sits within rounding of the threshold. Engine.nms_reference picks the one to reproduce.
NMS_TRICK_MAX_NUMEL = {"cpu": 4000, "cuda": 20000}
MAX_GRAPHS = 8   # captured HIP graphs kept per engine (each pins the activations of its shape): least recently used is dropped;
                 # raised to the number of live (stream slot, pipeline lane) pairs when that is larger (Engine._graph_cap)

_engine_device = None   # the one device this process drives (one process per GPU: DESIGN.md §5)


def _ptr(t):
    return t.data_ptr() if t is not None else None


class Act:
    """NHWC activation living in a torch allocation."""
    __slots__ = ("t", "N", "H", "W", "C")

    def __init__(self, t, N, H, W, C):
        self.t, self.N, self.H, self.W, self.C = t, N, H, W, C


class Engine:
    def __init__(self, cfg, state, dtype="bf16", device="cuda:0"):
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
        # Measured (bench.py, 2 runs each, same box): none 889 / 892 img/s, FPN 896 / 897, RPN 915 / 914, FPN + RPN 915 / 903;
        # the decoder's heads fork from a stream that is itself a fork, which hipGraph capture does not survive (segfault in
        # capture_end on ROCm 7.2) - so only the RPN levels are forked by default (DP_FORK overrides, for experiments)
        import os as _os
        self.fork_levels = int(_os.environ.get("DP_FORK", "2"))
        self.frames_direct = _os.environ.get("DP_FRAMES_DIRECT", "1") != "0"     # A/B knob: 0 = stack the frames of a batch first (round 3)
        self._forked = {}
        self.fuse_stem_pool = True    # stem conv + ReLU + max-pool in one launch (dp_stem_pool_nhwc)
        self.fuse_rpn_head = True     # RPN 3x3 conv + 1x1 heads in one launch where the 256-cout ring kernel runs the level
        self.fuse_bottleneck = True   # res2 blocks: conv2 -> conv3 -> next conv1 in one launch (bottleneck_tail)
        self.overlap_decoder = True   # decoder on a side stream beside the RPN / box branch (see _phase_a)
        self.fuse_shortcut = _os.environ.get("DP_FUSE_SHORTCUT", "1") != "0"   # block-0 projection shortcut as K planes of conv3 (16-bit modes)
        self.fuse_sc_tail = _os.environ.get("DP_FUSE_SC_TAIL", "1") != "0"     # ... of res2.0 too (stride 1: inside the fused bottleneck tail)
        self.fuse_pair = _os.environ.get("DP_FUSE_PAIR", "1") != "0"   # A/B knob: 0 = conv3 and the next block's conv1 of res3's plain blocks as two launches
        self.group_deconv = _os.environ.get("DP_GROUP_DECONV", "1") != "0"   # A/B knob: 0 = the predictor's four sub-pixel convolutions as four launches
        self.split_k_on = _os.environ.get("DP_SPLIT_K", "1") != "0"   # A/B knob: layers with PackedConv.split_k run unsplit
        self.decoder_fold = True      # 16-bit modes: the decoder's level sum in the conv epilogues (post_res) instead of a merge pass
        self._shared_chip = 0         # dp_conv_params.shared_chip of the launches being issued: 1 beside other large launches, 2 beside the top-k / NMS chain
        self.decoder_after_rpn_heads = _os.environ.get("DP_DEC_LATE", "1") != "0"   # where the decoder's side stream forks (see _phase_a)
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

    def _empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.tdt, device=self.device)

    def conv(self, layer, x, relu=False, residual=None, rshift=0, out_f32=False, out=None, out_c_stride=None, out_c_off=0,
             out_geom=None, out_hw=None, head=None, post=None, post_mode=0, n_dev=None, in2=None, groups=None, ring_order=False):
        """x: Act. Returns Act. out_geom: (osN, osH, osW, base_elems) override for the sub-pixel deconv; out_hw: (Ho, Wo)
        override (the paired-pixel stem, whose input is narrower than its output is wide). head: (weight [16, Cout], bias [16],
        macs per pixel) of a fused 1x1 head on this layer's ReLU output - the call then returns the HEAD's fp32 output
        [N, Ho, Wo, 16] and the hidden tensor is never written (caller checks head_fusable first). post / post_mode: an Act added
        AFTER the activation (dp_conv_params.post_res: 1 = same geometry, 2 = half-size map through a bilinear x2; caller checks
        post_fusable first). n_dev: int32 device tensor [1] = how many of the x.N images hold data (dp_conv_params.n_dev).
        in2: second source Act of a pack.dual_source_pointwise layer (dp_conv_params.in2), read at stride layer.stride2.
        groups: [(layer_g, base element of its output inside `out`)] - 2 .. 4 layers of `layer`'s geometry in ONE launch
        (dp_conv_params.n_groups; out_geom gives the shared strides, its base is ignored; caller checks groups_fusable first)."""
        if (in2 is not None or post is not None) and head is None and out_geom is None and out_c_stride is None and not out_f32:
            # The kernels behind in2 / post address their tensors with 32-bit byte offsets: a batch whose largest tensor exceeds
            # 2 GiB (64 frames of 800x1344 at the res3 / p2 levels) goes image chunk by image chunk. Per-pixel arithmetic does not
            # depend on the chunking (nor on the batch size: the same kernels run either way).
            es_ = x.t.element_size()
            per_img = max(x.H * x.W * x.C, x.H * x.W * layer.cout, in2.H * in2.W * in2.C if in2 is not None else 0) * es_
            per = max(1, ((1 << 31) - 1) // per_img)
            if per < x.N:
                sl = lambda a, n0, n: Act(a.t[n0:n0 + n], n, a.H, a.W, a.C)   # noqa: E731
                if out is None:
                    out = self._empty((x.N, x.H, x.W, layer.cout))
                for n0 in range(0, x.N, per):
                    n = min(per, x.N - n0)
                    self.conv(layer, sl(x, n0, n), relu=relu, residual=None if residual is None else sl(residual, n0, n), rshift=rshift,
                              out=out[n0:n0 + n], post=None if post is None else sl(post, n0, n), post_mode=post_mode,
                              in2=None if in2 is None else sl(in2, n0, n))
                return Act(out, x.N, x.H, x.W, layer.cout)
        p = L.ConvParams()
        N, H, W = x.N, x.H, x.W
        if in2 is not None:
            assert x.C == layer.cin1 and in2.C == layer.cin2 and in2.N == N, (layer.name, x.C, in2.C)
            assert in2.H >= (H - 1) * layer.stride2 + 1 and in2.W >= (W - 1) * layer.stride2 + 1
            p.in2, p.H2, p.W2, p.Cin2, p.stride2 = in2.t.data_ptr(), in2.H, in2.W, in2.C, layer.stride2
        else:
            assert x.C == layer.cin, (layer.name, x.C, layer.cin)
        s = layer.stride
        if s == 1:
            Ho, Wo = H, W
        else:
            # every strided conv of this model has pad = (k-1)/2 -> Ho = floor((H - 1) / s) + 1
            Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        if out_hw is not None:
            Ho, Wo = out_hw
        cs = out_c_stride or layer.cout
        odt = torch.float32 if out_f32 else self.tdt
        head_out = None
        if head is not None:
            head_out = self._empty((N, Ho, Wo, 16), torch.float32)
            p.head_w, p.head_b, p.head_out = head[0].data_ptr(), head[1].data_ptr(), head_out.data_ptr()
        elif out is None:
            out = self._empty((N, Ho, Wo, cs), odt)
        p.in_, p.weight, p.ktab, p.bias = x.t.data_ptr(), layer.weight.data_ptr(), layer.ktab.data_ptr(), layer.bias.data_ptr()
        p.residual = residual.t.data_ptr() if residual is not None else None
        es_out = out.element_size() if out is not None else x.t.element_size()
        if head is not None:
            p.out = None
            p.osN, p.osH, p.osW = Ho * Wo * cs, Wo * cs, cs
        elif out_geom is None:
            p.out = out.data_ptr() + out_c_off * es_out
            p.osN, p.osH, p.osW = Ho * Wo * cs, Wo * cs, cs
        else:
            osN, osH, osW, base = out_geom
            p.out = out.data_ptr() + base * es_out
            p.osN, p.osH, p.osW = osN, osH, osW
        p.N, p.H, p.W, p.Cin = N, H, W, x.C
        p.Ho, p.Wo, p.Cout = Ho, Wo, layer.cout
        p.Cout_w, p.Kpad = layer.cout_w, layer.kpad
        p.stride = s
        p.stride_w = layer.stride_w
        p.ntaps = layer.ntaps
        if residual is not None:
            p.rsN, p.rsH, p.rsW = residual.H * residual.W * residual.C, residual.W * residual.C, residual.C
            assert residual.C == layer.cout
        p.rshift = rshift
        p.relu = 1 if relu else 0
        p.dtype = self.dt
        p.out_f32 = 1 if out_f32 else 0
        p.hi_off, p.wi_off = layer.hi_off, layer.wi_off
        p.shared_chip = int(self._shared_chip)
        p.ring_order = 1 if ring_order else 0      # dp_conv_params.ring_order: the bits of the LDS-ring family whatever the batch
        if n_dev is not None:
            p.n_dev = n_dev.data_ptr()
        if post is not None:
            assert post.C == layer.cout and post.N == N and (post.H, post.W) == ((Ho, Wo) if post_mode == 1 else (Ho // 2, Wo // 2))
            p.post_res, p.post_mode = post.t.data_ptr(), post_mode
        if groups is not None:
            assert out_geom is not None and 2 <= len(groups) <= 4 and residual is None and head is None and in2 is None and post is None
            p.n_groups = len(groups)
            for g, (lg, base) in enumerate(groups):
                assert (lg.cout, lg.cout_w, lg.kpad, lg.ntaps, lg.stride, lg.hi_off, lg.wi_off) == (
                    layer.cout, layer.cout_w, layer.kpad, layer.ntaps, layer.stride, layer.hi_off, layer.wi_off), lg.name
                p.weight_g[g], p.ktab_g[g], p.out_g[g] = lg.weight.data_ptr(), lg.ktab.data_ptr(), out.data_ptr() + base * es_out
        split_ws = None
        if (getattr(layer, "split_k", 0) > 1 and self.split_k_on and residual is None and head is None and in2 is None and post is None
                and n_dev is None and not out_f32 and out_geom is None and out_c_stride is None and N * Ho * Wo > 0):
            # long-K layers (fc1, res5's 3x3): K in layer.split_k segments of fp32 partial sums + one reduction pass - the count is
            # the layer's, whatever the batch (dp_conv_params.split_k)
            split_ws = self._empty((layer.split_k, N * Ho * Wo, layer.cout), torch.float32)
            p.split_k, p.split_ws = layer.split_k, split_ws.data_ptr()
        flops = 2 * (layer.macs_per_pixel + (head[2] if head is not None else 0)) * N * Ho * Wo * (len(groups) if groups is not None else 1)
        if self.prof is not None and N * Ho * Wo > 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
            L.check(self.lib.dp_conv2d_nhwc(C.byref(p), self._stream()), "dp_conv2d_nhwc[%s]" % layer.name)
            e1.record(torch.cuda.current_stream(self.device))
            cls = ("conv_igemm_kernel<64>", "conv_igemm_kernel<128>", "conv_ring_kernel<256x256>", "conv_ring_kernel<128x128>", "conv_ring2_kernel<256x128>", "conv1x1_stream_kernel", "conv3x3_wsr_kernel", "conv3x3_rows_kernel", "conv3x3_rows2_kernel", "conv1x1_pws_kernel", "conv3x3_wsq_kernel")[self.lib.dp_conv2d_kernel_class(C.byref(p))]
            if cls.startswith("conv_ring_kernel<"):   # one template instance (= one rocprofv3 kernel name) per tile height
                cls = "conv_ring_kernel<%dx%s" % (self.lib.dp_conv2d_tile_rows(C.byref(p)), cls.split("x")[1])
                if in2 is not None:                   # ... and per source count (the two-source form is its own instance)
                    cls = cls[:-1] + ",2src>"
                if split_ws is not None:              # ... and the split-K form (+ its reduction pass)
                    cls = cls[:-1] + ",splitk%d>" % layer.split_k
            if cls in ("conv3x3_rows_kernel", "conv3x3_rows2_kernel"):          # ... per input channel count for the row-streaming kernels
                cls = "%s<%d>" % (cls, x.C)
            if cls == "conv1x1_pws_kernel":           # ... per K length for the weight-stationary pointwise kernel
                cls = ("conv1x1_pwq_kernel<%d>" if x.C == 256 else "conv1x1_pws_kernel<%d>") % x.C
            if cls == "conv3x3_wsq_kernel":
                cls = "conv3x3_wsq_kernel<%s%s>" % ("relu" if relu else "linear", ",post%d" % post_mode if post is not None else "")
            if cls == "conv3x3_wsr_kernel":           # ... and per (channels, ReLU) for the weight-stationary kernel
                cls = "conv3x3_wsr_kernel<%d,%s%s>" % (x.C, "relu" if relu else "linear", ",post%d" % post_mode if post is not None else "")
            es = x.t.element_size()
            if in2 is not None:
                nbytes_in2 = N * Ho * Wo * in2.C * es
            nbytes = ((nbytes_in2 if in2 is not None else 0) + N * (H * W if s == 1 else Ho * Wo * min(layer.ntaps, s * s)) * x.C * es + layer.weight.numel() * es
                      + N * Ho * Wo * layer.cout * (es_out + (es if residual is not None else 0) // (4 if rshift else 1)))
            if groups is not None:   # every group has its own weights and output; the input is read once
                nbytes += (len(groups) - 1) * (layer.weight.numel() * es + N * Ho * Wo * layer.cout * es_out)
                cls = cls[:-1] + ",x%d>" % len(groups) if cls.endswith(">") else cls
            if head is not None:   # the hidden tensor is never written; the head's 16 fp32 channels are
                nbytes += N * Ho * Wo * (16 * 4 - layer.cout * es_out)
            if post is not None:
                nbytes += post.t.numel() * es
            self.prof.append((cls, flops, e0, e1, "%s%s %dx%dx%d->%d t%d" % (layer.name, "+head" if head is not None else "", Ho, Wo, x.C,
                                                                             layer.cout, layer.ntaps), nbytes))
        else:
            L.check(self.lib.dp_conv2d_nhwc(C.byref(p), self._stream()), "dp_conv2d_nhwc[%s]" % layer.name)
        self.flops_last += flops
        if head is not None:
            return Act(head_out, N, Ho, Wo, 16)
        return Act(out, N, Ho, Wo, cs)

    def stem_pool(self, layer, x):
        """resnet.py:350-354 in one launch (dp_stem_pool_nhwc). x: the paired-pixel image [N, Hp, Wp/2 + 3, 8]. Returns the pooled
        Act, or None when the library has no fused kernel for the shape (fp32 parity mode, tiny widths)."""
        p = L.StemPoolParams()
        Hp, Wp = x.H, 2 * (x.W - 3)
        p.N, p.Hp, p.Wp, p.Cout, p.Kpad, p.dtype = x.N, Hp, Wp, layer.cout, layer.kpad, self.dt
        if layer.stride != 2 or layer.stride_w != 1 or layer.ntaps != 28 or not self.lib.dp_stem_pool_supported(C.byref(p)):
            return None
        Ho, Wo = (Hp // 2 - 1) // 2 + 1, (Wp // 2 - 1) // 2 + 1
        out = self._empty((x.N, Ho, Wo, layer.cout))
        p.in_, p.weight, p.bias, p.out = x.t.data_ptr(), layer.weight.data_ptr(), layer.bias.data_ptr(), out.data_ptr()
        flops = 2 * layer.macs_per_pixel * x.N * (Hp // 2) * (Wp // 2)
        prof = self.prof is not None
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        L.check(self.lib.dp_stem_pool_nhwc(C.byref(p), self._stream()), "dp_stem_pool_nhwc")
        if prof:
            e1.record(torch.cuda.current_stream(self.device))
            es = out.element_size()
            self.prof.append(("stem_pool_kernel", flops, e0, e1, "%s+maxpool %dx%d->%dx%dx%d" % (layer.name, Hp, Wp, Ho, Wo, layer.cout),
                              x.t.numel() * es + out.numel() * es))
        self.flops_last += flops
        return Act(out, x.N, Ho, Wo, layer.cout)

    def head_fusable(self, layer, x):
        """True when dp_conv2d_nhwc can apply a fused 1x1 head in this layer's epilogue for input x: the launch lands on the
        256-cout LDS-ring kernel (all 256 channels of a pixel in one workgroup), 16-bit storage."""
        if self.dt == L.DP_F32 or layer.cout != 256 or layer.stride != 1 or not self.fuse_rpn_head:
            return False
        p = L.ConvParams()
        p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout = x.N, x.H, x.W, x.C, x.H, x.W, layer.cout
        p.Cout_w, p.Kpad, p.stride, p.ntaps, p.dtype = layer.cout_w, layer.kpad, 1, layer.ntaps, self.dt
        p.hi_off, p.wi_off = layer.hi_off, layer.wi_off
        p.osN, p.osH, p.osW = x.H * x.W * layer.cout, x.W * layer.cout, layer.cout
        return self.lib.dp_conv2d_kernel_class(C.byref(p)) == 2

    def groups_fusable(self, layer, x, n_dev=None):
        """True when dp_conv2d_nhwc takes a grouped launch (dp_conv_params.n_groups) of this layer's shape on input x: the launch lands on
        one of the 128-cout LDS-ring kernels (classes 3 / 4)."""
        if not self.group_deconv:
            return False
        p = L.ConvParams()
        p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout = x.N, x.H, x.W, x.C, x.H, x.W, layer.cout
        p.Cout_w, p.Kpad, p.stride, p.ntaps, p.dtype = layer.cout_w, layer.kpad, layer.stride, layer.ntaps, self.dt
        p.hi_off, p.wi_off, p.out_f32 = layer.hi_off, layer.wi_off, 1
        p.osN, p.osH, p.osW = 1, 1, layer.cout          # (a pixel-shuffle output: not a plain NHWC tensor)
        p.out = 4096
        if n_dev is not None:
            p.n_dev = n_dev.data_ptr()
        return layer.stride == 1 and self.lib.dp_conv2d_kernel_class(C.byref(p)) in (3, 4)

    def post_fusable(self, layer, x, post_mode):
        """True when dp_conv2d_nhwc can add a tensor after this layer's ReLU for input x (dp_conv_params.post_res: the
        weight-stationary 3x3 kernel, 256 channels, 16-bit storage; mode 2 needs even H and W)."""
        if self.dt == L.DP_F32 or layer.stride != 1 or (post_mode == 2 and (x.H % 2 or x.W % 2)):
            return False
        p = L.ConvParams()
        p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout = x.N, x.H, x.W, x.C, x.H, x.W, layer.cout
        p.Cout_w, p.Kpad, p.stride, p.ntaps, p.dtype, p.relu = layer.cout_w, layer.kpad, 1, layer.ntaps, self.dt, 1
        p.hi_off, p.wi_off = layer.hi_off, layer.wi_off
        p.osN, p.osH, p.osW = x.H * x.W * layer.cout, x.W * layer.cout, layer.cout
        p.out = 4096                      # placeholders: only NULL / non-NULL matters to the class query
        p.post_res, p.post_mode = 4096, post_mode
        p.shared_chip = int(self._shared_chip)
        return self.lib.dp_conv2d_kernel_class(C.byref(p)) in (6, 10)

    def bottleneck_tail(self, l2, l3, l1n, t1, residual, sc_in=None):
        """conv2 -> conv3 (+ residual, ReLU) -> conv1 of the next block in one launch (dp_bottleneck_tail_nhwc).
        Returns (block output, next block's conv1 output or None), or None when the library has no fused kernel for the
        shape (fp32 parity mode, every stage but res2, tiny widths): the caller then runs the layers one by one.
        sc_in: the block's input when l3 is the block's conv3 + projection shortcut as one dual-source layer (pack.dual_source_pointwise,
        stride 1): the shortcut rides in conv3's K axis, there is no residual tensor and no next-conv1 stage."""
        p = L.BottleneckParams()
        N, H, W = t1.N, t1.H, t1.W
        p.N, p.H, p.W = N, H, W
        p.Cmid, p.Cout, p.Cmid_next = l2.cout, l3.cout, (l1n.cout if l1n is not None else 0)
        p.Kpad2, p.Kpad3, p.Kpad1n = l2.kpad, l3.kpad, (l1n.kpad if l1n is not None else 0)
        p.ntaps2, p.hi_off2, p.wi_off2, p.dtype = l2.ntaps, l2.hi_off, l2.wi_off, self.dt
        p.k_order2 = 0 if l2.plane_major else 1
        if sc_in is not None:
            if (l1n is not None or residual is not None or getattr(l3, "stride2", 0) != 1 or l3.cin1 != l2.cout or sc_in.C != l3.cin2
                    or (sc_in.N, sc_in.H, sc_in.W) != (N, H, W) or l2.stride != 1 or l3.ntaps != 1 or t1.C != l2.cin):
                return None
            p.Csc, p.sc_in = sc_in.C, 4096             # placeholder: only NULL / non-NULL matters to the support query
        elif (l2.stride != 1 or l3.stride != 1 or l3.ntaps != 1 or t1.C != l2.cin or l2.cout != l3.cin or residual.C != l3.cout
                or (l1n is not None and (l1n.stride != 1 or l1n.ntaps != 1 or l1n.cin != l3.cout))):
            return None
        # 32-bit buffer offsets inside the kernel: large batches go image chunk by image chunk
        per = max(1, ((1 << 31) // (l3.cout * 2) - (1 << 17)) // (H * W))
        p.N = min(N, per)
        p.next_t1 = 1 if l1n is not None else None   # placeholder: only NULL / non-NULL matters to the support query
        if not self.lib.dp_bottleneck_tail_supported(C.byref(p)):
            return None
        out = self._empty((N, H, W, l3.cout))
        t1n = self._empty((N, H, W, l1n.cout)) if l1n is not None else None
        p.w2, p.w3, p.ktab2, p.b2, p.b3 = l2.weight.data_ptr(), l3.weight.data_ptr(), l2.ktab.data_ptr(), l2.bias.data_ptr(), l3.bias.data_ptr()
        if l1n is not None:
            p.w1n, p.b1n = l1n.weight.data_ptr(), l1n.bias.data_ptr()
        es = out.element_size()
        macs = l2.macs_per_pixel + l3.macs_per_pixel + (l1n.macs_per_pixel if l1n is not None else 0)
        flops = 2 * macs * N * H * W
        prof = self.prof is not None and N * H * W > 0
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        for n0 in range(0, N, per):
            n = min(per, N - n0)
            px = n0 * H * W
            p.N = n
            p.t1, p.out = t1.t.data_ptr() + px * t1.C * es, out.data_ptr() + px * l3.cout * es
            if sc_in is not None:
                p.sc_in = sc_in.t.data_ptr() + px * sc_in.C * es
            else:
                p.residual = residual.t.data_ptr() + px * residual.C * es
            p.next_t1 = (t1n.data_ptr() + px * l1n.cout * es) if l1n is not None else None
            L.check(self.lib.dp_bottleneck_tail_nhwc(C.byref(p), self._stream()), "dp_bottleneck_tail_nhwc[%s]" % l2.name)
        if prof:
            e1.record(torch.cuda.current_stream(self.device))
            nbytes = N * H * W * es * (t1.C + l3.cout + (sc_in.C if sc_in is not None else l3.cout) + (l1n.cout if l1n is not None else 0)) + (
                l2.weight.numel() + l3.weight.numel()) * es
            self.prof.append(("bottleneck_tail64_kernel", flops, e0, e1, "%s+conv3%s %dx%dx%d->%d" % (
                l2.name, "+shortcut" if sc_in is not None else "+next conv1" if l1n is not None else "", H, W, t1.C, l3.cout), nbytes))
        self.flops_last += flops
        return Act(out, N, H, W, l3.cout), (Act(t1n, N, H, W, l1n.cout) if l1n is not None else None)

    def fused_shortcut_blocks(self):
        """Prefixes of the bottleneck blocks whose projection shortcut rides in conv3's K axis (no rounded shortcut tensor): what the
        storage-emulating oracle of the tests has to mirror."""
        bu = "backbone.bottom_up."
        return [p for p, stride, sc in (("%s%s.%d." % (bu, st, b), stride, sc) for st, b, _, _, _, stride, sc in resnet_blocks(self.cfg))
                if sc and self.fuse_shortcut and (stride != 1 or self.fuse_sc_tail) and (p + "conv3+shortcut") in self.model.layers]

    def bottleneck_pair(self, l3, l1n, t2, residual):
        """conv3 (+ residual, ReLU) -> conv1 of the next block in one launch (dp_bottleneck_pair_nhwc: the plain blocks of res3).
        Returns (block output, next block's conv1 output), or None when the library has no fused kernel for the shape (fp32 parity
        mode, other stages, tiny widths): the caller then runs the layers one by one."""
        if not self.fuse_pair or l1n is None:
            return None
        p = L.PairParams()
        N, H, W = t2.N, t2.H, t2.W
        p.Cmid, p.Cout, p.Cmid_next, p.Kpad3, p.Kpad1n, p.dtype = l3.cin, l3.cout, l1n.cout, l3.kpad, l1n.kpad, self.dt
        if (l3.stride != 1 or l3.ntaps != 1 or l1n.stride != 1 or l1n.ntaps != 1 or t2.C != l3.cin or residual.C != l3.cout or l1n.cin != l3.cout
                or (residual.H, residual.W) != (H, W)):
            return None
        # 32-bit buffer offsets inside the kernel: large batches go image chunk by image chunk (pixels are independent)
        per = max(1, ((1 << 31) // (l3.cout * 2) - 64) // (H * W))
        p.M = min(N, per) * H * W
        if not self.lib.dp_bottleneck_pair_supported(C.byref(p)):
            return None
        out = self._empty((N, H, W, l3.cout))
        t1n = self._empty((N, H, W, l1n.cout))
        p.w3, p.w1n, p.b3, p.b1n = l3.weight.data_ptr(), l1n.weight.data_ptr(), l3.bias.data_ptr(), l1n.bias.data_ptr()
        es = out.element_size()
        flops = 2 * (l3.macs_per_pixel + l1n.macs_per_pixel) * N * H * W
        prof = self.prof is not None and N * H * W > 0
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        for n0 in range(0, N, per):
            n = min(per, N - n0)
            px = n0 * H * W
            p.M = n * H * W
            p.t2, p.residual = t2.t.data_ptr() + px * t2.C * es, residual.t.data_ptr() + px * residual.C * es
            p.out, p.next_t1 = out.data_ptr() + px * l3.cout * es, t1n.data_ptr() + px * l1n.cout * es
            L.check(self.lib.dp_bottleneck_pair_nhwc(C.byref(p), self._stream()), "dp_bottleneck_pair_nhwc[%s]" % l3.name)
        if prof:
            e1.record(torch.cuda.current_stream(self.device))
            nbytes = N * H * W * es * (t2.C + 2 * l3.cout + l1n.cout) + (l3.weight.numel() + l1n.weight.numel()) * es
            self.prof.append(("bottleneck_pair%d_kernel" % l3.cin, flops, e0, e1, "%s+next conv1 %dx%dx%d->%d->%d" % (l3.name, H, W, t2.C, l3.cout, l1n.cout), nbytes))
        self.flops_last += flops
        return Act(out, N, H, W, l3.cout), Act(t1n, N, H, W, l1n.cout)

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
        p.n_img, p.n_slots, p.iou_thr, p.max_out, p.trick_max_numel = n, slots, thr, max_out, NMS_TRICK_MAX_NUMEL[self.nms_reference]
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
        import os as _os
        fold = (self.decoder_fold and _os.environ.get("DP_DECODER_FOLD", "1") != "0"
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
            # everything that changes the captured launch sequence is part of the key
            key = (shape, hwc, slot, self.overlap_decoder, self.decoder_after_rpn_heads, self.fuse_bottleneck, self.fuse_rpn_head, self.fuse_stem_pool, self.fork_levels, self.nms_reference, self.decoder_fold, self.fuse_shortcut, self.fuse_sc_tail, self.fuse_pair, self.group_deconv, self.split_k_on)
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
        # the returned `scores` are slices of this tensor: with graph replay st[...] lives in the graph's memory pool and is
        # overwritten by the next replay, so the results get their own copy (n x D floats)
        det_scores = det_scores.clone()
        if self.keep_intermediates:
            self.inter["detections"] = (det_boxes, det_scores, det_counts)    # network-input coordinates, before detector_postprocess
        flops0 = self.flops_last
        slots = self._dp_slots(n, det_boxes.shape[1])
        coarse, fine, u, v = self.densepose_branch(st["feats"], det_boxes, det_counts, st.get("dec"), slots=slots)
        flops_dp = self.flops_last - flops0
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
