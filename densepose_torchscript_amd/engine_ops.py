"""Layer launches of the engine: one method per C-ABI convolution entry point (dp_conv2d_nhwc and the fused forms), the questions the
stages ask the library before they pick a fused form (head / groups / post fusable), and the per-launch profile records. Mixed into
engine.Engine; nothing here schedules streams or graphs (engine.py) or knows the network's stage order (engine_stages.py)."""
import ctypes as C

import torch

from . import lib as L
from .weights import resnet_blocks


class Act:
    """NHWC activation living in a torch allocation."""
    __slots__ = ("t", "N", "H", "W", "C")

    def __init__(self, t, N, H, W, C):
        self.t, self.N, self.H, self.W, self.C = t, N, H, W, C


# dp_conv2d_kernel_class() -> the kernel family the launch lands on (include/densepose_hip.h), as the per-launch profile labels it
KERNEL_CLASS_NAMES = ("conv_igemm_kernel<64>", "conv_igemm_kernel<128>", "conv_ring_kernel<256x256>", "conv_ring_kernel<128x128>",
                      "conv_ring2_kernel<256x128>", "conv1x1_stream_kernel", "conv3x3_wsr_kernel", "conv3x3_rows_kernel", "conv3x3_rows2_kernel",
                      "conv1x1_pws_kernel", "conv3x3_wsq_kernel")


class LayerOps:
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
            cls = KERNEL_CLASS_NAMES[self.lib.dp_conv2d_kernel_class(C.byref(p))]
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
            if cls == "conv3x3_wsq_kernel":           # kernel class 10: conv3x3_ws1_kernel (16x16x32, default) or conv3x3_wsq_kernel (policy wsq_shape = 32)
                cls = "%s<%s%s>" % ("conv3x3_wsq_kernel" if L.get_policy("wsq_shape") == 32 else "conv3x3_ws1_kernel",
                                    "relu" if relu else "linear", ",post%d" % post_mode if post is not None else "")
            if cls == "conv3x3_wsr_kernel":           # ... and per (channels, ReLU) for the weight-stationary kernel
                cls = "conv3x3_wsr_kernel<%d,%s%s>" % (x.C, "relu" if relu else "linear", ",post%d" % post_mode if post is not None else "")
            es = x.t.element_size()
            if in2 is not None:
                nbytes_in2 = N * Ho * Wo * in2.C * es
            nbytes = ((nbytes_in2 if in2 is not None else 0) + N * (H * W if s == 1 else Ho * Wo * min(layer.ntaps, s * s)) * x.C * es
                      + layer.weight.numel() * es
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

    def kernel_class(self, layer, x, relu=True):
        """the kernel class (include/densepose_hip.h) a plain stride-1 launch of `layer` on x lands on"""
        if layer.stride != 1:
            return -1
        p = L.ConvParams()
        p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout = x.N, x.H, x.W, x.C, x.H, x.W, layer.cout
        p.Cout_w, p.Kpad, p.stride, p.ntaps, p.dtype, p.relu = layer.cout_w, layer.kpad, 1, layer.ntaps, self.dt, 1 if relu else 0
        p.hi_off, p.wi_off = layer.hi_off, layer.wi_off
        p.osN, p.osH, p.osW = x.H * x.W * layer.cout, x.W * layer.cout, layer.cout
        p.out = 4096                      # placeholder: only NULL / non-NULL matters to the class query
        p.shared_chip = int(self._shared_chip)
        return self.lib.dp_conv2d_kernel_class(C.byref(p))

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
            self.prof.append(("bottleneck_pair%d_kernel" % l3.cin, flops, e0, e1,
                              "%s+next conv1 %dx%dx%d->%d->%d" % (l3.name, H, W, t2.C, l3.cout, l1n.cout), nbytes))
        self.flops_last += flops
        return Act(out, N, H, W, l3.cout), Act(t1n, N, H, W, l1n.cout)

