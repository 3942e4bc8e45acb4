"""Kernel-level parity tests: each C-ABI entry point against a plain torch fp32 / oracle computation (GPU box only)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
    from densepose_torchscript_amd.engine import Engine
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
    return {dt: Engine(cfg, make_synthetic_state(cfg, 0), dtype=dt) for dt in ("fp32", "bf16", "fp16")}


def _nhwc(x, calloc, tdt, dev):
    n, c, h, w = x.shape
    t = torch.zeros((n, h, w, calloc), dtype=torch.float32)
    t[..., :c] = x.permute(0, 2, 3, 1)
    return t.to(tdt).to(dev)


def _round(x, dt):
    """round to the storage type of mode `dt` (exactly representable operands -> only accumulation order differs)"""
    return x.to({"bf16": torch.bfloat16, "fp16": torch.float16}[dt]).to(torch.float32)


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil, relu, residual
    (1, 64, 20, 24, 64, 1, 1, 0, 1, True, False),
    (2, 64, 17, 19, 256, 1, 1, 0, 1, True, True),
    (1, 256, 20, 26, 128, 1, 2, 0, 1, False, False),
    (1, 64, 23, 31, 64, 3, 1, 1, 1, True, False),
    (2, 256, 13, 21, 256, 3, 1, 1, 1, False, False),
    (1, 3, 64, 96, 64, 7, 2, 3, 1, True, False),
    (1, 32, 28, 28, 32, 3, 1, 6, 6, False, False),
    (1, 256, 14, 14, 15, 1, 1, 0, 1, False, False),
    (3, 8, 9, 11, 16, 3, 1, 1, 1, True, False),
    (1, 512, 28, 28, 512, 3, 1, 1, 1, True, False),
]


BIG_CASES = [
    # shapes the large-tile (256x256, LDS ring) kernel is legal for; DP_CONV_BIG=1 forces it even for few tiles
    (2, 256, 13, 21, 256, 3, 1, 1, 1, False, False),
    (1, 512, 28, 28, 512, 3, 1, 1, 1, True, False),
    (3, 64, 20, 37, 256, 1, 1, 0, 1, True, True),
    (1, 256, 40, 56, 256, 3, 1, 1, 1, True, False),
    (2, 1024, 9, 11, 512, 1, 2, 0, 1, False, False),
    (1, 32, 30, 30, 256, 3, 1, 2, 2, False, False),
    (1, 64, 17, 23, 40, 3, 1, 1, 1, True, False),     # Cout not a multiple of 128 (128x128 ring only; 256 falls back)
    (2, 32, 5, 7, 256, 1, 1, 0, 1, False, False),     # a single K plane
    (1, 96, 12, 12, 128, 1, 1, 0, 1, True, True),     # three K planes
]


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("tp", ["4", "5", "6", "7"])   # tile heights 128 .. 224 of the 256-cout ring kernel (8 = default, above)
@pytest.mark.parametrize("case", BIG_CASES[:6])
def test_conv2d_ring_tile_heights(eng, dt, case, tp, policy):
    policy.set("conv_big", "1")
    policy.set("conv_tp", tp)
    test_conv2d_matches_torch(eng, dt, case)


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("force", ["1", "2", "3"])   # 1: 256x256 ring kernel, 2: 128x128 ring kernel, 3: 256x128 two-WG ring kernel
@pytest.mark.parametrize("case", BIG_CASES)
def test_conv2d_ring_kernels(eng, dt, case, force, policy):
    policy.set("conv_big", force)
    test_conv2d_matches_torch(eng, dt, case)


def _random_conv_cases(n, seed):
    """Seeded random layer shapes over everything the engine can emit: ragged M, channel counts that are not tile
    multiples, stride 2 (pad = (k-1)/2 like every strided conv of the model), dilation, residual / ReLU."""
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        k = int(rng.choice([1, 3]))
        s = int(rng.choice([1, 1, 2]))
        d = int(rng.choice([1, 1, 2])) if (k == 3 and s == 1) else 1
        cin = int(rng.choice([8, 16, 32, 64, 96, 128, 256]))
        cout = int(rng.choice([8, 16, 40, 64, 128, 256, 512]))
        n_img = int(rng.integers(1, 4))
        h, w = int(rng.integers(5, 41)), int(rng.integers(5, 41))
        if n_img * h * w * cin * k * k * cout > 3e9:
            continue
        cases.append((n_img, cin, h, w, cout, k, s, d * (k - 1) // 2, d, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return cases


@pytest.mark.parametrize("force", [None, "1", "2", "3", "5"])
@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_conv2d_random_shapes(eng, dt, force, policy):
    """Every kernel class (forced where it is legal for the shape, else the generic fallback) on 24 seeded random shapes."""
    if force is not None:
        policy.set("conv_big", force)
    for case in _random_conv_cases(24, 1234 + (0 if force is None else int(force))):
        test_conv2d_matches_torch(eng, dt, case)


STREAM_CASES = [
    # pointwise, stride 1, Cin*2 B in {128, 256, 512}, Cout % 256 == 0, M >= 4096: the streaming 1x1 kernel (storage-type output)
    (2, 64, 50, 45, 256, 1, 1, 0, 1, True, True),      # res2 conv3 shape class, M = 4500 (ragged last tile)
    (3, 64, 40, 37, 256, 1, 1, 0, 1, False, False),    # block-0 shortcut
    (1, 128, 70, 61, 512, 1, 1, 0, 1, True, True),     # res3 conv3: two cout slices
    (1, 256, 65, 64, 256, 1, 1, 0, 1, False, False),   # decoder predictor
    (1, 256, 66, 63, 1024, 1, 1, 0, 1, True, True),    # res4 conv3: four cout slices
    (2, 512, 50, 47, 128, 1, 1, 0, 1, True, False),    # res3 conv1: 128 couts, K = 16 planes (the NB = 2 instance)
    (2, 64, 50, 47, 64, 1, 1, 0, 1, True, False),      # res2.0 conv1: the 64-cout instance (Cout 64 / Cout_w 128), ragged M
]


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 256, 48, 64, 256), (3, 64, 40, 38, 256), (1, 128, 70, 62, 512)])
def test_conv2d_stream_kernel_upsampled_residual(eng, dt, shape, policy):
    """FPN lateral + top-down add (fpn.py:150-155: lateral 1x1 conv + F.interpolate(top, x2, nearest)) on the streaming 1x1
    kernel: the residual is the half-size map read at (ho >> 1, wo >> 1). Same result as the generic kernel bit for bit."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(Cin + H)
    x = _round(torch.randn((N, Cin, H, W), generator=g), dt)
    top = _round(torch.randn((N, Cout, H // 2, W // 2), generator=g), dt)
    w = _round(torch.randn((Cout, Cin, 1, 1), generator=g) * (1.0 / Cin) ** 0.5, dt)
    b = torch.randn((Cout,), generator=g)
    layer = conv_from_oihw("lat", w.numpy(), b.numpy(), Cin, 1, 0, 1, e.dt, e.device)
    xa = Act(_nhwc(x, Cin, e.tdt, e.device), N, H, W, Cin)
    ta = Act(_nhwc(top, Cout, e.tdt, e.device), N, H // 2, W // 2, Cout)
    policy.set("conv_stream", "0")
    want = e.conv(layer, xa, residual=ta, rshift=1)
    policy.default("conv_stream")
    p = L.ConvParams()
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = N, H, W, Cin, H, W, Cout, Cout, Cin
    p.stride, p.ntaps, p.dtype, p.out_f32, p.rshift = 1, 1, e.dt, 0, 1
    p.osN, p.osH, p.osW = H * W * Cout, W * Cout, Cout
    p.rsN, p.rsH, p.rsW = (H // 2) * (W // 2) * Cout, (W // 2) * Cout, Cout
    p.residual = ta.t.data_ptr()
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == 5      # this launch really is the streaming kernel
    got = e.conv(layer, xa, residual=ta, rshift=1)
    torch.cuda.synchronize()
    assert torch.equal(got.t, want.t)
    ref = F.conv2d(x.double(), w.double(), b.double()) + F.interpolate(top.double(), scale_factor=2.0, mode="nearest")
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    gd = got.t.float().cpu().permute(0, 3, 1, 2).double()
    assert bool(((gd - ref).abs() <= ulp * ref.abs() + 1e-3).all())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", STREAM_CASES)
def test_conv2d_stream_kernel(eng, dt, case, policy):
    from densepose_torchscript_amd import lib as L
    policy.set("conv_big", "5")
    orig = eng[dt].lib.dp_conv2d_kernel_class
    test_conv2d_matches_torch(eng, dt, case)
    # the storage-type launch of this shape really is the streaming kernel (class 5)
    p = L.ConvParams()
    N, Cin, H, W, Cout = case[:5]
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = N, H, W, Cin, H, W, Cout, Cout, Cin
    p.Cout_w = max(Cout, 128)
    p.stride, p.ntaps, p.dtype, p.out_f32 = 1, 1, eng[dt].dt, 0
    p.osN, p.osH, p.osW = H * W * Cout, W * Cout, Cout
    if Cout == 64:      # the 64-cout instance is the policy's own choice (the conv_big override sends Cout <= 64 to the generic kernel)
        policy.default("conv_big")
        test_conv2d_matches_torch(eng, dt, case)      # ... and computes the same layer correctly there (fp32-out launch: generic; see below)
        from densepose_torchscript_amd.engine import Act
        from densepose_torchscript_amd.pack import conv_from_oihw
        e = eng[dt]
        g = torch.Generator().manual_seed(64)
        x = _round(torch.randn((N, Cin, H, W), generator=g), dt)
        w = _round(torch.randn((Cout, Cin, 1, 1), generator=g) * (1.0 / Cin) ** 0.5, dt)
        layer = conv_from_oihw("c64", w.numpy(), np.zeros(Cout, np.float32), Cin, 1, 0, 1, e.dt, e.device)
        xa = Act(_nhwc(x, Cin, e.tdt, e.device), N, H, W, Cin)
        got = e.conv(layer, xa, relu=True)                   # storage-type output: the streaming instance
        policy.set("conv_stream", 0)
        want = e.conv(layer, xa, relu=True)                  # the generic K64 kernel: same K order, same bits
        policy.default("conv_stream")
        torch.cuda.synchronize()
        assert torch.equal(got.t, want.t)
    assert orig(C.byref(p)) == 5
    # a launch sized on the device (n_dev) never goes to the persistent kernels that ignore the count (classes 5 / 6)
    p.n_dev = 4096
    assert orig(C.byref(p)) in (0, 1, 2, 3, 4)


def test_split_k_is_a_hint_and_n_dev_launches_skip_dead_images(eng):
    """dp_conv_params.split_k on a layer the split instances do not take (a residual here) runs UNSPLIT, same bits, no error; and an n_dev
    launch of a 256 -> 256 3x3 runs on the kernel the layer always runs on (class 10 sizes its work from the live count: the choice of a
    kernel with a summation order of its own must not depend on whether a count is passed) and leaves the slots behind the live count alone."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw, set_split_k
    e = eng["bf16"]
    g = torch.Generator().manual_seed(31)
    N, Ci, H, W, Co = 4, 256, 24, 32, 256
    x = Act(torch.randn((N, H, W, Ci), generator=g).to(e.tdt).to(e.device), N, H, W, Ci)
    r = Act(torch.randn((N, H, W, Co), generator=g).to(e.tdt).to(e.device), N, H, W, Co)
    w = (torch.randn((Co, Ci, 3, 3), generator=g) * (1.0 / (9 * Ci)) ** 0.5).numpy()
    layer = conv_from_oihw("l", w, np.zeros(Co, np.float32), Ci, 1, 1, 1, e.dt, e.device)
    want = e.conv(layer, x, relu=True, residual=r)
    set_split_k(layer, 3)
    assert layer.split_k == 3
    got = e.conv(layer, x, relu=True, residual=r)          # residual: the engine does not even pass split_k; ask the library directly too
    torch.cuda.synchronize()
    assert torch.equal(got.t, want.t)
    p = L.ConvParams()
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = N, H, W, Ci, H, W, Co, layer.cout_w, layer.kpad
    p.stride, p.ntaps, p.dtype, p.hi_off, p.wi_off, p.relu = 1, 9, e.dt, -1, -1, 1
    p.osN, p.osH, p.osW = H * W * Co, W * Co, Co
    p.rsN, p.rsH, p.rsW = H * W * Co, W * Co, Co
    out2 = torch.empty_like(want.t)
    ws = torch.empty((3, N * H * W, Co), dtype=torch.float32, device=e.device)
    p.in_, p.weight, p.ktab, p.bias, p.residual, p.out = x.t.data_ptr(), layer.weight.data_ptr(), layer.ktab.data_ptr(), layer.bias.data_ptr(), r.t.data_ptr(), out2.data_ptr()
    p.split_k, p.split_ws = 3, ws.data_ptr()
    assert e.lib.dp_conv2d_nhwc(C.byref(p), e._stream()) == 0       # DP_OK: the hint is ignored where it is illegal
    torch.cuda.synchronize()
    assert torch.equal(out2, want.t)
    layer.split_k = 0
    n_dev = torch.tensor([2], dtype=torch.int32, device=e.device)
    p.residual, p.split_k, p.split_ws, p.n_dev = None, 0, None, n_dev.data_ptr()
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == 10
    out3 = torch.full((N, H, W, Co), 7.0, dtype=e.tdt, device=e.device)
    got3 = e.conv(layer, x, relu=True, out=out3, n_dev=n_dev)
    torch.cuda.synchronize()
    assert bool((got3.t[2:].float() == 7.0).all())
    plain = e.conv(layer, x, relu=True)
    assert torch.equal(got3.t[:2], plain.t[:2])
    with L.policy(conv_wsq=0):       # ... and the tiled kernels behind it (the persistent class 6 ignores the count and is never given one)
        assert e.lib.dp_conv2d_kernel_class(C.byref(p)) in (0, 1, 2, 3, 4)
        out4 = torch.full((N, H, W, Co), 7.0, dtype=e.tdt, device=e.device)
        got4 = e.conv(layer, x, relu=True, out=out4, n_dev=n_dev)
        plain4 = e.conv(layer, x, relu=True)
        torch.cuda.synchronize()
        assert bool((got4.t[3:].float() == 7.0).all())      # (the image right behind the count may share the last live tile)
        assert torch.equal(got4.t[:2], plain4.t[:2])


PWS_CASES = [
    # N, Cin, H, W, Cout, relu, residual (None / "lin" / "up"): the weight-stationary pointwise kernel (dp_conv_pw.hip, kernel class 9)
    (2, 1024, 50, 84, 256, True, None),      # res4 conv1: 2 cout slices x 128 pixel groups, 8400 pixels = 2 - 3 steps per workgroup
    (3, 1024, 13, 21, 256, True, None),      # fewer steps than pixel groups, ragged last step (819 pixels)
    (2, 1024, 26, 40, 256, False, "up"),     # fpn_lateral4: top-down map through the nearest x2 up-sampling
    (1, 1024, 1, 1000, 1024, True, None),    # fc2 (box_head.py:71-73): rows of a [1000, 1024] matrix, 8 cout slices
    (2, 512, 25, 42, 2048, True, "lin"),     # res5 conv3 + residual: no K split, 8 cout slices
    (1, 512, 37, 45, 256, False, "up"),      # fpn_lateral3 geometry class: one cout slice (odd sizes: "up" needs even ones -> below)
    (2, 2048, 25, 42, 512, True, None),      # res5 conv1: four K slices, 16-pixel steps
    (1, 2048, 26, 42, 256, False, None),     # fpn_lateral5
    (3, 2048, 7, 9, 128, False, "lin"),      # two cout slices of 64, 189 pixels
    (2, 256, 50, 84, 1024, True, "lin"),     # res4 conv3 + residual: the 64-couts-per-wave form (conv1x1_pwq_kernel), 2 cout slices of 512
    (3, 256, 13, 21, 512, False, None),      # one slice, ragged last step
]


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", PWS_CASES)
def test_conv1x1_weight_stationary_pointwise(eng, dt, case, policy):
    """conv1 / conv3 of res4 / res5 (resnet.py:189-205), the FPN laterals (fpn.py:140-157) and fc2 on the weight-stationary pointwise
    kernel: against torch in fp64 on operands rounded to the storage type, against the LDS-ring kernels (same products, another
    summation order), and every image alone against the image inside the batch, bit for bit (the K slices are added in a fixed order)."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    N, Cin, H, W, Cout, relu, rmode = case
    if rmode == "up" and (H % 2 or W % 2):
        H, W = H + H % 2, W + W % 2
    g = torch.Generator().manual_seed(Cin + Cout + N * 1000 + H * 10 + W)
    x = _round(torch.randn((N, Cin, H, W), generator=g), dt)
    w = _round(torch.randn((Cout, Cin, 1, 1), generator=g) * (1.0 / Cin) ** 0.5, dt)
    b = torch.randn((Cout,), generator=g) * 0.3
    layer = conv_from_oihw("pw", w.numpy(), b.numpy(), Cin, 1, 0, 1, e.dt, e.device)
    xa = Act(_nhwc(x, Cin, e.tdt, e.device), N, H, W, Cin)
    ra, rshift, rref = None, 0, 0.0
    if rmode == "lin":
        r = _round(torch.randn((N, Cout, H, W), generator=g), dt)
        ra, rref = Act(_nhwc(r, Cout, e.tdt, e.device), N, H, W, Cout), r.double()
    elif rmode == "up":
        r = _round(torch.randn((N, Cout, H // 2, W // 2), generator=g), dt)
        ra, rshift = Act(_nhwc(r, Cout, e.tdt, e.device), N, H // 2, W // 2, Cout), 1
        rref = F.interpolate(r.double(), scale_factor=2.0, mode="nearest")
    p = L.ConvParams()
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = N, H, W, Cin, H, W, Cout, layer.cout_w, Cin
    p.stride, p.ntaps, p.dtype, p.rshift = 1, 1, e.dt, rshift
    p.osN, p.osH, p.osW = H * W * Cout, W * Cout, Cout
    p.in_, p.weight, p.out = xa.t.data_ptr(), layer.weight.data_ptr(), xa.t.data_ptr()
    if ra is not None:
        p.residual = ra.t.data_ptr()
        p.rsN, p.rsH, p.rsW = ra.H * ra.W * Cout, ra.W * Cout, Cout
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == 9
    got = e.conv(layer, xa, relu=relu, residual=ra, rshift=rshift)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), w.double(), b.double()) + rref
    ref = F.relu(ref) if relu else ref
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    gd = got.t.float().cpu().permute(0, 3, 1, 2).double()
    assert bool(((gd - ref).abs() <= ulp * ref.abs() + 2e-3).all()), float((gd - ref).abs().max())
    # every image alone == the image inside the batch, bit for bit
    for i in sorted({0, N - 1}):
        ri = None if ra is None else Act(ra.t[i:i + 1].contiguous(), 1, ra.H, ra.W, Cout)
        one = e.conv(layer, Act(xa.t[i:i + 1].contiguous(), 1, H, W, Cin), relu=relu, residual=ri, rshift=rshift)
        assert torch.equal(one.t[0], got.t[i]), i
    # the ring kernels on the same operands: equal up to the summation order
    policy.set("conv_pws", 0)
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) != 9
    want = e.conv(layer, xa, relu=relu, residual=ra, rshift=rshift)
    torch.cuda.synchronize()
    d = (want.t.float() - got.t.float()).abs()
    assert bool((d <= 2 * ulp * want.t.float().abs() + 1e-3).all()), float(d.max())


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(eng, dt, case):
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw, round_up
    e = eng[dt]
    N, Cin, H, W, Cout, k, s, p, d, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    b = torch.randn((Cout,), generator=g)
    if dt != "fp32":
        x, w = _round(x, dt), _round(w, dt)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p, dilation=d)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        if dt != "fp32":
            res = _round(res, dt)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    cin_a = round_up(Cin, 8)
    layer = conv_from_oihw("t", w.numpy(), b.numpy(), cin_a, s, p, d, e.dt, e.device)
    xa = Act(_nhwc(x, cin_a, e.tdt, e.device), N, H, W, cin_a)
    ra = None
    if use_res:
        ra = Act(_nhwc(res, layer.cout, e.tdt, e.device), N, ref.shape[2], ref.shape[3], layer.cout)
    out = e.conv(layer, xa, relu=relu, residual=ra, out_f32=True)
    torch.cuda.synchronize()
    got = out.t.cpu()[..., :Cout].permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    # fp32 MFMA is an exact fp32 FMA chain; bf16 inputs are exact in both -> only accumulation-order error remains
    tol = 2e-5 if dt == "fp32" else 2e-5
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err <= tol * max(scale, 1.0) * (Cin * k * k) ** 0.5, (case, err)
    # padded output channels must be exact zeros (they feed the next layer's zero weights)
    if layer.cout > Cout and not use_res:
        pad = out.t.cpu()[..., Cout:]
        assert float(pad.abs().max()) == 0.0
    if dt != "fp32":
        # storage-type output (what the layers of the model actually write): one rounding to bf16 / fp16 on top
        out16 = e.conv(layer, xa, relu=relu, residual=ra)
        torch.cuda.synchronize()
        assert out16.t.dtype == e.tdt
        got16 = out16.t.float().cpu()[..., :Cout].permute(0, 3, 1, 2).double()
        ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
        bound = ulp * ref.abs() + tol * max(scale, 1.0) * (Cin * k * k) ** 0.5 + 1e-7
        assert bool(((got16 - ref).abs() <= bound).all()), (case, float((got16 - ref).abs().max()))


TAIL_CASES = [
    # N, H, W, with next conv1
    (1, 20, 24, True),
    (2, 17, 19, True),      # ragged: rows and the pixel count are no multiple of the 16 / 32-pixel tiles
    (3, 40, 37, False),
    (1, 1, 1, True),        # a single pixel: every tap but the centre is padding
    (2, 3, 70, True),
    (1, 96, 160, True),     # more wave tiles than one round of the chip takes (persistent loop, in-place prefetch)
]


def _tail_layers(e, seed):
    from densepose_torchscript_amd.pack import conv_from_oihw
    g = torch.Generator().manual_seed(seed)
    mk = lambda co, ci, k: torch.randn((co, ci, k, k), generator=g) * (1.0 / (ci * k * k)) ** 0.5  # noqa: E731
    w2, w3, w1 = mk(64, 64, 3), mk(256, 64, 1), mk(64, 256, 1)
    b2, b3, b1 = (torch.randn((c,), generator=g) * 0.5 for c in (64, 256, 64))
    l2 = conv_from_oihw("conv2", w2.numpy(), b2.numpy(), 64, 1, 1, 1, e.dt, e.device, plane_major=False)   # tap-major K, as PackedModel packs it
    l3 = conv_from_oihw("conv3", w3.numpy(), b3.numpy(), 64, 1, 0, 1, e.dt, e.device)
    l1 = conv_from_oihw("conv1n", w1.numpy(), b1.numpy(), 256, 1, 0, 1, e.dt, e.device)
    return (l2, l3, l1), (w2, b2, w3, b3, w1, b1)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", TAIL_CASES)
def test_bottleneck_tail_equals_layer_by_layer(eng, dt, case):
    """dp_bottleneck_tail_nhwc (conv2 -> conv3 + residual -> next conv1 chained through registers, resnet.py:189-205) is
    BIT-identical to the three dp_conv2d_nhwc launches it replaces, and both agree with torch in fp64."""
    from densepose_torchscript_amd.engine import Act
    e = eng[dt]
    N, H, W, with_next = case
    (l2, l3, l1), (w2, b2, w3, b3, w1, b1) = _tail_layers(e, N * 1000 + H * 10 + W)
    g = torch.Generator().manual_seed(H * 1000 + W)
    t1 = F.relu(torch.randn((N, 64, H, W), generator=g))
    res = torch.randn((N, 256, H, W), generator=g)
    t1, res = _round(t1, dt), _round(res, dt)
    ta = Act(_nhwc(t1, 64, e.tdt, e.device), N, H, W, 64)
    ra = Act(_nhwc(res, 256, e.tdt, e.device), N, H, W, 256)
    t2 = e.conv(l2, ta, relu=True)
    x_ref = e.conv(l3, t2, relu=True, residual=ra)
    n_ref = e.conv(l1, x_ref, relu=True)
    fused = e.bottleneck_tail(l2, l3, l1 if with_next else None, ta, ra)
    assert fused is not None, "the library must have a fused kernel for the res2 shape"
    x_f, n_f = fused
    torch.cuda.synchronize()
    assert torch.equal(x_f.t, x_ref.t)
    if with_next:
        assert torch.equal(n_f.t, n_ref.t)
    else:
        assert n_f is None
    # and against torch (fp64 math on the same rounded operands; intermediate tensors rounded where the kernels store them)
    rw = lambda w: _round(w, dt).double()  # noqa: E731
    y2 = _round(F.relu(F.conv2d(t1.double(), rw(w2), b2.double(), padding=1)).float(), dt).double()
    y3 = F.relu(F.conv2d(y2, rw(w3), b3.double()) + res.double())
    got = x_f.t.float().cpu().permute(0, 3, 1, 2).double()
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    assert bool(((got - y3).abs() <= 2 * ulp * y3.abs() + 2e-2).all()), float((got - y3).abs().max())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [(1, 20, 24), (2, 17, 19), (3, 40, 37), (1, 1, 1), (2, 3, 70), (1, 96, 160), (2, 200, 336)])
def test_bottleneck_tail_with_projection_shortcut(eng, dt, case):
    """First block of res2 (resnet.py:189-205 with the projection shortcut of :189-190): conv2 -> [conv3 | shortcut] over [t2 ; block
    input] in ONE launch (dp_bottleneck_params.sc_in) is BIT-identical to conv2 followed by the dual-source pointwise layer
    (dp_conv_params.in2), agrees with torch in fp64, and every image alone equals the image inside the batch."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import dual_source_pointwise
    e = eng[dt]
    N, H, W = case
    (l2, _, _), (w2, b2, w3, b3, _, _) = _tail_layers(e, N * 1000 + H * 10 + W)
    g = torch.Generator().manual_seed(H * 1000 + W + 5)
    ws = torch.randn((256, 64, 1, 1), generator=g) * (1.0 / 64) ** 0.5
    bs = torch.randn((256,), generator=g) * 0.5
    l3s = dual_source_pointwise("conv3+shortcut", w3.numpy(), b3.numpy(), 64, ws.numpy(), bs.numpy(), 64, 1, e.dt, e.device)
    t1 = _round(F.relu(torch.randn((N, 64, H, W), generator=g)), dt)
    x = _round(F.relu(torch.randn((N, 64, H, W), generator=g)), dt)
    ta = Act(_nhwc(t1, 64, e.tdt, e.device), N, H, W, 64)
    xa = Act(_nhwc(x, 64, e.tdt, e.device), N, H, W, 64)
    t2 = e.conv(l2, ta, relu=True)
    ref = e.conv(l3s, t2, relu=True, in2=xa)
    fused = e.bottleneck_tail(l2, l3s, None, ta, None, sc_in=xa)
    assert fused is not None, "the library must take the shortcut form of the res2 shape"
    x_f, n_f = fused
    torch.cuda.synchronize()
    assert n_f is None and torch.equal(x_f.t, ref.t)
    rw = lambda w: _round(w, dt).double()  # noqa: E731
    y2 = _round(F.relu(F.conv2d(t1.double(), rw(w2), b2.double(), padding=1)).float(), dt).double()
    y3 = F.relu(F.conv2d(y2, rw(w3), b3.double()) + F.conv2d(x.double(), rw(ws), bs.double()))
    got = x_f.t.float().cpu().permute(0, 3, 1, 2).double()
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    assert bool(((got - y3).abs() <= 2 * ulp * y3.abs() + 2e-2).all()), float((got - y3).abs().max())
    for i in range(N if N * H * W < 20000 else 1):
        one = e.bottleneck_tail(l2, l3s, None, Act(ta.t[i:i + 1].contiguous(), 1, H, W, 64), None, sc_in=Act(xa.t[i:i + 1].contiguous(), 1, H, W, 64))
        torch.cuda.synchronize()
        assert torch.equal(one[0].t[0], x_f.t[i]), i
    # no next-conv1 stage and no residual tensor in this form; the fp32 parity mode has no fused kernel at all
    assert e.bottleneck_tail(l2, l3s, _tail_layers(e, 1)[0][2], ta, None, sc_in=xa) is None
    p = __import__("densepose_torchscript_amd.lib", fromlist=["x"]).BottleneckParams()
    p.N, p.H, p.W, p.Cmid, p.Cout, p.Kpad2, p.Kpad3, p.ntaps2, p.k_order2, p.dtype = 1, 8, 8, 64, 256, 576, 128, 9, 1, e.dt
    p.hi_off2 = p.wi_off2 = -1
    p.Csc, p.sc_in = 64, 4096
    assert e.lib.dp_bottleneck_tail_supported(C.byref(p)) == 1
    p.next_t1 = 4096
    assert e.lib.dp_bottleneck_tail_supported(C.byref(p)) == 0
    p.next_t1, p.Kpad3 = None, 64
    assert e.lib.dp_bottleneck_tail_supported(C.byref(p)) == 0


PAIR_CASES = [
    # Cmid, N, H, W of a res3-shaped pair (128 -> 512 -> 128) / a res4-shaped one (256 -> 1024 -> 256)
    (128, 2, 100, 168),      # two frames at the headline geometry: 33600 pixels = 2100 steps over 256 workgroups (8 - 9 steps each)
    (128, 3, 13, 21),        # 819 pixels: fewer steps than workgroups, ragged last step
    (128, 1, 1, 1),          # a single pixel
    (128, 2, 37, 45),        # odd sizes, a few steps per workgroup
    (256, 8, 50, 84),        # res4 at the headline geometry: 33600 pixels = 2100 tiles: one round of 256 full jobs + 52 one-tile jobs
    (256, 1, 50, 84),        # one frame: 263 tiles over 256 jobs of one or two tiles
    (256, 3, 13, 21),        # 819 pixels: 52 jobs of one tile, ragged last tile
    (256, 1, 1, 1),          # a single pixel
    (256, 2, 37, 45),        # 209 tiles
    (256, 5, 64, 70),        # 1400 tiles: jobs of five and six tiles (uneven pixel blocks)
]


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", PAIR_CASES)
def test_bottleneck_pair_res3(eng, dt, case, policy):
    """dp_bottleneck_pair_nhwc: conv3 + residual + ReLU of a res3 / res4 block and conv1 + ReLU of the next one (resnet.py:199-205, :192-193;
    resnet.py:659-688 builds the blocks) in one launch - res3 with both matrices in registers (dp_pair.hip), res4 with both streamed
    through LDS in 64-channel chunks (dp_pair256.hip). The block output is BIT-identical to the separate conv3 launch; the next block's conv1
    output equals the separate launch up to its summation order and torch in fp64; every image alone equals the image inside the batch bit
    for bit (the job split depends on the pixel count, a pixel's arithmetic does not)."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    Cm, N, H, W = case
    Co = 4 * Cm
    if Cm == 256:
        assert e.bottleneck_pair is not None
        policy.set("pair256", "1")       # the res4 form is off by default (slower than the two launches at the benchmark geometry)
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W + Cm)
    mk = lambda co, ci: torch.randn((co, ci, 1, 1), generator=g) * (1.0 / ci) ** 0.5  # noqa: E731
    w3, w1 = _round(mk(Co, Cm), dt), _round(mk(Cm, Co), dt)
    b3, b1 = torch.randn((Co,), generator=g) * 0.5, torch.randn((Cm,), generator=g) * 0.5
    l3 = conv_from_oihw("conv3", w3.numpy(), b3.numpy(), Cm, 1, 0, 1, e.dt, e.device)
    l1 = conv_from_oihw("conv1n", w1.numpy(), b1.numpy(), Co, 1, 0, 1, e.dt, e.device)
    t2 = _round(F.relu(torch.randn((N, Cm, H, W), generator=g)), dt)
    res = _round(torch.randn((N, Co, H, W), generator=g), dt)
    ta = Act(_nhwc(t2, Cm, e.tdt, e.device), N, H, W, Cm)
    ra = Act(_nhwc(res, Co, e.tdt, e.device), N, H, W, Co)
    x_ref = e.conv(l3, ta, relu=True, residual=ra)
    n_ref = e.conv(l1, x_ref, relu=True)
    fused = e.bottleneck_pair(l3, l1, ta, ra)
    assert fused is not None, "the library must have a fused kernel for the res3 / res4 shapes"
    x_f, n_f = fused
    torch.cuda.synchronize()
    # the block output: one chain over K from the bias, + residual, ReLU - the separate conv3 launch's order in both forms
    assert torch.equal(x_f.t, x_ref.t)
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    d = (n_f.t.float() - n_ref.t.float()).abs()
    assert bool((d <= 2 * ulp * n_ref.t.float().abs() + 1e-3).all()), float(d.max())
    y3 = _round(F.relu(F.conv2d(t2.double(), w3.double(), b3.double()) + res.double()).float(), dt).double()
    y1 = F.relu(F.conv2d(y3, w1.double(), b1.double()))
    got = n_f.t.float().cpu().permute(0, 3, 1, 2).double()
    assert bool(((got - y1).abs() <= ulp * y1.abs() + 2e-3).all()), float((got - y1).abs().max())
    for i in sorted({0, N - 1}):
        one = e.bottleneck_pair(l3, l1, Act(ta.t[i:i + 1].contiguous(), 1, H, W, Cm), Act(ra.t[i:i + 1].contiguous(), 1, H, W, Co))
        torch.cuda.synchronize()
        assert torch.equal(one[0].t[0], x_f.t[i]) and torch.equal(one[1].t[0], n_f.t[i]), i
    # other widths and the fp32 parity mode have no fused kernel: the helper says so
    assert eng["fp32"].bottleneck_pair(l3, l1, ta, ra) is None


def test_bottleneck_tail_unsupported_shapes_fall_back(eng):
    """fp32 parity mode and non-res2 widths have no fused kernel: the engine helper says so instead of launching."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng["fp32"]
    (l2, l3, l1), _ = _tail_layers(e, 1)
    ta = Act(torch.zeros((1, 8, 8, 64), device=e.device), 1, 8, 8, 64)
    ra = Act(torch.zeros((1, 8, 8, 256), device=e.device), 1, 8, 8, 256)
    assert e.bottleneck_tail(l2, l3, l1, ta, ra) is None
    eb = eng["bf16"]
    g = torch.Generator().manual_seed(2)
    w2 = torch.randn((32, 32, 3, 3), generator=g)
    w3 = torch.randn((128, 32, 1, 1), generator=g)
    l2s = conv_from_oihw("c2", w2.numpy(), np.zeros(32, np.float32), 32, 1, 1, 1, eb.dt, eb.device)
    l3s = conv_from_oihw("c3", w3.numpy(), np.zeros(128, np.float32), 32, 1, 0, 1, eb.dt, eb.device)
    ta = Act(torch.zeros((1, 8, 8, 32), dtype=eb.tdt, device=eb.device), 1, 8, 8, 32)
    ra = Act(torch.zeros((1, 8, 8, 128), dtype=eb.tdt, device=eb.device), 1, 8, 8, 128)
    assert eb.bottleneck_tail(l2s, l3s, None, ta, ra) is None
    p = __import__("densepose_torchscript_amd.lib", fromlist=["x"]).BottleneckParams()
    p.N, p.H, p.W, p.Cmid, p.Cout, p.Kpad2, p.Kpad3, p.ntaps2, p.dtype = 1, 8, 8, 32, 128, 288, 64, 9, eb.dt
    assert eb.lib.dp_bottleneck_tail_nhwc(C.byref(p), eb._stream()) == -2   # DP_ERR_UNSUPPORTED


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 61, 90, 64, 96), (1, 64, 96, 64, 96), (3, 33, 37, 64, 64), (1, 32, 32, 32, 32)])
def test_paired_preprocess_and_stem(eng, dt, shape):
    """preprocess_image (rcnn.py:156-181: (x - mean) / std, zero pad to /32) written in the paired-pixel layout + the stem
    (resnet.py:350-353: 7x7 stride 2 pad 3 + FrozenBN + ReLU) as a 7 x 4-tap stride-(2,1) convolution over 8-channel cells,
    against F.conv2d on the plainly normalised, padded image."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import stem_paired_conv
    e = eng[dt]
    n, h, w, Hp, Wp = shape
    g = torch.Generator().manual_seed(n * 100 + h + w)
    img = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    mean, std = (103.53, 116.28, 123.675), (1.0, 57.375, 58.395)
    x = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    xp = torch.zeros((n, 3, Hp, Wp))
    xp[:, :, :h, :w] = x
    # the paired buffer itself: cell j = pixels 2j - 3, 2j - 2 (4 channels each, the 4th zero)
    Wq = Wp // 2 + 3
    buf = torch.full((n, Hp, Wq, 8), 7.0, dtype=e.tdt, device=e.device)
    p = L.PreprocessParams()
    src = img.to(e.device)
    p.src, p.dst, p.paired = src.data_ptr(), buf.data_ptr(), 1
    p.n_img, p.h, p.w, p.Hp, p.Wp, p.dtype = n, h, w, Hp, Wp, e.dt
    for i in range(3):
        p.mean[i], p.std[i] = mean[i], std[i]
    L.check(e.lib.dp_preprocess_u8(C.byref(p), e._stream()), "dp_preprocess_u8")
    cells = buf.float().cpu().view(n, Hp, 2 * Wq, 4)
    want = torch.zeros((n, Hp, 2 * Wq, 4))
    want[:, :, 3:3 + Wp, :3] = xp.permute(0, 2, 3, 1)
    ulp = {"fp32": 0.0, "bf16": 2.0 ** -8, "fp16": 2.0 ** -11}[dt]
    assert bool(((cells - want).abs() <= ulp * want.abs() + 1e-6).all())
    # the same frames handed over interleaved ([n, h, w, 3], src_hwc): bit-identical buffer (the scale-1 path that skips the resize)
    buf2 = torch.full((n, Hp, Wq, 8), 9.0, dtype=e.tdt, device=e.device)
    src2 = img.permute(0, 2, 3, 1).contiguous().to(e.device)
    p.src, p.dst, p.src_hwc = src2.data_ptr(), buf2.data_ptr(), 1
    L.check(e.lib.dp_preprocess_u8(C.byref(p), e._stream()), "dp_preprocess_u8")
    assert torch.equal(buf, buf2)
    p.paired = 0
    assert e.lib.dp_preprocess_u8(C.byref(p), e._stream()) == -1      # interleaved frames need the paired layout
    # stem on it
    wt = torch.randn((16, 3, 7, 7), generator=g) * 0.05
    b = torch.randn((16,), generator=g)
    if dt != "fp32":
        wt = _round(wt, dt)
    layer = stem_paired_conv("stem", wt.numpy(), b.numpy(), e.dt, e.device)
    out = e.conv(layer, Act(buf, n, Hp, Wq, 8), relu=True, out_hw=(Hp // 2, Wp // 2), out_f32=True)
    torch.cuda.synchronize()
    xin = want[:, :, 3:3 + Wp, :3].permute(0, 3, 1, 2) if dt == "fp32" else cells[:, :, 3:3 + Wp, :3].permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(xin.double(), wt.double(), b.double(), stride=2, padding=3))
    got = out.t.cpu()[..., :16].permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())) * (147 ** 0.5)
    assert layer.kpad == (224 if dt == "fp32" else 256) and layer.macs_per_pixel == 16 * 147   # 7 x 4 taps x 8, padded to 128 B


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("tp", [None, "4", "5", "6", "7"])          # every tile height of the 256-cout ring kernel
@pytest.mark.parametrize("shape", [(1, 40, 56), (2, 13, 21), (3, 9, 11), (1, 50, 84)])
def test_conv_with_fused_rpn_head(eng, dt, tp, shape, policy):
    """RPN head (rpn.py:168-171: 3x3 conv + ReLU, then the 1x1 objectness / anchor-delta convolutions) in ONE launch: the head is
    applied in the ring kernel's epilogue and the 256-channel hidden tensor is never written. BIT-identical to the two-launch
    form (same rounded hidden tensor, head accumulated over the K planes in the same order - the four 64-channel blocks are
    chained through the accumulator), and close to torch in fp64."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    policy.set("conv_big", "1")       # the 256-cout ring kernel whatever the tile count
    if tp:
        policy.set("conv_tp", tp)
    N, H, W = shape
    g = torch.Generator().manual_seed(N * 100 + H + W)
    x = _round(torch.randn((N, 256, H, W), generator=g), dt)
    w = _round(torch.randn((256, 256, 3, 3), generator=g) * (1.0 / 2304) ** 0.5, dt)
    b = torch.randn((256,), generator=g) * 0.3
    wh = _round(torch.randn((15, 256, 1, 1), generator=g) * (1.0 / 256) ** 0.5, dt)
    bh = torch.randn((15,), generator=g)
    conv = conv_from_oihw("rpn_conv", w.numpy(), b.numpy(), 256, 1, 1, 1, e.dt, e.device)
    headl = conv_from_oihw("rpn_head", wh.numpy(), bh.numpy(), 256, 1, 0, 1, e.dt, e.device)
    whp = torch.zeros((16, 256))
    whp[:15] = wh.view(15, 256)
    bhp = torch.zeros((16,))
    bhp[:15] = bh
    xa = Act(_nhwc(x, 256, e.tdt, e.device), N, H, W, 256)
    assert e.head_fusable(conv, xa)
    hidden = e.conv(conv, xa, relu=True)
    want = e.conv(headl, hidden, out_f32=True)
    got = e.conv(conv, xa, relu=True, head=(whp.to(e.tdt).to(e.device), bhp.to(e.device), 15 * 256))
    torch.cuda.synchronize()
    assert got.t.shape == want.t.shape == (N, H, W, 16) and got.t.dtype == torch.float32
    assert torch.equal(got.t, want.t)
    assert float(got.t[..., 15].abs().max()) == 0.0
    hid = _round(F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).float(), dt).double()
    ref = F.conv2d(hid, wh.double(), bh.double()).permute(0, 2, 3, 1)
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    assert float((got.t[..., :15].double().cpu() - ref).abs().max()) <= 4 * ulp * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 61, 90, 64, 96), (1, 64, 96, 64, 96), (3, 33, 37, 64, 64), (1, 30, 500, 32, 512), (1, 200, 230, 224, 256)])
def test_stem_pool_fused_equals_conv_then_pool(eng, dt, shape):
    """dp_stem_pool_nhwc (resnet.py:350-354: conv 7x7/2 + FrozenBN + ReLU + max_pool2d(3, 2, 1) in one launch) is BIT-identical to
    dp_conv2d_nhwc + dp_maxpool3x3s2_nhwc, and agrees with torch. Shapes: several column strips (> 56 pooled columns), ragged
    last strip, more pooled rows than one job, image narrower / shorter than its padded size."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import stem_paired_conv
    e = eng[dt]
    n, h, w, Hp, Wp = shape
    g = torch.Generator().manual_seed(n * 100 + h + w)
    img = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    mean, std = (103.53, 116.28, 123.675), (1.0, 1.0, 1.0)
    Wq = Wp // 2 + 3
    buf = torch.empty((n, Hp, Wq, 8), dtype=e.tdt, device=e.device)
    p = L.PreprocessParams()
    src = img.to(e.device)
    p.src, p.dst, p.paired = src.data_ptr(), buf.data_ptr(), 1
    p.n_img, p.h, p.w, p.Hp, p.Wp, p.dtype = n, h, w, Hp, Wp, e.dt
    for i in range(3):
        p.mean[i], p.std[i] = mean[i], std[i]
    L.check(e.lib.dp_preprocess_u8(C.byref(p), e._stream()), "dp_preprocess_u8")
    wt = _round(torch.randn((64, 3, 7, 7), generator=g) * 0.01, dt)
    b = torch.randn((64,), generator=g) * 20
    layer = stem_paired_conv("stem", wt.numpy(), b.numpy(), e.dt, e.device)
    xa = Act(buf, n, Hp, Wq, 8)
    conv = e.conv(layer, xa, relu=True, out_hw=(Hp // 2, Wp // 2))
    Ho, Wo = (Hp // 2 - 1) // 2 + 1, (Wp // 2 - 1) // 2 + 1
    want = torch.empty((n, Ho, Wo, 64), dtype=e.tdt, device=e.device)
    assert e.lib.dp_maxpool3x3s2_nhwc(conv.t.data_ptr(), want.data_ptr(), n, Hp // 2, Wp // 2, 64, e.dt, e._stream()) == 0
    got = e.stem_pool(layer, xa)
    assert got is not None, "the library must have a fused kernel for the 64-channel stem"
    torch.cuda.synchronize()
    assert got.t.shape == want.shape
    assert torch.equal(got.t, want)
    # and against torch on the same rounded operands
    cells = buf.float().cpu().view(n, Hp, 2 * Wq, 4)[:, :, 3:3 + Wp, :3].permute(0, 3, 1, 2)
    ref = F.max_pool2d(_round(F.relu(F.conv2d(cells.double(), wt.double(), b.double(), stride=2, padding=3)).float(), dt), 3, 2, 1)
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    gd = got.t.float().cpu().permute(0, 3, 1, 2)
    assert bool(((gd - ref).abs() <= 2 * ulp * ref.abs() + 1e-3).all()), float((gd - ref).abs().max())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(128, 1, 50, 84, True), (128, 2, 37, 45, True), (128, 3, 33, 32, False), (128, 1, 100, 168, True),
                                   (128, 2, 64, 35, False), (128, 1, 47, 130, True), (256, 1, 50, 84, True), (256, 2, 37, 45, False),
                                   (256, 3, 25, 42, True), (256, 1, 100, 168, False), (256, 2, 64, 35, True), (256, 1, 47, 130, True),
                                   (256, 8, 13, 32, True), (128, 16, 9, 17, False)])
def test_conv3x3_weight_stationary_equals_ring_kernel(eng, dt, shape, policy):
    """The weight-stationary 3x3 kernels (weights in registers, pixel rows in an LDS ring; C -> C channels, C = 128: res3 conv2,
    resnet.py:195-197; C = 256: res4 conv2, FPN outputs, decoder) against the LDS-ring kernel they replace - bit-identical
    (same K order, for C = 256 chained through two waves) - and torch in fp64. Odd heights (a ragged last step), widths that are
    no multiple of the 16-pixel strip, several images (steps flow across column and image boundaries), more workgroups than steps."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    Cc, N, H, W, relu = shape
    policy.default("conv_ws")
    policy.set("conv_wsq", "0")      # (the 256-channel layers run on kernel class 10 by default: test_conv3x3_one_wave_per_simd_kernel below)
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W + Cc)
    x = _round(torch.randn((N, Cc, H, W), generator=g), dt)
    w = _round(torch.randn((Cc, Cc, 3, 3), generator=g) * (1.0 / (9 * Cc)) ** 0.5, dt)
    b = torch.randn((Cc,), generator=g) * 0.3
    layer = conv_from_oihw("conv2", w.numpy(), b.numpy(), Cc, 1, 1, 1, e.dt, e.device)
    xa = Act(_nhwc(x, Cc, e.tdt, e.device), N, H, W, Cc)
    p = L.ConvParams()
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = N, H, W, Cc, H, W, Cc, Cc, 9 * Cc
    p.stride, p.ntaps, p.dtype, p.hi_off, p.wi_off = 1, 9, e.dt, -1, -1
    p.osN, p.osH, p.osW = H * W * Cc, W * Cc, Cc
    p.out = 4096
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == 6
    p.shared_chip = 1      # the host's hint "this launch runs beside other streams": same kernel, twice as many workgroups
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == 6
    p.shared_chip = 0
    got = e.conv(layer, xa, relu=relu)
    e._shared_chip = True
    try:
        hinted = e.conv(layer, xa, relu=relu)
    finally:
        e._shared_chip = False
    assert torch.equal(got.t, hinted.t)
    policy.set("conv_ws", "0")
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) != 6
    want = e.conv(layer, xa, relu=relu)
    torch.cuda.synchronize()
    assert torch.equal(got.t, want.t)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = F.relu(ref) if relu else ref
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    gd = got.t.float().cpu().permute(0, 3, 1, 2).double()
    assert bool(((gd - ref).abs() <= ulp * ref.abs() + 2e-3).all()), float((gd - ref).abs().max())


@pytest.mark.parametrize("cin,cout,res", [(1024, 256, False), (256, 1024, True), (2048, 512, False), (512, 256, "up")])
def test_conv1x1_weight_stationary_pointwise_chunks_large_batches(eng, cin, cout, res, policy):
    """Kernel class 9 is chosen by the channel counts alone (dp_conv_pws_ok looks at ONE image's size): a batch whose tensors pass the 32-bit
    offset range is cut into image chunks inside dp_conv_pws_launch. With the limit lowered to two images the bits are the unchunked launch's,
    with a linear residual and with the FPN's nearest x2 top-down map (per-image residual pointers)."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng["bf16"]
    N, H, W = 5, 12, 22
    g = torch.Generator().manual_seed(cin + cout)
    x = Act(_round(torch.randn((N, H, W, cin), generator=g), "bf16").to(e.tdt).to(e.device), N, H, W, cin)
    w = _round(torch.randn((cout, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5, "bf16")
    layer = conv_from_oihw("c", w.numpy(), np.zeros(cout, np.float32), cin, 1, 0, 1, e.dt, e.device)
    r, rshift = None, 0
    if res == "up":
        r, rshift = Act(_round(torch.randn((N, H // 2, W // 2, cout), generator=g), "bf16").to(e.tdt).to(e.device), N, H // 2, W // 2, cout), 1
    elif res:
        r = Act(_round(torch.randn((N, H, W, cout), generator=g), "bf16").to(e.tdt).to(e.device), N, H, W, cout)
    p = L.ConvParams()
    p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = H, W, cin, H, W, cout, layer.cout_w, layer.kpad
    p.stride, p.ntaps, p.dtype = 1, 1, e.dt
    p.osN, p.osH, p.osW = H * W * cout, W * cout, cout
    p.out = 4096
    for n in (1, 5, 4000):
        p.N = n
        assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == 9, n       # whatever the batch: also one whose tensors are far beyond 2 GiB
    whole = e.conv(layer, x, relu=True, residual=r, rshift=rshift).t.clone()
    policy.set("rows_chunk_bytes", str(2 * (H * W + 64) * max(cin, cout) * 2))
    parts = e.conv(layer, x, relu=True, residual=r, rshift=rshift).t
    torch.cuda.synchronize()
    assert torch.equal(whole, parts)


WSQ_CASES = [(1, 13, 21, True), (2, 9, 17, False), (3, 37, 45, True), (1, 8, 16, True), (2, 25, 42, True), (1, 50, 84, True), (2, 64, 35, False),
             (1, 47, 130, True), (8, 13, 32, True), (1, 4, 40, True), (2, 3, 50, True), (1, 1, 130, False), (6, 16, 25, True), (2, 100, 168, False)]


@pytest.mark.parametrize("mfma", [16, 32])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", WSQ_CASES)
def test_conv3x3_one_wave_per_simd_kernel(eng, dt, case, mfma, policy):
    """Kernel class 10 (dp_conv_wq.hip): the 256 -> 256 3x3 layers (res4 conv2 resnet.py:195-197, FPN outputs fpn.py:134-157, the decoder's
    scale heads roi_head.py:48-68) with ONE wave per SIMD - conv3x3_ws1_kernel on v_mfma_f32_16x16x32 (default) and conv3x3_wsq_kernel on
    v_mfma_f32_32x32x16 (policy key wsq_shape). Against torch in fp64 on operands rounded to the storage type, against kernel class 6
    within two rounding steps (another summation order: K halves of two waves added through LDS), every image alone == in the batch
    bit for bit, the scheduling hints (other splits of the steps over workgroups) bit for bit, and the class is a function of the
    per-image geometry alone: the same for N = 1 and N = 64, not taken when the caller pins the ring family's order (ring_order)."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    N, H, W, relu = case
    Cc = 256
    policy.set("wsq_shape", str(mfma))
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W + mfma)
    x = _round(torch.randn((N, Cc, H, W), generator=g), dt)
    w = _round(torch.randn((Cc, Cc, 3, 3), generator=g) * (1.0 / (9 * Cc)) ** 0.5, dt)
    b = torch.randn((Cc,), generator=g) * 0.3
    layer = conv_from_oihw("conv2", w.numpy(), b.numpy(), Cc, 1, 1, 1, e.dt, e.device)
    xa = Act(_nhwc(x, Cc, e.tdt, e.device), N, H, W, Cc)
    p = L.ConvParams()
    p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = H, W, Cc, H, W, Cc, Cc, 9 * Cc
    p.stride, p.ntaps, p.dtype, p.hi_off, p.wi_off = 1, 9, e.dt, -1, -1
    p.osN, p.osH, p.osW = H * W * Cc, W * Cc, Cc
    p.out = 4096
    want_cls = 10 if H * W >= 128 else None
    for n in (1, N, 64):
        p.N = n
        for hint in (0, 1, 2):
            p.shared_chip = hint
            k = e.lib.dp_conv2d_kernel_class(C.byref(p))
            assert (k == 10) == (want_cls == 10), (n, hint, k)
    p.shared_chip, p.ring_order = 0, 1
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) != 10
    if want_cls is None:
        return
    got = e.conv(layer, xa, relu=relu)
    for hint in (1, 2):
        e._shared_chip = hint
        try:
            assert torch.equal(got.t, e.conv(layer, xa, relu=relu).t), hint
        finally:
            e._shared_chip = False
    for i in range(N):
        one = e.conv(layer, Act(xa.t[i:i + 1].contiguous(), 1, H, W, Cc), relu=relu)
        assert torch.equal(one.t[0], got.t[i]), i
    pinned = e.conv(layer, xa, relu=relu, ring_order=True)
    policy.set("conv_wsq", "0")
    older = e.conv(layer, xa, relu=relu)
    torch.cuda.synchronize()
    assert torch.equal(pinned.t, older.t)          # ring_order = the bits of classes 2 - 6
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    d = (got.t.float() - older.t.float()).abs()
    assert bool((d <= 2 * ulp * older.t.float().abs() + 1e-3).all()), float(d.max())
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = F.relu(ref) if relu else ref
    gd = got.t.float().cpu().permute(0, 3, 1, 2).double()
    assert bool(((gd - ref).abs() <= ulp * ref.abs() + 2e-3).all()), float((gd - ref).abs().max())


def test_conv3x3_one_wave_per_simd_kernel_chunks_large_batches(eng, policy):
    """A batch whose tensors pass the 32-bit offset range is cut into image chunks INSIDE the launch function (the class never depends on N):
    with the chunk limit lowered to three images the bits are those of the unchunked launch."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng["bf16"]
    N, H, W, Cc = 7, 20, 30, 256
    g = torch.Generator().manual_seed(77)
    x = _round(torch.randn((N, Cc, H, W), generator=g), "bf16")
    w = _round(torch.randn((Cc, Cc, 3, 3), generator=g) * (1.0 / (9 * Cc)) ** 0.5, "bf16")
    layer = conv_from_oihw("c", w.numpy(), np.zeros(Cc, np.float32), Cc, 1, 1, 1, e.dt, e.device)
    xa = Act(_nhwc(x, Cc, e.tdt, e.device), N, H, W, Cc)
    whole = e.conv(layer, xa, relu=True).t.clone()
    policy.set("rows_chunk_bytes", str(3 * H * W * Cc * 2))
    parts = e.conv(layer, xa, relu=True).t
    torch.cuda.synchronize()
    assert torch.equal(whole, parts)


ROWS_CASES = [
    # Cin, Cout, N images, H, W, live images (None = all), relu
    (512, 512, 8, 28, 28, None, True),      # DensePose head, v1convx.py:44-59 (two whole strip groups of four ROIs)
    (512, 512, 7, 28, 28, 5, True),         # ragged group, device-side live count below N
    (512, 512, 3, 28, 28, 0, True),         # R = 0: nothing live, nothing written
    (512, 512, 1, 28, 28, None, False),     # one ROI (DeepLab head: no ReLU, deeplab.py:52)
    (256, 512, 6, 28, 28, None, True),      # body_conv_fcn1: 256 -> 512
    (256, 512, 9, 14, 14, 7, True),         # legacy pooler resolution: strips of 14 + 2
    (512, 512, 2, 25, 42, None, True),      # res5 conv2 at 800 x 1344 (resnet.py:195-197): groups of eight images
    (512, 512, 9, 25, 42, None, True),
    (512, 512, 1, 1, 16, None, True),       # one row, one strip
    (512, 512, 2, 2, 33, None, False),
    (512, 96, 3, 5, 17, None, True),        # three cout slices only: more workgroups than rows
    (256, 512, 2, 37, 48, None, True),
    (512, 512, 70, 28, 28, 64, True),       # the benchmark's 64 ROIs inside a larger slot count
    (512, 512, 9, 28, 28, None, True),      # a ragged group of the 32-pixel form (groups of eight ROIs)
    (512, 96, 5, 25, 40, None, True),       # 40 columns: groups of four images, strips of 32 + (8 | 24), (16 | 16), ...
    (512, 512, 3, 9, 56, 2, False),
    (512, 160, 2, 30, 32, None, True),      # whole strips only
    (512, 512, 5, 7, 24, None, True),
    (256, 256, 8, 50, 84, None, True),      # res4 conv2 / FPN p4 geometry on the 64-cout 32-pixel form (DP_CONV_ROWS2_256=1)
    (256, 256, 2, 25, 42, None, False),     # groups of 16 images, two live: strips counted from the live columns
    (256, 128, 3, 11, 168, 2, True),
]


def _rows2_width_ok(W, max_group=8):
    """mirror of rows2_width_ok / rows2_ok (dp_conv_rows.hip): every 32-pixel strip at most two segments; the 512-channel policy also
    wants strip groups of at most 8 images"""
    import math
    if W < 16:
        return False
    G = 32 // math.gcd(W, 32)
    if G > max_group:
        return False
    for k in range(G * W // 32):
        c00 = (32 * k) % W
        if 32 - min(W - c00, 32) > W:
            return False
    return True


# rows: the 16-pixel form; rows2: the 32-pixel form (4 K quarters x 2 cout halves) that the default policy picks where the width allows;
# rows2_lockstep: the same on one schedule for all waves (policy key conv_rows2_lockstep). (The barrier-free chain variant of round 4
# measured slower and is compiled into `make exp` builds only.)
@pytest.mark.parametrize("variant", ["rows", "rows2", "rows2_lockstep"])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ROWS_CASES)
def test_conv3x3_rows_kernel(eng, dt, case, variant, policy):
    """The row-streaming K-split weight-stationary 3x3 kernel (dp_conv_rows.hip, kernel class 7: the DensePose head's
    body_conv_fcn1..8, v1convx.py:44-59 / deeplab.py:64-74, and res5's conv2, resnet.py:195-197) against torch in fp64 on operands
    rounded to the storage type, against the LDS-ring kernel (same products, another summation order), and against itself image by
    image: an image's result must not depend on what else is in the batch, on the strip group it lands in or on the device-side
    live count."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    Ci, Co, N, H, W, live, relu = case
    two_ok = (Ci == 512 and _rows2_width_ok(W)) or (Ci == 256 and _rows2_width_ok(W, 1 << 30))
    policy.default("conv_rows2_lockstep")
    if variant.startswith("rows2"):
        if not two_ok:
            pytest.skip("the 32-pixel form takes 512 input channels and widths with strip groups of at most 8 images")
        policy.default("conv_rows2")
        policy.set("conv_rows2_256", "1")     # every 256-channel layer (default: only those the weight-stationary kernel does not take)
        if variant == "rows2_lockstep":
            policy.set("conv_rows2_lockstep", "1")
    else:
        policy.set("conv_rows2", "0")
    policy.set("conv_rows", "2")     # every shape the kernels take
    g = torch.Generator().manual_seed(Ci + Co + N * 1000 + H * 10 + W)
    x = _round(torch.randn((N, Ci, H, W), generator=g), dt)
    w = _round(torch.randn((Co, Ci, 3, 3), generator=g) * (1.0 / (9 * Ci)) ** 0.5, dt)
    b = torch.randn((Co,), generator=g) * 0.3
    layer = conv_from_oihw("fcn", w.numpy(), b.numpy(), Ci, 1, 1, 1, e.dt, e.device)
    xa = Act(_nhwc(x, Ci, e.tdt, e.device), N, H, W, Ci)
    p = L.ConvParams()
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = N, H, W, Ci, H, W, Co, layer.cout_w, 9 * Ci
    p.stride, p.ntaps, p.dtype, p.hi_off, p.wi_off = 1, 9, e.dt, -1, -1
    p.osN, p.osH, p.osW = H * W * Co, W * Co, Co
    p.out = 4096
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == (8 if variant.startswith("rows2") else 7)
    n_dev = None if live is None else torch.tensor([live], dtype=torch.int32, device=e.device)
    # the default policy: a call site is on this kernel class for every batch or for none - 512-channel layers whose width suits the
    # 32-pixel form always (the DensePose head's ROI maps, sized on the device or not), other 512-channel layers on plain tensors
    # (res5's conv2) on the 16-pixel form; the 256 -> 512 layer and device-sized launches of other widths are not
    policy.default("conv_rows")
    p.n_dev = None if n_dev is None else n_dev.data_ptr()
    if variant.startswith("rows2"):
        assert e.lib.dp_conv2d_kernel_class(C.byref(p)) == 8
        # 256 input channels: by default only the layers with another cout count (the head's 256 -> 512 first layer)
        policy.default("conv_rows2_256")
        assert (e.lib.dp_conv2d_kernel_class(C.byref(p)) == 8) == (Ci == 512 or Co != 256)
        policy.set("conv_rows2_256", "1")
    else:
        assert (e.lib.dp_conv2d_kernel_class(C.byref(p)) == 7) == (Ci == 512 and live is None)     # (DP_CONV_ROWS2=0 in this variant)
    p.n_dev = None
    policy.set("conv_rows", "2")
    nl = N if live is None else live
    out = torch.full((N, H, W, Co), 7.0, dtype=e.tdt, device=e.device)      # slots behind the live count must stay untouched
    got = e.conv(layer, xa, relu=relu, out=out, n_dev=n_dev)
    torch.cuda.synchronize()
    assert bool((got.t[nl:].float() == 7.0).all())
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = F.relu(ref) if relu else ref
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    gd = got.t[:nl].float().cpu().permute(0, 3, 1, 2).double()
    assert bool(((gd - ref[:nl]).abs() <= ulp * ref[:nl].abs() + 2e-3).all()), float((gd - ref[:nl]).abs().max())
    # every image alone == the image inside the batch, bit for bit
    for i in sorted({0, nl // 2, nl - 1} & set(range(nl))):
        one = e.conv(layer, Act(xa.t[i:i + 1].contiguous(), 1, H, W, Ci), relu=relu)
        assert torch.equal(one.t[0], got.t[i]), i
    # a launch whose tensors exceed the 32-bit offset range goes as several launches over chunks of whole strip groups (never to
    # another kernel class): the limit lowered to a few images, same bits, same untouched slots, the live count cut across chunks
    policy.set("rows_chunk_bytes", str(3 * H * W * max(Ci, Co) * 2))
    out2 = torch.full((N, H, W, Co), 7.0, dtype=e.tdt, device=e.device)
    got2 = e.conv(layer, xa, relu=relu, out=out2, n_dev=n_dev)
    torch.cuda.synchronize()
    assert torch.equal(got2.t, got.t)
    policy.default("rows_chunk_bytes")
    # the ring kernels on the same operands: equal up to the summation order
    policy.set("conv_rows", "0")
    assert e.lib.dp_conv2d_kernel_class(C.byref(p)) not in (7, 8)
    want = e.conv(layer, xa, relu=relu)
    torch.cuda.synchronize()
    d = (want.t[:nl].float() - got.t[:nl].float()).abs()
    assert bool((d <= 2 * ulp * want.t[:nl].float().abs() + 1e-3).all()), float(d.max())


@pytest.mark.parametrize("kernel", ["ws1", "wsq32", "wsr"])     # kernel class 10 in its two MFMA shapes, kernel class 6
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("shape", [(1, 50, 84), (2, 24, 46), (3, 30, 26), (1, 100, 168), (8, 14, 22)])
def test_conv3x3_post_activation_sum(eng, dt, mode, shape, kernel, policy):
    """dp_conv_params.post_res (the decoder's level sum in the conv epilogue, roi_head.py:71-79): out = relu(conv(x) + b) + post
    (mode 1, same geometry) and out = relu(conv(x) + b) + bilinear_x2(post) (mode 2, half-size map, align_corners=False) on the
    weight-stationary 3x3 kernel, against torch in fp64 - ragged heights / strip widths, several images, more workgroups than
    steps; exact up to the one rounding to the storage type."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    N, H, W = shape
    Cc = 256
    if kernel == "wsr":
        if N * H * W < 2048:
            pytest.skip("class 6 takes a post tensor from 2048 output pixels per launch on")
        policy.set("conv_wsq", "0")
    else:
        policy.set("wsq_shape", "32" if kernel == "wsq32" else "16")
    g = torch.Generator().manual_seed(1000 * mode + H * W + N)
    x = _round(torch.randn((N, Cc, H, W), generator=g), dt)
    w = _round(torch.randn((Cc, Cc, 3, 3), generator=g) * (2.0 / (Cc * 9)) ** 0.5, dt)
    b = torch.randn((Cc,), generator=g) * 0.1
    hp, wp = (H, W) if mode == 1 else (H // 2, W // 2)
    post = _round(torch.randn((N, Cc, hp, wp), generator=g), dt)
    layer = conv_from_oihw("t", w.numpy(), b.numpy(), Cc, 1, 1, 1, e.dt, e.device)
    xa = Act(_nhwc(x, Cc, e.tdt, e.device), N, H, W, Cc)
    pa = Act(_nhwc(post, Cc, e.tdt, e.device), N, hp, wp, Cc)
    assert e.post_fusable(layer, xa, mode)
    got = e.conv(layer, xa, relu=True, post=pa, post_mode=mode)
    hinted = []
    for hint in (1, 2):      # the host's scheduling hints move work between workgroups, never bits
        e._shared_chip = hint
        try:
            hinted.append(e.conv(layer, xa, relu=True, post=pa, post_mode=mode))
        finally:
            e._shared_chip = False
    assert all(torch.equal(got.t, h.t) for h in hinted)
    plain = e.conv(layer, xa, relu=True)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    ref = ref + (post.double() if mode == 1 else F.interpolate(post.double(), scale_factor=2.0, mode="bilinear", align_corners=False))
    gd = got.t.float().cpu().permute(0, 3, 1, 2).double()
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    assert bool(((gd - ref).abs() <= ulp * ref.abs() + 2e-4).all()), float((gd - ref).abs().max())
    # ... and the conv part is the plain launch's: subtracting the post term in fp64 leaves relu(conv) up to the two roundings
    pd = plain.t.float().cpu().permute(0, 3, 1, 2).double()
    add = post.double() if mode == 1 else F.interpolate(post.double(), scale_factor=2.0, mode="bilinear", align_corners=False)
    assert bool(((gd - add - pd).abs() <= ulp * (ref.abs() + pd.abs()) + 2e-4).all())
    # unsupported combinations are refused, not ignored
    from densepose_torchscript_amd import lib as L
    with pytest.raises(L.DensePoseHipError):
        e.conv(layer, xa, relu=False, post=pa, post_mode=mode)


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("shape", [(2, 128, 256, 512, 25, 42, 2), (1, 256, 512, 1024, 13, 21, 2), (3, 64, 192, 256, 17, 9, 1),
                                   (8, 128, 256, 512, 50, 84, 2), (1, 512, 1024, 2048, 7, 11, 2)])
def test_conv_pointwise_two_sources(eng, dt, shape, policy):
    """dp_conv_params.in2: out = relu(W1 x1 + W2 x2[::s, ::s] + b) as one pointwise layer whose K axis is x1's channels followed
    by x2's (conv3 of a stage's first bottleneck block + its projection shortcut, resnet.py:189-205), on both LDS-ring kernels,
    against torch in fp64."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import dual_source_pointwise
    e = eng[dt]
    N, c1, c2, co, H, W, s2 = shape
    g = torch.Generator().manual_seed(c1 + c2 + H * W)
    x1 = torch.randn((N, c1, H, W), generator=g)
    H2, W2 = (H - 1) * s2 + 1 + (s2 - 1), (W - 1) * s2 + 1          # a source that is not an exact multiple of the stride
    x2 = torch.randn((N, c2, H2, W2), generator=g)
    w1 = torch.randn((co, c1, 1, 1), generator=g) * (1.0 / c1) ** 0.5
    w2 = torch.randn((co, c2, 1, 1), generator=g) * (1.0 / c2) ** 0.5
    b1, b2 = torch.randn((co,), generator=g), torch.randn((co,), generator=g)
    if dt != "fp32":
        x1, x2, w1, w2 = _round(x1, dt), _round(x2, dt), _round(w1, dt), _round(w2, dt)
    ref = F.relu(F.conv2d(x1.double(), w1.double(), b1.double()) + F.conv2d(x2.double()[:, :, ::s2, ::s2][:, :, :H, :W], w2.double(), b2.double()))
    layer = dual_source_pointwise("t", w1.numpy(), b1.numpy(), c1, w2.numpy(), b2.numpy(), c2, s2, e.dt, e.device)
    a1 = Act(_nhwc(x1, c1, e.tdt, e.device), N, H, W, c1)
    a2 = Act(_nhwc(x2, c2, e.tdt, e.device), N, H2, W2, c2)
    for force in (None, "2"):                  # the policy's choice and the 128x128 ring kernel
        if force:
            policy.set("conv_big", force)
        out = e.conv(layer, a1, relu=True, in2=a2, out_f32=True)
        torch.cuda.synchronize()
        got = out.t.cpu().permute(0, 3, 1, 2).double()
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1.0) * (c1 + c2) ** 0.5, (shape, force, err)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(1000, 1, 1, 12544, 1024, 1, 2), (5000, 1, 1, 12544, 1024, 1, 2), (1, 25, 42, 512, 512, 3, 3),
                                   (8, 25, 42, 512, 512, 3, 3), (3, 9, 7, 128, 128, 3, 2), (37, 1, 1, 192, 256, 1, 3)])
def test_conv_split_k(eng, dt, shape):
    """dp_conv_params.split_k (the box head's fc1, box_head.py:60-67; res5's 3x3, resnet.py:195-197): K in segments of fp32 partial
    sums + one reduction pass. Against torch in fp64, against the unsplit launch of the same layer, and - the point of fixing the
    segment count per layer - bit for bit between a batch and any of its rows run alone (other tile shape, same segments)."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw, set_split_k
    e = eng[dt]
    N, H, W, cin, cout, k, seg = shape
    g = torch.Generator().manual_seed(cin + cout + N)
    x = _round(torch.randn((N, cin, H, W), generator=g), dt)
    w = _round(torch.randn((cout, cin, k, k), generator=g) * (1.0 / (cin * k * k)) ** 0.5, dt)
    b = torch.randn((cout,), generator=g)
    layer = conv_from_oihw("t", w.numpy(), b.numpy(), cin, 1, k // 2, 1, e.dt, e.device)
    set_split_k(layer, seg)
    assert layer.split_k == seg
    a = Act(_nhwc(x, cin, e.tdt, e.device), N, H, W, cin)
    got = e.conv(layer, a, relu=True).t.float().cpu()
    e.split_k_on = False
    try:
        plain = e.conv(layer, a, relu=True).t.float().cpu()
    finally:
        e.split_k_on = True
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)).permute(0, 2, 3, 1)
    scale = max(ref.abs().max().item(), 1.0)
    tol = (2.0 ** -8 if dt == "bf16" else 2.0 ** -11) * scale + 2e-5 * scale * (cin * k * k) ** 0.5     # storage rounding + accumulation
    assert (got.double() - ref).abs().max().item() <= tol
    assert (got - plain).abs().max().item() <= (2.0 ** -7 if dt == "bf16" else 2.0 ** -10) * scale     # at most one step of the storage type apart
    for n0, n1 in ((0, 1), (N // 2, N // 2 + max(1, N // 8))):
        sub = e.conv(layer, Act(a.t[n0:n1].contiguous(), n1 - n0, H, W, cin), relu=True).t.float().cpu()
        assert torch.equal(sub, got[n0:n1]), (shape, n0, n1)


def test_conv_fpn_lateral_plus_nearest_upsample(eng):
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng["fp32"]
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 64, 12, 20), generator=g)
    top = torch.randn((2, 32, 6, 10), generator=g)
    w = torch.randn((32, 64, 1, 1), generator=g) * 0.1
    b = torch.randn((32,), generator=g)
    ref = F.conv2d(x, w, b) + F.interpolate(top, scale_factor=2.0, mode="nearest")
    layer = conv_from_oihw("lat", w.numpy(), b.numpy(), 64, 1, 0, 1, e.dt, e.device)
    out = e.conv(layer, Act(_nhwc(x, 64, e.tdt, e.device), 2, 12, 20, 64),
                 residual=Act(_nhwc(top, 32, e.tdt, e.device), 2, 6, 10, 32), rshift=1)
    got = out.t.cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5)


def test_linear_and_deconv_forms(eng):
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import deconv_parity_convs, linear_as_conv
    e = eng["fp32"]
    g = torch.Generator().manual_seed(4)
    x = torch.randn((300, 392), generator=g)
    w = torch.randn((64, 392), generator=g) * 0.05
    b = torch.randn((64,), generator=g)
    layer = linear_as_conv("fc", w.numpy(), b.numpy(), 392, e.dt, e.device)
    out = e.conv(layer, Act(x.to(e.device).view(300, 1, 1, 392), 300, 1, 1, 392), relu=True)
    assert torch.allclose(out.t.cpu().view(300, 64), F.relu(F.linear(x, w, b)), atol=1e-5, rtol=1e-5)
    # ConvTranspose2d(k4, s2, p1) as four 2x2 sub-pixel convolutions
    xi = torch.randn((3, 32, 7, 7), generator=g)
    w1 = torch.randn((32, 2, 4, 4), generator=g) * 0.1
    w2 = torch.randn((32, 25, 4, 4), generator=g) * 0.1
    b1, b2 = torch.randn((2,), generator=g), torch.randn((25,), generator=g)
    ref = torch.cat([F.conv_transpose2d(xi, w1, b1, stride=2, padding=1), F.conv_transpose2d(xi, w2, b2, stride=2, padding=1)], dim=1)
    convs = deconv_parity_convs("d", [w1.numpy(), w2.numpy()], [b1.numpy(), b2.numpy()], 32, e.dt, e.device)
    Ci = convs[(0, 0)].cout
    low = torch.zeros((3, 14, 14, Ci), dtype=torch.float32, device=e.device)
    xa = Act(_nhwc(xi, 32, e.tdt, e.device), 3, 7, 7, 32)
    for (a, bb), l in convs.items():
        e.conv(l, xa, out_f32=True, out=low, out_c_stride=Ci, out_geom=(14 * 14 * Ci, 2 * 14 * Ci, 2 * Ci, (a * 14 + bb) * Ci))
    got = low.cpu()[..., :27].permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("ring2", [False, True])       # the 128x128 ring tile / the 256x128 two-workgroup tile (policy line lowered)
@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_deconv_parity_classes_in_one_launch(eng, dt, ring2, policy):
    """The chart predictor's ConvTranspose2d layers (chart.py:45-60: 512 -> 2 + 25 + 25 + 25 channels, k4 s2 p1) on R ROI maps of
    28 x 28: the four sub-pixel convolutions as ONE grouped launch (dp_conv_params.n_groups) against the four separate launches - bit
    for bit, with a device-side live count, slots behind it untouched - and against torch's conv_transpose2d."""
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import deconv_parity_convs
    e = eng[dt]
    R, Cin, P, live = 11, 512, 28, 9
    g = torch.Generator().manual_seed(77)
    xi = torch.randn((R, Cin, P, P), generator=g)
    ws = [torch.randn((Cin, c, 4, 4), generator=g) * (1.0 / (4 * Cin)) ** 0.5 for c in (2, 25, 25, 25)]
    bs = [torch.randn((c,), generator=g) for c in (2, 25, 25, 25)]
    if dt != "fp32":
        xi, ws = _round(xi, dt), [_round(w, dt) for w in ws]
    convs = deconv_parity_convs("d", [w.numpy() for w in ws], [b.numpy() for b in bs], Cin, e.dt, e.device)
    items = list(convs.items())
    Ci, P2 = items[0][1].cout, 2 * P
    xa = Act(_nhwc(xi, Cin, e.tdt, e.device), R, P, P, Cin)
    n_dev = torch.tensor([live], dtype=torch.int32, device=e.device)
    policy.set("conv_ring2_m", 1 if ring2 else 0)
    geom = lambda base: (P2 * P2 * Ci, 2 * P2 * Ci, 2 * Ci, base)      # noqa: E731
    assert e.groups_fusable(items[0][1], xa, n_dev)
    want = torch.full((R, P2, P2, Ci), 7.0, dtype=torch.float32, device=e.device)
    for (a, b), l in items:
        e.conv(l, xa, out_f32=True, out=want, out_c_stride=Ci, out_geom=geom((a * P2 + b) * Ci), n_dev=n_dev)
    got = torch.full((R, P2, P2, Ci), 7.0, dtype=torch.float32, device=e.device)
    e.conv(items[0][1], xa, out_f32=True, out=got, out_c_stride=Ci, out_geom=geom(0), n_dev=n_dev,
           groups=[(l, (a * P2 + b) * Ci) for (a, b), l in items])
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert bool((got[live + 1:] == 7.0).all())     # (rows behind the count inside the last live tile are computed: dp_conv_params.n_dev)
    ref = torch.cat([F.conv_transpose2d(xi.double(), w.double(), b.double(), stride=2, padding=1) for w, b in zip(ws, bs)], dim=1)
    gd = got[:live, :, :, :77].cpu().permute(0, 3, 1, 2).double()
    assert bool(((gd - ref[:live]).abs() <= 1e-5 * ref[:live].abs() + 2e-4).all()), float((gd - ref[:live]).abs().max())
    # a layer the 128-cout ring kernels do not take (a single K plane on the generic kernel) refuses the grouped form
    small = deconv_parity_convs("s", [ws[0][:8].numpy()], [bs[0].numpy()], 8, e.dt, e.device)
    assert not e.groups_fusable(list(small.values())[0], Act(xa.t[..., :8].contiguous(), R, P, P, 8))


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_pool_subsample_upsample(eng, dt):
    e = eng[dt]
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 16, 13, 18), generator=g)
    if dt != "fp32":
        x = _round(x, dt)
    xa = _nhwc(x, 16, e.tdt, e.device)
    s = e._stream()
    out = torch.empty((2, 7, 9, 16), dtype=e.tdt, device=e.device)
    assert e.lib.dp_maxpool3x3s2_nhwc(xa.data_ptr(), out.data_ptr(), 2, 13, 18, 16, e.dt, s) == 0
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1))
    out = torch.empty((2, 7, 9, 16), dtype=e.tdt, device=e.device)
    assert e.lib.dp_subsample2_nhwc(xa.data_ptr(), out.data_ptr(), 2, 13, 18, 16, e.dt, s) == 0
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), F.max_pool2d(x, 1, 2, 0))
    up = torch.empty((2, 26, 36, 16), dtype=e.tdt, device=e.device)
    assert e.lib.dp_upsample_bilinear2x_nhwc(xa.data_ptr(), up.data_ptr(), 2, 13, 18, 16, 0, e.dt, s) == 0
    ref = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
    tol = 1e-6 if dt == "fp32" else 2e-2
    assert torch.allclose(up.float().cpu().permute(0, 3, 1, 2), ref, atol=tol)
    assert e.lib.dp_upsample_bilinear2x_nhwc(xa.data_ptr(), up.data_ptr(), 2, 13, 18, 16, 1, e.dt, s) == 0
    assert torch.allclose(up.float().cpu().permute(0, 3, 1, 2), 2 * ref, atol=2 * tol)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 5, 7, 16, 3), (1, 1, 1, 8, 1), (3, 1, 6, 24, 2), (1, 9, 1, 8, 3), (2, 12, 20, 256, 3)])
def test_merge_upsample2x(eng, dt, shape):
    """decoder level sum (roi_head.py:71-79): out = base + sum_k bilinear_x2(ups[k]), 2x2-blocked kernel vs torch"""
    e = eng[dt]
    N, H, W, Cc, n_ups = shape
    g = torch.Generator().manual_seed(N * 1000 + H * 100 + W)
    base = torch.randn((N, Cc, 2 * H, 2 * W), generator=g)
    ups = [torch.randn((N, Cc, H, W), generator=g) for _ in range(n_ups)]
    if dt != "fp32":
        base, ups = _round(base, dt), [_round(u, dt) for u in ups]
    ref = base.clone()
    for u in ups:
        ref = ref + F.interpolate(u, scale_factor=2.0, mode="bilinear", align_corners=False)
    bt = _nhwc(base, Cc, e.tdt, e.device)
    uts = [_nhwc(u, Cc, e.tdt, e.device) for u in ups]
    out = torch.empty_like(bt)
    arr = (C.c_void_p * n_ups)(*[u.data_ptr() for u in uts])
    assert e.lib.dp_merge_upsample2x_nhwc(bt.data_ptr(), arr, n_ups, out.data_ptr(), N, H, W, Cc, e.dt, e._stream()) == 0
    got = out.float().cpu().permute(0, 3, 1, 2)
    if dt == "fp32":
        assert torch.allclose(got, ref, atol=2e-6, rtol=0)
    else:
        assert torch.allclose(got, _round(ref, dt), atol=4e-2, rtol=1e-2)
    # in place (out aliases base), as the engine calls it
    assert e.lib.dp_merge_upsample2x_nhwc(bt.data_ptr(), arr, n_ups, bt.data_ptr(), N, H, W, Cc, e.dt, e._stream()) == 0
    assert torch.equal(bt, out)


def _random_boxes(rng, n, size=400.0, zero_frac=0.05):
    xy = rng.uniform(0, size, (n, 2)).astype(np.float32)
    wh = rng.uniform(2, size / 3, (n, 2)).astype(np.float32)
    b = np.concatenate([xy, xy + wh], 1)
    z = rng.random(n) < zero_frac
    b[z, 2] = b[z, 0]
    return b


@pytest.mark.parametrize("reference", ["cpu", "cuda"])
@pytest.mark.parametrize("n_slots,groups", [(1000, 1), (900, 5), (5000, 5), (37, 2), (1, 1), (63, 1), (64, 2), (65, 3), (129, 1), (4097, 5)])
def test_batched_nms_matches_oracle(eng, n_slots, groups, reference, monkeypatch):
    """Both strategy switches of torchvision's batched_nms: 4000 box elements (the reference on the CPU) and 20000 (its CUDA mode,
    run.py:22-29); n_slots = 4097 / 5000 with ~90 % valid boxes sit between the two, so the two modes take different strategies there."""
    from oracle import ops_ref
    from densepose_torchscript_amd.engine import NMS_TRICK_MAX_NUMEL
    e = eng["fp32"]
    monkeypatch.setattr(e, "nms_reference", reference)
    rng = np.random.default_rng(n_slots)
    n_img = 2
    boxes = np.stack([_random_boxes(rng, n_slots) for _ in range(n_img)])
    scores = np.stack([rng.permutation(n_slots).astype(np.float32) / n_slots for _ in range(n_img)])
    group = rng.integers(0, groups, (n_img, n_slots)).astype(np.int32)
    valid = (rng.random((n_img, n_slots)) > 0.1).astype(np.int32)
    dev = e.device
    ob, os_, oi, oc = e.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), torch.from_numpy(group).to(dev),
                            torch.from_numpy(valid).to(dev), n_img, n_slots, 0.7, 300)
    torch.cuda.synchronize()
    for i in range(n_img):
        v = valid[i].astype(bool)
        idx = np.nonzero(v)[0]
        keep = ops_ref.batched_nms(torch.from_numpy(boxes[i][v]), torch.from_numpy(scores[i][v]), torch.from_numpy(group[i][v]).long(), 0.7,
                                   trick_max_numel=NMS_TRICK_MAX_NUMEL[reference])
        keep = idx[keep.numpy()][:300]
        cnt = int(oc[i])
        assert cnt == len(keep)
        assert np.array_equal(oi[i, :cnt].cpu().numpy(), keep)
        assert np.array_equal(ob[i, :cnt].cpu().numpy(), boxes[i][keep])
        assert np.array_equal(os_[i, :cnt].cpu().numpy(), scores[i][keep])


@pytest.mark.parametrize("reference", ["cuda", "cpu"])
@pytest.mark.parametrize("layout", ["rpn", "rpn_one_run_unsorted", "nine_runs"])
def test_batched_nms_presorted_runs(eng, layout, reference, monkeypatch):
    """The RPN's candidate layout: one run of slots per pyramid level, each run already in score order (ties between and inside
    runs, invalid candidates anywhere, an all-padding tail) - nms_sort_kernel merges the runs instead of sorting. One run out of
    order, or more runs than the merge handles, must fall back to the sort and give the same answer. Tied scores only with the
    coordinate-offset strategy ("cuda": 4 * 4119 <= 20000), whose order is score, then index; torchvision's per-group strategy
    ("cpu" at this size) orders the survivors with an unstable sort, so that case gets distinct scores."""
    from oracle import ops_ref
    from densepose_torchscript_amd.engine import NMS_TRICK_MAX_NUMEL
    e = eng["fp32"]
    monkeypatch.setattr(e, "nms_reference", reference)
    rng = np.random.default_rng(5)
    n_img = 2
    run_len = [1000, 1000, 1000, 700, 300, 119] if layout != "nine_runs" else [500] * 9
    n_slots = sum(run_len)
    boxes = np.stack([_random_boxes(rng, n_slots) for _ in range(n_img)])
    scores = np.zeros((n_img, n_slots), np.float32)
    group = np.zeros((n_img, n_slots), np.int32)
    valid = np.ones((n_img, n_slots), np.int32)
    for i in range(n_img):
        o = 0
        distinct = rng.permutation(n_slots).astype(np.float32) / 64 - 30
        for g, ln in enumerate(run_len):
            sc = np.sort(rng.integers(-40, 40, ln).astype(np.float32) / 8)[::-1]     # few distinct values: many ties
            if reference == "cpu":
                sc = np.sort(distinct[o:o + ln])[::-1]
            scores[i, o:o + ln], group[i, o:o + ln] = sc, g
            valid[i, o:o + ln] = rng.random(ln) > 0.05
            if g == 3:                      # a level with fewer anchors than slots: padding (score 0, invalid) behind the real ones
                scores[i, o + ln - 80:o + ln], valid[i, o + ln - 80:o + ln] = (0.0 if reference == "cuda" else -1000.0 - i), 0
            if g == 4 and i == 1:
                valid[i, o:o + ln] = 0      # a run without a single valid candidate
            o += ln
        if layout == "rpn_one_run_unsorted":
            scores[i, 1200], scores[i, 1700] = scores[i, 1700], scores[i, 1200] + 1.0
    dev = e.device
    ob, os_, oi, oc = e.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), torch.from_numpy(group).to(dev),
                            torch.from_numpy(valid).to(dev), n_img, n_slots, 0.7, 1000)
    torch.cuda.synchronize()
    for i in range(n_img):
        v = valid[i].astype(bool)
        idx = np.nonzero(v)[0]
        keep = ops_ref.batched_nms(torch.from_numpy(boxes[i][v]), torch.from_numpy(scores[i][v]), torch.from_numpy(group[i][v]).long(), 0.7,
                                   trick_max_numel=NMS_TRICK_MAX_NUMEL[e.nms_reference])
        keep = idx[keep.numpy()][:1000]
        cnt = int(oc[i])
        assert cnt == len(keep)
        assert np.array_equal(oi[i, :cnt].cpu().numpy(), keep)
        assert np.array_equal(os_[i, :cnt].cpu().numpy(), scores[i][keep])


@pytest.mark.parametrize("kind", ["all_invalid", "clusters", "identical"])
def test_batched_nms_degenerate_inputs(eng, kind):
    """No valid candidate (count 0), dense clusters where most boxes are suppressed by an earlier one of their cluster, and
    identical boxes with distinct scores (exactly one survives per group)."""
    from oracle import ops_ref
    e = eng["fp32"]
    rng = np.random.default_rng(99)
    n_img, n = 2, 700
    if kind == "clusters":
        centres = rng.random((12, 2)) * 400
        c = centres[rng.integers(0, 12, n)]
        wh = 40 + rng.random((n, 2)) * 6
        xy = c + rng.normal(0, 3, (n, 2))
        b1 = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        boxes = np.stack([b1, b1[::-1].copy()])
    elif kind == "identical":
        boxes = np.tile(np.array([[10, 20, 110, 220]], np.float32), (n_img, n, 1))
    else:
        boxes = np.stack([_random_boxes(rng, n) for _ in range(n_img)]).astype(np.float32)
    scores = np.stack([rng.permutation(n).astype(np.float32) / n for _ in range(n_img)])
    group = rng.integers(0, 3, (n_img, n)).astype(np.int32)
    valid = np.zeros((n_img, n), np.int32) if kind == "all_invalid" else np.ones((n_img, n), np.int32)
    dev = e.device
    ob, os_, oi, oc = e.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), torch.from_numpy(group).to(dev),
                            torch.from_numpy(valid).to(dev), n_img, n, 0.5, 100)
    torch.cuda.synchronize()
    for i in range(n_img):
        cnt = int(oc[i])
        if kind == "all_invalid":
            assert cnt == 0
            continue
        keep = ops_ref.batched_nms(torch.from_numpy(boxes[i]), torch.from_numpy(scores[i]), torch.from_numpy(group[i]).long(), 0.5).numpy()[:100]
        assert cnt == len(keep) and np.array_equal(oi[i, :cnt].cpu().numpy(), keep)
        if kind == "identical":
            assert cnt == len(np.unique(group[i]))


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("multi", [True, False])
@pytest.mark.parametrize("sampling", [2, 3])   # 2: the kernel with the sample geometry tabulated per ROI; 3: the per-sample kernel
def test_roi_align_matches_oracle(eng, dt, multi, sampling):
    from densepose_torchscript_amd.engine import Act
    from oracle import ops_ref
    from oracle.ref_cpu import OracleModel
    e = eng[dt]
    rng = np.random.default_rng(7)
    g = torch.Generator().manual_seed(7)
    n_img, Cc = 2, 32
    shapes = [(40, 56), (20, 28), (10, 14), (5, 7)] if multi else [(40, 56)]
    maps = [torch.randn((n_img, Cc, h, w), generator=g) for h, w in shapes]
    if dt != "fp32":
        maps = [_round(m, dt) for m in maps]
    scales = [1.0 / 4, 1.0 / 8, 1.0 / 16, 1.0 / 32][: len(shapes)]
    max_rois, P = 50, 7
    boxes = np.stack([_random_boxes(rng, max_rois, 220.0, 0.1) for _ in range(n_img)])
    boxes[0, 0] = [-30, -20, 500, 400]  # far outside + huge
    boxes[0, 1] = [10, 10, 10, 10]      # zero area
    counts = np.array([50, 33], dtype=np.int32)
    acts = [Act(_nhwc(m, Cc, e.tdt, e.device), n_img, m.shape[2], m.shape[3], Cc) for m in maps]
    out = torch.full((n_img * max_rois, P, P, Cc), 7.0, dtype=e.tdt, device=e.device)   # garbage: the kernel owns every row
    e.roi_align(acts, scales, torch.from_numpy(boxes).to(e.device), torch.from_numpy(counts).to(e.device), n_img, max_rois, P, sampling, out)
    torch.cuda.synchronize()
    got = out.float().cpu().view(n_img, max_rois, P, P, Cc).permute(0, 1, 4, 2, 3)
    assert float(got[1, counts[1]:].abs().max()) == 0.0   # padded slots of the fixed-size layout are written as zeros
    for i in range(n_img):
        b = torch.from_numpy(boxes[i, : counts[i]])
        rois = torch.cat([torch.zeros((len(b), 1)), b], 1)
        if multi:
            lv = OracleModel.assign_levels(b, 2, 5)
            ref = torch.zeros((len(b), Cc, P, P))
            for l in range(4):
                idx = torch.nonzero(lv == l)[:, 0]
                ref[idx] = ops_ref.roi_align(maps[l][i:i + 1], rois[idx], P, scales[l], sampling, False)
        else:
            ref = ops_ref.roi_align(maps[0][i:i + 1], rois, P, scales[0], sampling, False)
        tol = 2e-6 if dt == "fp32" else 3e-2
        assert torch.allclose(got[i, : counts[i]], ref, atol=tol), (got[i, : counts[i]] - ref).abs().max()


@pytest.mark.parametrize("Hi,Wi", [(48, 80), (100, 150)])   # 11,520 anchors: single-stage select ; 45,000: chunked two-stage select
def test_rpn_topk_decode_matches_oracle(eng, Hi, Wi):
    from densepose_torchscript_amd import lib as L
    from oracle.ref_cpu import OracleModel
    e = eng["fp32"]
    g = torch.Generator().manual_seed(9)
    n_img, A = 2, 3
    head = torch.randn((n_img, Hi, Wi, 16), generator=g)
    head[..., 3:15] *= 0.3
    head[0, 0, 0, 5] = float("nan")     # a NaN delta -> that anchor must be dropped if selected
    head[0, 0, 0, 0] = 50.0             # ... and it IS selected
    head[1, 3, 4, 1] = 40.0
    head[1, 3, 4, 9] = 20.0             # dw above the clamp
    head[1, :, :5, 2] = 2.2 if Hi == 48 else 2.75   # exact ties (240 / 500 anchors) that straddle the top-k boundary
    kmax, stride = 200, 8
    cell = [[-22.6, -11.3, 22.6, 11.3], [-16.0, -16.0, 16.0, 16.0], [-11.3, -22.6, 11.3, 22.6]]
    dev = e.device
    slots = 2 * kmax
    cb = torch.zeros((n_img, slots, 4), device=dev)
    cs = torch.zeros((n_img, slots), device=dev)
    cl = torch.zeros((n_img, slots), dtype=torch.int32, device=dev)
    cv = torch.zeros((n_img, slots), dtype=torch.int32, device=dev)
    ws = torch.empty((e.lib.dp_rpn_topk_workspace_bytes(n_img, Hi, Wi, A),), dtype=torch.uint8, device=dev)
    hd = head.to(dev)
    p = L.RpnLevelParams()
    p.head, p.n_img, p.Hi, p.Wi, p.A, p.head_c, p.stride_px = hd.data_ptr(), n_img, Hi, Wi, A, 16, stride
    for a in range(3):
        for c in range(4):
            p.cell_anchors[a][c] = cell[a][c]
    p.level, p.kmax, p.slot_off, p.slots_per_img = 1, kmax, kmax, slots
    p.clip_x, p.clip_y = 300.0, 500.0
    p.cand_boxes, p.cand_scores, p.cand_level, p.cand_valid = cb.data_ptr(), cs.data_ptr(), cl.data_ptr(), cv.data_ptr()
    p.workspace = ws.data_ptr()
    L.check(e.lib.dp_rpn_topk_decode(C.byref(p), e._stream()))
    torch.cuda.synchronize()
    ca = torch.tensor(cell)
    sx = torch.arange(0, Wi * stride, stride, dtype=torch.float32)
    sy = torch.arange(0, Hi * stride, stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), 1)
    anchors = (shifts.view(-1, 1, 4) + ca.view(1, -1, 4)).reshape(-1, 4)
    for i in range(n_img):
        logits = head[i, :, :, :3].reshape(-1)
        deltas = head[i, :, :, 3:15].reshape(-1, 4)
        sc, idx = logits.topk(kmax)
        got_s = cs[i, kmax:].cpu()
        assert torch.equal(got_s, sc)   # same multiset & order of scores (ties have equal values)
        # compare boxes through the index our kernel effectively chose: recompute with OUR order for tie groups
        props = OracleModel.apply_deltas(deltas, anchors, (1.0, 1.0, 1.0, 1.0))
        ours_boxes = cb[i, kmax:].cpu()
        ours_valid = cv[i, kmax:].cpu().bool()
        # for non-tied scores the index is unique -> direct comparison
        uniq = torch.tensor([(logits == s).sum().item() == 1 for s in sc])
        ref_b = props[idx]
        fin = torch.isfinite(ref_b).all(1)
        ref_c = OracleModel.clip_boxes(ref_b, (500.0, 300.0))  # (h=clip_y, w=clip_x) as clip_boxes reads it
        ref_valid = fin & OracleModel.nonempty(ref_c)
        assert torch.equal(ours_valid[uniq], ref_valid[uniq])
        m = uniq & ref_valid
        assert torch.allclose(ours_boxes[m], ref_c[m], atol=1e-3, rtol=1e-5)
        assert int(cl[i, kmax:].min()) == 1 and int(cl[i, kmax:].max()) == 1
    # tie handling: exactly the needed number of tied entries is taken
    tv = 2.2 if Hi == 48 else 2.75
    lg = head[1, :, :, :3].reshape(-1)
    n_gt = int((lg > tv).sum())
    assert n_gt < kmax < n_gt + int((lg == tv).sum())
    assert int((cs[1, kmax:].cpu() == tv).sum()) == kmax - n_gt
    # the tied survivors are the ones with the LOWEST anchor indices (x < 5 columns scanned row-major): rows y = 0.. first
    tied = (cs[1, kmax:].cpu() == tv)
    got_boxes = cb[1, kmax:].cpu()[tied]
    tie_idx = torch.nonzero(lg == tv)[:, 0][: kmax - n_gt]
    props1 = OracleModel.apply_deltas(head[1, :, :, 3:15].reshape(-1, 4), anchors, (1.0, 1.0, 1.0, 1.0))
    exp = OracleModel.clip_boxes(props1[tie_idx], (500.0, 300.0))
    assert torch.allclose(got_boxes, exp, atol=1e-3, rtol=1e-5)


def test_box_decode_and_groupnorm_gap(eng):
    from densepose_torchscript_amd import lib as L
    from oracle.ref_cpu import OracleModel
    e = eng["fp32"]
    dev = e.device
    g = torch.Generator().manual_seed(11)
    n_img, R = 2, 40
    logits = torch.randn((n_img * R, 8), generator=g)
    props = torch.from_numpy(np.stack([_random_boxes(np.random.default_rng(i), R) for i in range(n_img)]))
    counts = torch.tensor([40, 25], dtype=torch.int32)
    cb = torch.zeros((n_img, R, 4), device=dev)
    cs = torch.zeros((n_img, R), device=dev)
    cg = torch.zeros((n_img, R), dtype=torch.int32, device=dev)
    cv = torch.zeros((n_img, R), dtype=torch.int32, device=dev)
    p = L.BoxDecodeParams()
    ld, pd, cd = logits.to(dev), props.to(dev), counts.to(dev)
    p.logits, p.ld, p.prop_boxes, p.prop_counts, p.n_img, p.max_rois = ld.data_ptr(), 8, pd.data_ptr(), cd.data_ptr(), n_img, R
    p.wx, p.wy, p.ww, p.wh, p.score_thresh = 10.0, 10.0, 5.0, 5.0, 0.3
    p.cand_boxes, p.cand_scores, p.cand_group, p.cand_valid = cb.data_ptr(), cs.data_ptr(), cg.data_ptr(), cv.data_ptr()
    L.check(e.lib.dp_box_decode_score(C.byref(p), e._stream()))
    torch.cuda.synchronize()
    ref_b = OracleModel.apply_deltas(logits[:, 2:6], props.view(-1, 4), (10.0, 10.0, 5.0, 5.0)).view(n_img, R, 4)
    ref_p = F.softmax(logits[:, :2], dim=-1)[:, 0].view(n_img, R)
    for i in range(n_img):
        c = int(counts[i])
        assert torch.allclose(cb[i, :c].cpu(), ref_b[i, :c], atol=1e-3, rtol=1e-5)
        assert torch.allclose(cs[i, :c].cpu(), ref_p[i, :c], atol=1e-6)
        near = (ref_p[i, :c] - 0.3).abs() < 1e-5
        assert torch.equal(cv[i, :c].cpu().bool()[~near], (ref_p[i, :c] > 0.3)[~near])
        assert int(cv[i, c:].sum()) == 0
    # GroupNorm(32) + ReLU on a channel slice, GAP, broadcast
    Rr, HW, Cc = 3, 49, 64
    x = torch.randn((Rr, Cc, 7, 7), generator=g) * 2 + 0.5
    gamma, beta = torch.randn((Cc,), generator=g), torch.randn((Cc,), generator=g)
    buf = torch.zeros((Rr, HW, 2 * Cc), device=dev)
    buf[..., Cc:] = x.permute(0, 2, 3, 1).reshape(Rr, HW, Cc).to(dev)
    q = L.GroupNormParams()
    gd, bd = gamma.to(dev), beta.to(dev)
    q.x, q.R, q.HW, q.C, q.c_stride, q.c_off, q.groups = buf.data_ptr(), Rr, HW, Cc, 2 * Cc, Cc, 32
    q.gamma, q.beta, q.eps, q.relu, q.dtype = gd.data_ptr(), bd.data_ptr(), 1e-5, 1, L.DP_F32
    L.check(e.lib.dp_groupnorm_relu_nhwc(C.byref(q), e._stream()))
    ref = F.relu(F.group_norm(x, 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1).reshape(Rr, HW, Cc)
    assert torch.allclose(buf[..., Cc:].cpu(), ref, atol=2e-5)
    assert float(buf[..., :Cc].abs().max()) == 0.0
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    pooled = torch.empty((Rr, Cc), device=dev)
    assert e.lib.dp_global_avgpool_nhwc(xin.data_ptr(), pooled.data_ptr(), Rr, HW, Cc, L.DP_F32, None, e._stream()) == 0
    assert torch.allclose(pooled.cpu(), x.mean(dim=(2, 3)), atol=1e-5)
    bc = torch.zeros((Rr, HW, 2 * Cc), device=dev)
    assert e.lib.dp_broadcast_hw_nhwc(pooled.data_ptr(), bc.data_ptr(), Rr, HW, Cc, 2 * Cc, Cc, L.DP_F32, None, e._stream()) == 0
    assert torch.equal(bc[..., Cc:].cpu(), pooled.cpu()[:, None, :].expand(Rr, HW, Cc))


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(3, 28, 256, 1280, 512), (2, 28, 512, 512, 0), (5, 7, 64, 128, 64), (2, 1, 256, 256, 0)])
def test_groupnorm_gap_broadcast_all_dtypes(eng, dt, shape):
    """DeepLab head pieces (deeplab.py:45,90,97,109) in every storage type at the real geometry: 28x28 ROI maps, GroupNorm(32)
    over 256 channels (groups of 8, written into a slice of the 1280-channel ASPP concat) and over 512 channels (groups of
    16), the 1x1 pooled branch (HW = 1). Statistics are fp32 whatever the storage type; the result is rounded once."""
    from densepose_torchscript_amd import lib as L
    e = eng[dt]
    Rr, P, Cc, cstride, coff = shape
    HW = P * P
    g = torch.Generator().manual_seed(Rr * 100 + P + Cc)
    x = torch.randn((Rr, Cc, P, P), generator=g) * 1.5 + 0.3
    gamma, beta = torch.randn((Cc,), generator=g), torch.randn((Cc,), generator=g)
    if dt != "fp32":
        x = _round(x, dt)
    ulp = {"fp32": 2.0 ** -22, "bf16": 2.0 ** -8, "fp16": 2.0 ** -11}[dt]
    buf = torch.zeros((Rr, HW, cstride), dtype=e.tdt, device=e.device)
    buf[..., coff:coff + Cc] = x.permute(0, 2, 3, 1).reshape(Rr, HW, Cc).to(e.tdt).to(e.device)
    q = L.GroupNormParams()
    gd, bd = gamma.to(e.device), beta.to(e.device)
    q.x, q.R, q.HW, q.C, q.c_stride, q.c_off, q.groups = buf.data_ptr(), Rr, HW, Cc, cstride, coff, 32
    q.gamma, q.beta, q.eps, q.relu, q.dtype = gd.data_ptr(), bd.data_ptr(), 1e-5, 1, e.dt
    L.check(e.lib.dp_groupnorm_relu_nhwc(C.byref(q), e._stream()))
    ref = F.relu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5)).permute(0, 2, 3, 1).reshape(Rr, HW, Cc)
    got = buf[..., coff:coff + Cc].double().cpu()
    assert bool(((got - ref).abs() <= ulp * ref.abs() + 3e-5).all()), float((got - ref).abs().max())
    other = torch.cat([buf[..., :coff], buf[..., coff + Cc:]], dim=-1)
    assert other.numel() == 0 or float(other.float().abs().max()) == 0.0       # channels outside the slice are untouched
    xin = x.permute(0, 2, 3, 1).contiguous().to(e.tdt).to(e.device)
    pooled = torch.empty((Rr, Cc), dtype=e.tdt, device=e.device)
    assert e.lib.dp_global_avgpool_nhwc(xin.data_ptr(), pooled.data_ptr(), Rr, HW, Cc, e.dt, None, e._stream()) == 0
    mref = x.double().mean(dim=(2, 3))
    assert bool(((pooled.double().cpu() - mref).abs() <= ulp * mref.abs() + 1e-5).all())
    bc = torch.zeros((Rr, HW, cstride), dtype=e.tdt, device=e.device)
    assert e.lib.dp_broadcast_hw_nhwc(pooled.data_ptr(), bc.data_ptr(), Rr, HW, Cc, cstride, coff, e.dt, None, e._stream()) == 0
    assert torch.equal(bc[..., coff:coff + Cc].cpu(), pooled.cpu()[:, None, :].expand(Rr, HW, Cc))


def test_resize_and_iuv_extract(eng):
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.resize import resize_u8_device
    from oracle.ref_cpu import extract_iuv
    e = eng["fp32"]
    dev = e.device
    rng = np.random.default_rng(13)
    for (H, W, mn, mx) in [(480, 640, 800, 1333), (1080, 1920, 800, 1333), (96, 160, 128, 213), (333, 517, 800, 1333)]:
        img = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
        k = min(mn / min(H, W), mx / max(H, W))
        ref = F.interpolate(img.permute(2, 0, 1)[None], scale_factor=k, mode="bilinear", align_corners=False)[0]
        got = resize_u8_device(e, img.to(dev), k, src_hwc=True)
        assert torch.equal(got.cpu(), ref), (H, W)
        got = resize_u8_device(e, img.permute(2, 0, 1).contiguous().to(dev), k, src_hwc=False)
        assert torch.equal(got.cpu(), ref), (H, W)
    # frames of one geometry in one launch per pass (more than one chunk of 64 frame pointers)
    from densepose_torchscript_amd.resize import resize_u8_device_batch
    frames = [torch.from_numpy(rng.integers(0, 256, (60, 90, 3), dtype=np.uint8)) for _ in range(70)]
    k = min(100 / 60, 140 / 90)
    got = resize_u8_device_batch(e, [f.to(dev) for f in frames], k, src_hwc=True).cpu()
    for f, gt in zip(frames, got):
        assert torch.equal(gt, F.interpolate(f.permute(2, 0, 1)[None], scale_factor=k, mode="bilinear", align_corners=False)[0])
    # IUV extraction (visualizer.py:10-30)
    g = torch.Generator().manual_seed(14)
    R, S = 3, 28
    out = {"pred_boxes": torch.tensor([[3.2, 4.9, 60.7, 90.1], [10.0, 10.0, 10.5, 80.0], [0.0, 0.0, 120.0, 33.3]]),
           "pred_densepose_coarse_segm": torch.randn((R, 2, S, S), generator=g),
           "pred_densepose_fine_segm": torch.randn((R, 25, S, S), generator=g),
           "pred_densepose_u": torch.rand((R, 25, S, S), generator=g),
           "pred_densepose_v": torch.rand((R, 25, S, S), generator=g)}
    ref = extract_iuv(out)
    xywh = out["pred_boxes"].clone()
    xywh[:, 2:] -= xywh[:, :2]
    xywh = xywh.long()
    xywh[:, 2:] = xywh[:, 2:].clamp(min=1)
    hw = (xywh[:, 2] * xywh[:, 3]).numpy()
    offs = np.zeros((R,), dtype=np.int64)
    offs[1:] = np.cumsum(hw)[:-1]
    labels = torch.zeros((int(hw.sum()),), dtype=torch.uint8, device=dev)
    uv = torch.zeros((2 * int(hw.sum()),), dtype=torch.float32, device=dev)
    t = {k: v.to(dev) for k, v in out.items()}
    bx, od = xywh.int().to(dev), torch.from_numpy(offs).to(dev)
    p = L.IuvExtractParams()
    p.coarse, p.fine, p.u, p.v = (t["pred_densepose_coarse_segm"].data_ptr(), t["pred_densepose_fine_segm"].data_ptr(),
                                  t["pred_densepose_u"].data_ptr(), t["pred_densepose_v"].data_ptr())
    p.R, p.S, p.n_coarse, p.n_fine = R, S, 2, 25
    p.box_xywh, p.out_offset, p.labels, p.uv, p.max_hw = bx.data_ptr(), od.data_ptr(), labels.data_ptr(), uv.data_ptr(), int(hw.max())
    L.check(e.lib.dp_iuv_extract(C.byref(p), e._stream()))
    torch.cuda.synchronize()
    for r in range(R):
        w, h = int(xywh[r, 2]), int(xywh[r, 3])
        lab = labels[offs[r]: offs[r] + h * w].cpu().view(h, w)
        u_ = uv[2 * offs[r]: 2 * offs[r] + 2 * h * w].cpu().view(2, h, w)
        # part index: bit-exact. The kernel restates ATen's CPU bilinear arithmetic operation by operation (dp_extra.hip,
        # src_index / sample); torch's CPU kernel itself switches its FMA contraction pattern with the tensor shape (output
        # narrower than 63 columns, channel count: measured in this container), so the resampled VALUES can sit one ulp
        # apart on such shapes - the argmax over them does not move
        assert torch.equal(lab.long(), ref[r][0]), (r, int((lab.long() != ref[r][0]).sum()))
        assert torch.allclose(u_, ref[r][1], atol=1e-6, rtol=0), r
        if w >= 63:
            assert torch.equal(u_, ref[r][1]), r

@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("hwc", [True, False])
def test_preprocess_frames_equals_preprocess_of_the_stacked_batch(eng, dt, hwc):
    """dp_preprocess_u8_frames (ABI 5): rcnn.py:156-181 on n separate frames of the test size == dp_preprocess_u8 on the stacked
    batch, bit for bit (paired layout, interleaved and planar frames, ragged padding)."""
    e = eng[dt]
    g = torch.Generator().manual_seed(7)
    n, h, w = 5, 37, 51
    shape = (h, w, 3) if hwc else (3, h, w)
    frames = [torch.randint(0, 256, shape, generator=g, dtype=torch.uint8).to(e.device) for _ in range(n)]
    Hp, Wp = 64, 64
    want = e.preprocess(torch.stack(frames), Hp, Wp, hwc=hwc)
    got = torch.full_like(want.t, 3.0)
    e.preprocess_frames(frames, hwc, got)
    torch.cuda.synchronize()
    assert torch.equal(got, want.t)

@pytest.mark.parametrize("geom", [(5, 56, 56, 80, 2, 25), (3, 28, 28, 96, 15, 25), (2, 9, 13, 80, 2, 25), (4, 7, 10, 80, 2, 25)])
def test_iuv_upsample_split_forms_and_torch(eng, geom, policy):
    """dp_iuv_upsample_split (chart_predictor.py:45-70: bilinear x2 of the four predictor outputs, NHWC -> four NCHW tensors): the
    four-outputs-per-thread form with 16-byte stores (widths that are a multiple of 4) == the one-output form bit for bit, both ==
    F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) on the channel slices; a device-side live count leaves the
    slots behind it untouched."""
    import ctypes as C
    from densepose_torchscript_amd import lib as L
    e = eng["fp32"]
    R, Hs, Ws, Ci, nc, nf = geom
    g = torch.Generator().manual_seed(R * 100 + Ws)
    low = torch.randn((R, Hs, Ws, Ci), generator=g).to(e.device)
    live = torch.tensor([R - 1], dtype=torch.int32, device=e.device)
    outs = {}
    for quad in ("1", "0"):
        policy.set("iuv_quad", quad)
        S0, S1 = 2 * Hs, 2 * Ws
        t = [torch.full((R, n, S0, S1), 7.0, device=e.device) for n in (nc, nf, nf, nf)]
        p = L.IuvParams()
        p.in_, p.R, p.Hs, p.Ws, p.in_c, p.n_coarse, p.n_fine = low.data_ptr(), R, Hs, Ws, Ci, nc, nf
        p.coarse, p.fine, p.u, p.v = [x.data_ptr() for x in t]
        p.r_dev = live.data_ptr()
        L.check(e.lib.dp_iuv_upsample_split(C.byref(p), e._stream()), "dp_iuv_upsample_split")
        torch.cuda.synchronize()
        outs[quad] = t
    ref = F.interpolate(low.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    c0 = 0
    for k, n in enumerate((nc, nf, nf, nf)):
        a, b = outs["1"][k], outs["0"][k]
        assert torch.equal(a, b), k
        assert bool((a[R - 1] == 7.0).all())                      # the slot behind the live count
        assert torch.allclose(a[:R - 1], ref[:R - 1, c0:c0 + n], atol=1e-5, rtol=1e-5), k
        c0 += n

@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("C_", [256, 128])
def test_conv3x3_wsr_small_maps_equal_the_ring_kernel(eng, dt, C_, policy):
    """Round 4 lowered the weight-stationary 3x3 kernel's size line to 256 output pixels per launch (a single frame's p5 / p6 levels,
    fpn.py:134-135 / rpn.py:168 on 25 x 42 and 13 x 21 maps): on small, narrow and ragged maps it must equal the LDS-ring kernel bit for
    bit (the same K order) - which is why that line may depend on the batch."""
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.engine import Act
    from densepose_torchscript_amd.pack import conv_from_oihw
    e = eng[dt]
    g = torch.Generator().manual_seed(C_)
    w = _round(torch.randn((C_, C_, 3, 3), generator=g) * (1.0 / (9 * C_)) ** 0.5, dt)
    b = torch.randn((C_,), generator=g) * 0.1
    layer = conv_from_oihw("l", w.numpy(), b.numpy(), C_, 1, 1, 1, e.dt, e.device)
    policy.set("conv_wsq", "0")      # (kernel class 10 takes the 256-channel layers by default; this test is about class 6)
    for (N, H, W) in [(3, 8, 13), (1, 13, 21), (1, 25, 42), (2, 7, 9), (1, 6, 16), (2, 9, 17), (1, 12, 33)]:
        x = Act(_nhwc(_round(torch.randn((N, C_, H, W), generator=g), dt), C_, e.tdt, e.device), N, H, W, C_)
        p = L.ConvParams()
        p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = N, H, W, C_, H, W, C_, layer.cout_w, 9 * C_
        p.stride, p.ntaps, p.dtype, p.hi_off, p.wi_off, p.relu = 1, 9, e.dt, -1, -1, 1
        p.osN, p.osH, p.osW = H * W * C_, W * C_, C_
        p.out = 4096
        policy.default("conv_ws")
        assert (e.lib.dp_conv2d_kernel_class(C.byref(p)) == 6) == (N * H * W >= 256 and H >= (6 if C_ == 256 else 8)), (N, H, W)
        policy.set("ws_min_m", "1")
        got = e.conv(layer, x, relu=True).t.clone()
        policy.default("ws_min_m")
        policy.set("conv_ws", "0")
        assert e.lib.dp_conv2d_kernel_class(C.byref(p)) != 6
        want = e.conv(layer, x, relu=True).t
        torch.cuda.synchronize()
        assert torch.equal(got, want), (N, H, W)
