"""End-to-end parity of the HIP path (through the C ABI) against the reference's golden vectors and the CPU oracle.

Tolerances (BASELINE.json north_star): IUV maps fp32 atol 1e-3, part-index argmax bit-exact, in fp32 parity mode.
"""
import numpy as np
import pytest
import torch

from conftest import golden_case_inputs, load_golden

pytestmark = pytest.mark.gpu

IUV_KEYS = ("pred_densepose_coarse_segm", "pred_densepose_fine_segm", "pred_densepose_u", "pred_densepose_v")
IUV_ATOL = 1e-3


def _nchw(act):
    return act.t.float().cpu().permute(0, 3, 1, 2)


def _run(name, dtype="fp32", keep=False, resize="host"):
    from densepose_torchscript_amd.predictor import DensePosePredictor
    meta, z = load_golden(name)
    cfg, state, img = golden_case_inputs(meta)
    pred = DensePosePredictor(cfg, state, dtype=dtype, resize=resize, check_keep=True)
    pred.engine.keep_intermediates = keep
    out = pred(torch.from_numpy(img))
    torch.cuda.synchronize()
    return meta, z, cfg, pred, {k: v.cpu() for k, v in out.items()}


def _assert_rows_match(boxes, scores, ref_boxes, ref_scores, score_tol=1e-4, box_tol=2e-3):
    """Score-sorted lists agree up to permutations inside groups of scores closer than the fp32 noise of ~50 layers
    (the reference's own order inside such a group is decided by 1e-6 differences: SURVEY §7 hard part 1, Q8)."""
    np.testing.assert_allclose(scores, ref_scores, atol=score_tol)
    used = np.zeros(len(boxes), dtype=bool)
    for i in range(len(ref_boxes)):
        cand = np.nonzero((np.abs(scores - ref_scores[i]) <= score_tol) & ~used)[0]
        d = np.abs(boxes[cand] - ref_boxes[i]).max(axis=1)
        assert len(cand) and d.min() <= box_tol * max(1.0, np.abs(ref_boxes[i]).max()), (i, ref_boxes[i], ref_scores[i])
        used[cand[d.argmin()]] = True


TINY = ["tiny_r50_s1x_a", "tiny_r50_s1x_b", "tiny_r50_legacy", "tiny_r101_s1x", "tiny_r50_dl", "tiny_r101_dl"]


@pytest.mark.parametrize("name", TINY + ["full_r50_s1x_small", "full_r101_s1x_small", "full_r50_legacy_small", "full_r50_s1x_800x1333", "full_r50_dl_p28", "full_r101_dl_p28_small", "tiny_r101_dl_p28_video"])
def test_fp32_matches_reference_golden(name):
    from oracle.ref_cpu import extract_iuv
    meta, z, cfg, pred, out = _run(name, "fp32", keep=True)
    s = meta["iuv_stride"]
    assert out["image_size"].tolist() == z["out/image_size"].tolist()
    R = z["out/scores"].shape[0]
    assert out["scores"].shape[0] == R, "selection mismatch: %d detections vs %d in the reference" % (out["scores"].shape[0], R)
    assert out["pred_classes"].dtype == torch.int64 and int(out["pred_classes"].abs().sum()) == 0
    np.testing.assert_allclose(out["pred_boxes"].numpy(), z["out/pred_boxes"], atol=2e-3, rtol=1e-5)
    np.testing.assert_allclose(out["scores"].numpy(), z["out/scores"], atol=1e-5)
    for k in IUV_KEYS:
        got = out[k].numpy()[:, :, ::s, ::s]
        assert got.shape == z["out/" + k].shape and out[k].dtype == torch.float32
        err = np.abs(got - z["out/" + k]).max()
        assert err <= IUV_ATOL, (k, err)
    # part-index argmax (visualizer.py:10-17), bit-exact on EVERY golden: the golden's labels / uv were recorded from the
    # reference's full-resolution maps, whatever stride its stored IUV tensors are subsampled with
    # (resampled to the reference's integer box sizes: the boxes themselves are compared above, and a 1e-3 px difference next
    # to an integer boundary would change the resample SIZE, which is not what this check is about)
    vis_in = dict(out)
    vis_in["pred_boxes"] = torch.from_numpy(z["out/pred_boxes"])
    for i, (labels, uv) in enumerate(extract_iuv(vis_in)):
        np.testing.assert_array_equal(labels.numpy().astype(np.uint8), z["vis/labels_%d" % i])
        np.testing.assert_allclose(uv.numpy(), z["vis/uv_%d" % i], atol=IUV_ATOL)
    for nm, cs in meta.get("stage_channel_stride", {}).items():   # DeepLab pool-28 cases: the head output, channel-subsampled
        ref = z["stage/" + nm]
        got = _nchw(pred.engine.inter[nm]).numpy()[:, ::cs][:, : ref.shape[1]]
        assert np.abs(got - ref).max() <= 5e-4 * max(1.0, np.abs(ref).max()), nm
    if "stage/p2" in z.files:
        inter = pred.engine.inter
        for k in ("p2", "p3", "p4", "p5", "p6"):
            ref = z["stage/" + k]
            got = _nchw(inter[k]).numpy()[:, : ref.shape[1]]
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k
        props, pscores, pcounts = inter["proposals"]
        n = int(pcounts[0])
        assert n == z["stage/proposal_boxes"].shape[0]
        _assert_rows_match(props[0, :n].cpu().numpy(), pscores[0, :n].cpu().numpy(), z["stage/proposal_boxes"],
                           z["stage/objectness_logits"])
        if cfg.dp_decoder_on:
            ref = z["stage/decoder_out"]
            got = _nchw(inter["decoder_out"]).numpy()[:, : ref.shape[1]]
            assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())
        ref = z["stage/dp_head_out"]
        got = _nchw(inter["dp_head_out"]).numpy()[:, : ref.shape[1]]
        assert np.abs(got - ref).max() <= 5e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", ["tiny_r50_s1x_a", "tiny_r50_legacy", "tiny_r50_dl", "full_r50_s1x_small", "full_r101_s1x_small", "full_r50_legacy_small", "full_r50_s1x_800x1333"])
def test_fp32_matches_cpu_oracle_live(name):
    """Same seeded inputs through the oracle on the host and the HIP path on the GPU: every pixel of the full-resolution
    IUV maps (the goldens of the full-width cases store a subsample), full channel width included - the BASELINE.json
    configs[1] geometry (800x1333, R = 8) too: stride-1 IUV of all 77 x 112 x 112 values per detection."""
    from oracle.ref_cpu import OracleModel
    meta, z, cfg, pred, out = _run(name, "fp32")
    _, state, img = golden_case_inputs(meta)
    ref = OracleModel(cfg, state)(torch.from_numpy(img))
    assert set(out) == set(ref)
    for k in ref:
        assert out[k].shape == ref[k].shape and out[k].dtype == ref[k].dtype, k
    for k in IUV_KEYS:
        assert (out[k] - ref[k]).abs().max().item() <= IUV_ATOL, k


def test_cuda_reference_nms_strategy_matches_oracle():
    """nms_reference="cuda": torchvision's batched_nms switches strategy at 20000 box elements in the reference's CUDA mode
    (run.py:22-29) instead of 4000 on the CPU. full_r50_s1x_small feeds the RPN NMS ~3400 boxes = ~13.5k elements: the
    per-level loop under "cpu", the coordinate-offset trick under "cuda". The oracle restates both (nms_trick_max_numel)."""
    from densepose_torchscript_amd.predictor import DensePosePredictor
    from oracle.ref_cpu import OracleModel
    meta, z = load_golden("full_r50_s1x_small")
    cfg, state, img = golden_case_inputs(meta)
    pred = DensePosePredictor(cfg, state, dtype="fp32", nms_reference="cuda", check_keep=True)
    pred.engine.keep_intermediates = True
    out = {k: v.cpu() for k, v in pred(torch.from_numpy(img)).items()}
    _, _, pcounts = pred.engine.inter["proposals"]
    cb, cs, cl, cv = pred.engine.inter["cand"]
    n_valid = int(cv[0].sum())
    assert 4000 < 4 * n_valid <= 20000, n_valid          # between the two switches: the two references take different strategies
    ref = OracleModel(cfg, state, nms_trick_max_numel=20000)(torch.from_numpy(img))
    assert out["scores"].shape == ref["scores"].shape
    np.testing.assert_allclose(out["pred_boxes"].numpy(), ref["pred_boxes"].numpy(), atol=2e-3, rtol=1e-5)
    np.testing.assert_allclose(out["scores"].numpy(), ref["scores"].numpy(), atol=1e-5)
    for k in IUV_KEYS:
        assert (out[k] - ref[k]).abs().max().item() <= IUV_ATOL, k


def test_batch_equals_single_calls():
    """a batch of N frames == N independent calls (bit-exact on the same device path)."""
    from densepose_torchscript_amd.predictor import DensePosePredictor
    meta, z = load_golden("tiny_r50_s1x_a")
    cfg, state, img = golden_case_inputs(meta)
    pred = DensePosePredictor(cfg, state, dtype="fp32")
    rng = np.random.default_rng(5)
    imgs = [torch.from_numpy(rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)) for _ in range(3)] + [
        torch.from_numpy(rng.integers(0, 256, (150, 100, 3), dtype=np.uint8))]
    batch = pred.predict_batch(imgs)
    for im, b in zip(imgs, batch):
        single = pred(im)
        for k in single:
            assert torch.equal(single[k].cpu(), b[k].cpu()), k


@pytest.mark.parametrize("dtype,cfgname", [("bf16", "densepose_rcnn_R_50_FPN_s1x"), ("fp16", "densepose_rcnn_R_50_FPN_s1x"),
                                           ("bf16", "densepose_rcnn_R_101_FPN_s1x"),      # BASELINE.json configs[2]: 33-block trunk in its stated dtype
                                           ("bf16", "densepose_rcnn_R_50_FPN_DL_s1x"), ("fp16", "densepose_rcnn_R_101_FPN_DL_s1x")])
def test_full_width_batch_equals_single_calls_16bit(dtype, cfgname):
    """Same property at full channel width in the 16-bit modes, where every conv kernel family is in play (256x256 /
    256x128 / 128x128 ring tiles, streaming 1x1, generic): a frame's result does not depend on what else is in the batch
    (tiles that span two frames, persistent waves that walk across frames), graph replay and pipeline lanes included."""
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    # (the DeepLab variants = BASELINE.json configs[3] / [4]: GroupNorm / pool / broadcast over device-sized ROI counts, R101 trunk)
    cfg = get_config(cfgname, ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 400, "TEST.DETECTIONS_PER_IMAGE", 6])
    state = make_synthetic_state(cfg, 3)
    rng = np.random.default_rng(11)
    imgs = [torch.from_numpy(rng.integers(0, 256, (256, 400, 3), dtype=np.uint8)).cuda() for _ in range(6)]
    single = DensePosePredictor(cfg, state, dtype=dtype, resize="device")
    want = [single(im) for im in imgs]
    torch.cuda.synchronize()
    batched = DensePosePredictor(cfg, state, dtype=dtype, resize="device", use_graphs=True, pipeline_depth=2)
    for order in (list(range(6)), [5, 0, 3, 1, 4, 2]):
        got = batched.predict_batch([imgs[i] for i in order])
        batched.join()
        torch.cuda.synchronize()
        for i, g in zip(order, got):
            for k in g:
                assert torch.equal(want[i][k].cpu(), g[k].cpu()), (dtype, order, i, k)
    assert sum(int(w["scores"].shape[0]) for w in want) > 0


def test_rpn_split_option_is_batch_invariant_and_close_to_the_default():
    """EngineOptions.rpn_split_min_hw (an A/B switch, default off: the RPN's large levels on kernel class 10 with the two 1x1 heads as a
    second launch): chosen by a level's size per image alone, so a frame's result still does not depend on the batch; and against the
    default path (LDS-ring kernel with the heads in its epilogue - another summation order of the same sums) the proposals move by
    rounding steps only: same detections within 1.5 px / 0.05."""
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.options import EngineOptions
    from densepose_torchscript_amd.predictor import DensePosePredictor
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 400, "TEST.DETECTIONS_PER_IMAGE", 6])
    state = make_synthetic_state(cfg, 3)
    rng = np.random.default_rng(11)
    imgs = [torch.from_numpy(rng.integers(0, 256, (256, 400, 3), dtype=np.uint8)).cuda() for _ in range(4)]
    opt = EngineOptions(rpn_split_min_hw=128)
    single = DensePosePredictor(cfg, state, dtype="bf16", resize="device", options=opt)
    want = [single(im) for im in imgs]
    names = set()
    single.engine.prof = []
    single(imgs[0])
    torch.cuda.synchronize()
    names = {rec[4] for rec in single.engine.prof}
    single.engine.prof = None
    assert any(n.startswith("rpn_head ") for n in names), sorted(names)[:8]      # the heads did run as a launch of their own
    batched = DensePosePredictor(cfg, state, dtype="bf16", resize="device", use_graphs=True, pipeline_depth=2, options=opt)
    got = batched.predict_batch(imgs)
    batched.join()
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for k in g:
            assert torch.equal(w[k].cpu(), g[k].cpu()), k
    ref = [DensePosePredictor(cfg, state, dtype="bf16", resize="device")(im) for im in imgs[:2]]
    for w, r in zip(want, ref):
        assert abs(int(w["scores"].shape[0]) - int(r["scores"].shape[0])) <= 1
        n = min(int(w["scores"].shape[0]), int(r["scores"].shape[0]))
        if n and w["scores"].shape[0] == r["scores"].shape[0]:
            assert float((w["pred_boxes"] - r["pred_boxes"]).abs().max()) <= 1.5
            assert float((w["scores"] - r["scores"]).abs().max()) <= 0.05


def test_full_width_network_runs_on_the_documented_kernel_classes():
    """DESIGN.md 4.1's table is what the engine really launches: one full-width bf16 frame with the per-launch profile on, and the
    kernel classes of the trunk / FPN / heads are the weight-stationary pointwise kernels (res4 / res5 conv1 + conv3, the laterals, fc2),
    the res3 pair kernel, the res2 tail kernel, the weight-stationary 3x3 kernels and ONE grouped launch for the predictor's four
    sub-pixel classes - and none of them was silently replaced by the generic kernel."""
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 400, "TEST.DETECTIONS_PER_IMAGE", 6])
    pred = DensePosePredictor(cfg, make_synthetic_state(cfg, 3), dtype="bf16", resize="device")
    rng = np.random.default_rng(11)
    img = torch.from_numpy(rng.integers(0, 256, (256, 400, 3), dtype=np.uint8)).cuda()
    pred(img)
    eng = pred.engine
    eng.prof = []
    out = pred(img)
    torch.cuda.synchronize()
    prof, eng.prof = eng.prof, None
    assert int(out["scores"].shape[0]) > 0
    by_layer = {}
    for cls, _, _, _, name, _ in prof:
        by_layer.setdefault(name.split(" ")[0], []).append(cls)
    def cls_of(prefix):
        hit = [c for n, cs in by_layer.items() if prefix in n for c in cs]
        assert hit, (prefix, sorted(by_layer))
        return hit
    for lname in ("res4.1.conv1", "res4.5.conv3", "res5.1.conv1", "res5.2.conv3", "fpn_lateral4", "fpn_lateral5", "fpn_lateral3"):
        assert all(c.startswith(("conv1x1_pws_kernel", "conv1x1_pwq_kernel")) for c in cls_of(lname)), (lname, cls_of(lname))
    assert all(c == "bottleneck_pair128_kernel" for c in cls_of("res3.1.conv3")), cls_of("res3.1.conv3")
    assert all(c == "bottleneck_tail64_kernel" for c in cls_of("res2.1.conv2")), cls_of("res2.1.conv2")
    assert all(c.startswith("conv3x3_ws1") for c in cls_of("res4.2.conv2")), cls_of("res4.2.conv2")      # kernel class 10 (dp_conv_wq.hip)
    assert all(c.startswith("conv3x3_wsr_kernel<128") for c in cls_of("res3.1.conv2")), cls_of("res3.1.conv2")
    dec = cls_of("dp_predictor")      # one grouped launch where the LDS-ring kernels take the shape, else the four launches
    assert (len(dec) == 1 and dec[0].endswith(",x4>")) or len(dec) == 4, dec


def test_device_resize_equals_host_resize():
    meta, z, cfg, pred, out_h = _run("tiny_r50_s1x_b", "fp32", resize="host")
    _, _, _, _, out_d = _run("tiny_r50_s1x_b", "fp32", resize="device")
    for k in out_h:
        assert torch.equal(out_h[k], out_d[k]), k


# part-label agreement with the fp32 golden on the matched detections, measured with tools/measure_bands.py (round 3): bf16 0.996 /
# 0.965 / 0.985 (tiny_r50_s1x_a / full_r50_s1x_small / the 800x1333 headline frame), fp16 1.0 / 0.9988 / 0.9987; the DeepLab cases
# have tiny boxes (3 - 384 label pixels in all: one flipped pixel is 0.3 - 33 %), bf16 0.67 - 0.95, fp16 0.948 - 0.996
BF16_LABEL_FLOOR = {"tiny_r50_s1x_a": 0.97, "full_r50_s1x_small": 0.93, "full_r50_s1x_800x1333": 0.93}
# largest IUV deviation on the matched detections, relative to the largest reference value of the same map (tools/measure_bands.py):
# 0.016 / 0.027 on the s1x cases - held to 3x that. (The DeepLab pool-28 case left this band test in round 4: it is gated against the
# storage-emulating oracle instead - test_16bit_layers_equal_the_storage_oracle_teacher_forced / test_16bit_end_to_end_... above.)
# (the headline frame: 7 of 8 detections within 1.5 px; one of the matched ones sits 1 px off and its maps deviate by 0.18 of their range)
# (round 5, full_r50_s1x_small: with seeded random weights the number is chaotic in the last place of ANY layer - the same build reads
# 0.038 with res2.0's projection shortcut stored as a tensor and 0.203 with it kept in fp32 inside the fused tail (one rounding FEWER; the
# headline frame moves the other way, 0.122 -> 0.095 and 7 -> 8 of 8 detections matched; tools/band_case.py prints both, detection by
# detection: one box at the image corner moves by 0.27 px and its maps by 0.19 of their range). The reference's OWN bf16 run reads 0.223
# on this case (tests/golden/full_r50_s1x_small__bf16.npz, tests/yardstick.py) and matches 2 of 4 boxes where this engine matches 3:
# the band is 1.25 x that, the reference-derived test below holds the engine to it case by case.)
# round 6 (tools/band_values.py, profiles/r6_band_values.txt; the kernels are deterministic, the numbers are the same on every box): 0.0167 / 0.2020 /
# 0.0373 with 3 of 3 / 3 of 4 / 7 of 8 detections matched - held to 1.25 x that. The storage-emulating oracle (per layer, teacher forced) carries
# the parity claim for the 16-bit modes; these bands only catch a regression of the whole.
BF16_IUV_BAND = {"tiny_r50_s1x_a": 0.021, "full_r50_s1x_small": 0.2525, "full_r50_s1x_800x1333": 0.0467}
FP16_LABEL_FLOOR = 0.9


def _match_to_reference_sub(out, z, s, box_tol, score_tol):
    """_match_to_reference for an `out` whose IUV maps are ALREADY subsampled like the golden's (the __half fixtures)."""
    return _match_to_reference(out, z, s, box_tol, score_tol)


def _match_to_reference(out, z, s, box_tol, score_tol):
    """Every reference detection matched to its nearest output row -> (hits, largest IUV deviation on the matched rows,
    relative to the largest reference value of the same map)."""
    gb, gs, rb, rs = out["pred_boxes"].numpy(), out["scores"].numpy(), z["out/pred_boxes"], z["out/scores"]
    hits, iuv = 0, 0.0
    for i in range(len(rb)):
        if len(gb) == 0:
            break
        d = np.abs(gb - rb[i]).max(axis=1)
        j = int(d.argmin())
        if d[j] < box_tol and abs(gs[j] - rs[i]) < score_tol:
            hits += 1
            for k in IUV_KEYS:
                ref = z["out/" + k][i]
                iuv = max(iuv, float(np.abs(out[k][j].numpy()[:, ::s, ::s] - ref).max()) / max(float(np.abs(ref).max()), 1e-6))
    return hits, iuv


def _label_agreement(out, z, box_tol):
    """Part-index agreement in a 16-bit mode (the north star's second parity clause is stated for fp32, where it is bit-exact -
    test_fp32_matches_reference_golden): the visualiser's labels computed from THIS run's maps of every matched detection, on
    the reference's box of that detection, against the golden's full-resolution labels -> fraction of equal pixels."""
    from oracle.ref_cpu import extract_iuv
    gb, rb = out["pred_boxes"].numpy(), z["out/pred_boxes"]
    eq = tot = 0
    for i in range(len(rb)):
        if len(gb) == 0:
            break
        d = np.abs(gb - rb[i]).max(axis=1)
        j = int(d.argmin())
        if d[j] >= box_tol:
            continue
        sub = {k: out[k][j:j + 1] for k in IUV_KEYS}
        sub["pred_boxes"] = torch.from_numpy(rb[i:i + 1])
        (labels, _), = extract_iuv(sub)
        want = z["vis/labels_%d" % i]
        eq, tot = eq + int((labels.numpy().astype(np.uint8) == want).sum()), tot + want.size
    return eq / max(tot, 1), tot


# ---------------------------------------------------------------------------------------------------------------------------
# The 16-bit modes against the STORAGE-EMULATING oracle (oracle/ref_storage.py): the CPU restatement with every tensor rounded to
# the storage type where the engine stores it. tools/emul_layers.py / tools/emul_stats.py print what the bounds below were read from.
TOP_ULP = {"bf16": 2.0 ** -7, "fp16": 2.0 ** -10}     # one unit in the last place of a tensor's largest value, relative to it
FORCED_CASES = [("full_r50_s1x_small", "bf16"), ("full_r50_s1x_small", "fp16"), ("full_r50_s1x_800x1333", "bf16"), ("full_r101_s1x_small", "bf16"),
                ("full_r50_dl_p28", "bf16"), ("full_r101_dl_p28_small", "fp16"), ("tiny_r101_dl_p28_video", "fp16"), ("tiny_r101_dl_p28_video", "bf16"), ("tiny_r50_legacy", "bf16"),
                ("full_r50_legacy_small", "bf16")]


@pytest.mark.parametrize("name,dt", FORCED_CASES)
def test_16bit_layers_equal_the_storage_oracle_teacher_forced(name, dt):
    """PARITY of the throughput dtypes, layer by layer: every convolution (and GroupNorm) output of the engine against the oracle's
    computation of that layer FROM THE ENGINE'S OWN INPUT TENSOR (teacher forcing). The two then differ by the order of the fp32
    accumulation alone: a handful of elements per ten thousand land on the other side of a rounding boundary - by one unit in the
    last place, or by less than that of the tensor's largest value where cancellation left a small result. Measured (round 4,
    tools/emul_layers.py): 0.0001 - 0.2 % of a layer's elements, <= 0.7 ulp of the top value for convolutions; GroupNorm <= 0.03 % and
    <= 0.4 ulp at the real DeepLab geometry (full_r50_dl_p28 = BASELINE.json configs[3]: the bf16 GroupNorm / dilated-tap kernels are
    right; the low label agreement of that mode with the fp32 golden is rounding noise through eight normalisations of random-weight
    maps), up to 36 % one-ulp flips in the tiny-width fixtures whose groups hold ONE channel. The fp32 IUV maps computed from the
    forced head output agree to 1e-6 and the part labels are identical."""
    from emul_common import run_forced
    stats, iuv, (npx, ndiff, margin) = run_forced(name, dt)
    assert len(stats) >= 30
    tiny = name.startswith("tiny")
    for layer, st in stats.items():
        frac = st["differ"] / st["n"]
        if layer.startswith("gn:"):
            assert st["max_rel_to_top"] <= (4.0 if not tiny else 8.0) * TOP_ULP[dt], (layer, st)
            assert tiny or frac <= 2e-3, (layer, st)
        else:
            assert st["max_rel_to_top"] <= 1.5 * TOP_ULP[dt], (layer, st)
            assert frac <= (5e-3 if not tiny else 4e-2), (layer, st)
    for k, st in iuv.items():
        assert st["max_rel_to_top"] <= 1e-5, (k, st)
    assert ndiff <= 2 and margin <= 1e-4, (npx, ndiff, margin)


def test_16bit_layers_equal_the_storage_oracle_with_the_rounding_point_fusions_off():
    """The three round-3 fusions that change where a value is rounded - the projection shortcut as K planes of conv3, the decoder's
    level sum inside the convolutions, split-K - switched OFF: the engine then stores the shortcut tensor and every decoder head, and
    the oracle, told so, must again agree layer by layer (the default test above covers them switched on: the 800 x 1333 case folds
    the decoder, every full-width case fuses three shortcuts)."""
    from emul_common import run_forced
    stats, iuv, (npx, ndiff, margin) = run_forced("full_r50_s1x_800x1333", "bf16", {"fuse_shortcut": False, "decoder_fold": False, "split_k_on": False})
    assert "backbone.bottom_up.res3.0.shortcut" in stats and "backbone.bottom_up.res3.0.conv3+shortcut" not in stats
    for layer, st in stats.items():
        assert st["max_rel_to_top"] <= 1.5 * TOP_ULP["bf16"] and st["differ"] <= 5e-3 * st["n"], (layer, st)
    for k, st in iuv.items():
        assert st["max_rel_to_top"] <= 1e-5, (k, st)
    assert ndiff <= 2 and margin <= 1e-4


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_rounding_point_fusions_end_to_end_ab(dtype):
    """... and engine against engine on the headline frame: with the three fusions off the result moves - they move rounding points -
    but stays inside the end-to-end bound of two runs that differ by rounding flips (6 ulp of a map's top value; measured 1.5 - 3):
    same detections, IUV maps within the bound. A defect in one of those kernels' epilogues would show here end to end."""
    from densepose_torchscript_amd.predictor import DensePosePredictor
    meta, z = load_golden("full_r50_s1x_800x1333")
    cfg, state, img = golden_case_inputs(meta)
    outs = []
    for on in (True, False):
        pred = DensePosePredictor(cfg, state, dtype=dtype, check_keep=True)
        pred.engine.fuse_shortcut = pred.engine.decoder_fold = pred.engine.split_k_on = on
        outs.append({k: v.cpu() for k, v in pred(torch.from_numpy(img)).items()})
    a, b = outs
    # random-weight scores crowd together: a borderline detection may come or go and near-equal scores swap places - match boxes
    ba, bb = a["pred_boxes"].numpy(), b["pred_boxes"].numpy()
    assert abs(len(ba) - len(bb)) <= 1 and len(ba) >= 7
    pairs = []
    for i in range(len(ba)):
        d = np.abs(bb - ba[i]).max(axis=1)
        j = int(d.argmin())
        if d[j] <= 1.5 and abs(float(a["scores"][i]) - float(b["scores"][j])) <= 0.05:
            pairs.append((i, j))
    assert len(pairs) >= len(ba) - 2, (len(pairs), len(ba))      # (measured: 6 of 8 in bf16, 8 of 8 in fp16)
    worst = 0.0
    for k in IUV_KEYS:
        top = float(a[k].abs().max())
        for i, j in pairs:
            worst = max(worst, float((a[k][i] - b[k][j]).abs().max()) / top)
    # (the matched boxes differ by a fraction of a pixel, which moves the ROIAlign samples: the maps of a detection then differ by
    # more than rounding alone would give - measured 0.06 of the top value in fp16). A regression guard with 2x headroom; the parity
    # statement for these kernels is the teacher-forced test above
    assert worst <= (0.25 if dtype == "bf16" else 0.12), worst


E2E_EMUL_CASES = [("full_r50_s1x_800x1333", "bf16"), ("full_r50_s1x_800x1333", "fp16"), ("full_r101_s1x_small", "bf16"), ("full_r50_dl_p28", "bf16"),
                  ("tiny_r50_s1x_a", "bf16"), ("tiny_r50_legacy", "bf16"), ("full_r50_legacy_small", "bf16")]


@pytest.mark.parametrize("name,dt", E2E_EMUL_CASES)
def test_16bit_end_to_end_against_the_storage_oracle(name, dt):
    """The same comparison WITHOUT forcing: engine and oracle each run their own 60 layers (the oracle's DensePose branch on the
    engine's detections, so that a borderline detection is out of the picture). A one-ulp flip in one layer moves hundreds of sums of
    the next by a fraction of their own ulp, so the two runs decorrelate at the one-ulp level within a few layers; what stays bounded
    is the deviation relative to a tensor's scale. Measured (tools/emul_stats.py): FPN maps / decoder / head output <= 1.7 ulp of the
    top value (bf16 0.0055 - 0.0135, fp16 0.0008 - 0.0017), IUV maps <= 3.4 (bf16 0.027, fp16 0.0019); part labels differ on 0.1 - 1.8 %
    of the box pixels, every one of them a pixel whose decision margin is inside the IUV deviation. Held to ~1.5x - 2x that."""
    from emul_common import label_stats, run_pair, stage_stats
    r = run_pair(name, dt)
    assert r["R"] > 0
    for k, (got, ref) in r["stages"].items():
        st = stage_stats(got, ref, dt)
        # (the DeepLab head normalises eight times: its output decorrelates further - 3.0 ulp measured with the 32-pixel row kernel's
        # summation order, 2.4 with the ring kernel's)
        bound = (6.0 if k.startswith("pred_") else 4.5 if ("_dl_" in name and k == "dp_head_out") else 3.0) * TOP_ULP[dt]
        assert st["max_rel_to_top"] <= bound, (k, st)
    npx, ndiff, margin = label_stats(r)
    iuv_dev = max(stage_stats(*r["stages"][k], dt)["max_rel_to_top"] * float(r["stages"][k][1].abs().max()) for k in IUV_KEYS[:2])
    assert npx > 0 and ndiff <= 0.03 * npx, (npx, ndiff)
    # a pixel may only differ where the oracle's own decision is closer than (twice) the largest logit deviation of this run
    assert margin <= 2.0 * iuv_dev + 1e-6, (npx, ndiff, margin, iuv_dev)


# bf16 has 8 significant bits and these are RANDOM-weight networks (no trained smoothness): through ~60 layers the IUV logits of
# a matched detection move by 1 - 20 % of their range (tools/measure_bands.py prints the per-detection numbers; bench.py reports
# the same quantities for the headline workload: 1 - 4 % there), and a borderline detection may be replaced by another one.
# The bands below are those measurements with ~2x headroom - a regression guard for the throughput mode, not a parity claim.
@pytest.mark.parametrize("name", ["tiny_r50_s1x_a", "full_r50_s1x_small", "full_r50_s1x_800x1333"])
def test_bf16_mode_stays_in_its_measured_band(name):
    """Throughput mode (bf16 operands, fp32 accumulate) against the fp32 reference golden: the detections are found (box within
    1.5 px, score within 0.05; at most one borderline detection may come or go) and the IUV maps of the matched detections
    stay within the per-case band BF16_IUV_BAND of the map's range - the headline frame (BASELINE.json configs[1], 800 x 1333, R = 8)
    included: every fusion that moves a rounding point moves THIS number, and bench.py only prints it."""
    meta, z, cfg, pred, out = _run(name, "bf16")
    for k in IUV_KEYS:
        assert torch.isfinite(out[k]).all()
    R = z["out/scores"].shape[0]
    assert abs(out["scores"].shape[0] - R) <= 1
    hits, iuv = _match_to_reference(out, z, meta["iuv_stride"], 1.5, 0.05)
    assert hits >= R - 1, (hits, R)
    assert iuv <= BF16_IUV_BAND[name], iuv
    agree, npx = _label_agreement(out, z, 1.5)
    assert npx > 0 and agree >= BF16_LABEL_FLOOR[name], (name, agree)


@pytest.mark.parametrize("name", ["tiny_r50_s1x_a", "full_r50_s1x_small", "tiny_r50_dl", "full_r50_dl_p28", "tiny_r101_dl_p28_video"])
def test_fp16_mode_matches_reference_half_semantics(name):
    """the reference's `.half()` export (export.py:36-37, run.py:26) = IEEE-half GEMM operands. Boxes, anchors,
    decode, NMS and softmax stay fp32 here (box_regression.py:84, nms.py:20 upcast in the reference too). 10 mantissa bits
    instead of bf16's 7: the measured distance to the fp32 golden must be well inside the bf16 one."""
    meta, z, cfg, pred, out = _run(name, "fp16")
    for k in IUV_KEYS:
        assert torch.isfinite(out[k]).all()
    R = z["out/scores"].shape[0]
    assert abs(out["scores"].shape[0] - R) <= 1
    # borderline detections (score next to the 0.3 threshold, degenerate boxes of the tiny random-weight cases) may come or go
    # and near-equal scores may swap places: match every reference detection to its nearest output row, allow one miss
    # (tiny_r101_dl_p28_video = BASELINE.json configs[4]'s combination: R101 + DeepLab pool 28, fp16, 1080x1920 video frame)
    hits, iuv = _match_to_reference(out, z, meta["iuv_stride"], 0.5, 0.02)   # half a pixel, 0.02 of score
    assert hits >= R - 1, (hits, R)
    assert iuv <= 0.12, iuv     # measured: <= 0.06 of the map's range on the matched detections (bf16: up to 0.3)
    agree, npx = _label_agreement(out, z, 0.5)
    assert npx > 0 and agree >= FP16_LABEL_FLOOR, (name, agree)


@pytest.mark.parametrize("name", ["tiny_r50_s1x_a", "full_r50_s1x_small", "tiny_r50_dl", "full_r101_dl_p28_small"])
def test_fp16_mode_against_the_reference_run_in_half(name):
    """The reference's OWN fp16 mode (`predictor.half()`, run.py:26) recorded on the CPU (tests/golden/<case>__half.npz,
    oracle/make_goldens.py --half): ATen's CPU half kernels round every layer's output to half like this engine does, but keep
    FrozenBN as separate half operations where the engine folds it into the weights - so this is a second yardstick, not a
    bit-exact oracle. Held here: (1) the engine's fp16 outputs are no further from the fp32 golden than the reference's own
    fp16 outputs are, within a factor 2; (2) engine-fp16 and reference-fp16 find the same detections."""
    meta, z, cfg, pred, out = _run(name, "fp16")
    _, zh = load_golden(name + "__half")
    s = meta["iuv_stride"]
    R = z["out/scores"].shape[0]
    assert abs(zh["out/scores"].shape[0] - R) <= 1 and abs(out["scores"].shape[0] - R) <= 1
    # distance of the REFERENCE's fp16 run to its fp32 run, on the detections both have
    ref_out = {k: torch.from_numpy(zh["out/" + k]) for k in IUV_KEYS + ("pred_boxes", "scores")}
    ref_hits, ref_iuv = _match_to_reference_sub(ref_out, z, 1, 0.5, 0.02)
    hits, iuv = _match_to_reference(out, z, s, 0.5, 0.02)
    assert hits >= ref_hits - 1, (hits, ref_hits)
    assert iuv <= max(2.0 * ref_iuv, 0.02), (iuv, ref_iuv)
    # engine fp16 vs reference fp16 directly
    hits_h, iuv_h = _match_to_reference(out, zh, s, 0.5, 0.02)
    assert hits_h >= zh["out/scores"].shape[0] - 1, hits_h
    assert iuv_h <= max(3.0 * ref_iuv, 0.03), (iuv_h, ref_iuv)


BF16_REF_CASES = ["tiny_r50_s1x_a", "full_r50_s1x_small", "tiny_r50_dl", "full_r50_dl_p28", "full_r101_dl_p28_small", "full_r50_s1x_800x1333"]


@pytest.mark.parametrize("name", BF16_REF_CASES)
def test_bf16_mode_against_the_reference_run_in_bfloat16(name):
    """The dtype every throughput configuration of BASELINE.json names, held to the reference's OWN run in that dtype
    (`predictor.bfloat16()` recorded on the CPU: tests/golden/<case>__bf16.npz, oracle/make_goldens.py --bf16; the reference picks its
    dtype by such a module call, run.py:20-29 / export.py:36-37). ATen's CPU bf16 kernels round every layer's output - and FrozenBN's
    intermediate results, and the box arithmetic - to 8 significant bits; the engine folds FrozenBN into the weights, accumulates in fp32
    and keeps boxes / scores / NMS in fp32. So: the engine's bf16 outputs must be NO FURTHER from the reference's fp32 outputs than the
    reference's own bf16 outputs are - detections found, IUV deviation on the matched ones, part-label agreement (tests/yardstick.py:
    one definition for both sides)."""
    from oracle.ref_cpu import extract_iuv
    from yardstick import engine_distance, reference_lowp_distance
    meta, z, cfg, pred, out = _run(name, "bf16")
    ref = reference_lowp_distance(name, "bf16")
    eng = engine_distance(out, z, meta["iuv_stride"], extract_iuv)
    print("\n%s: engine bf16 %s\n%s  reference bf16 %s" % (name, eng, " " * len(name), ref))
    R = eng["ref_detections"]
    assert abs(eng["detections"] - R) <= 1
    assert eng["box_match_rate"] >= ref["box_match_rate"] - 1e-9, (eng, ref)              # finds at least as many of the fp32 detections
    if ref["label_pixels"] > 0 and eng["label_pixels"] > 0:
        assert eng["label_agreement"] >= ref["label_agreement"] - 0.02, (eng, ref)        # part labels at least as close (2 % slack: other matched set)
    if ref["box_match_rate"] > 0:
        assert eng["iuv"] <= max(1.5 * ref["iuv"], 0.08), (eng, ref)                      # IUV maps of the matched detections


def test_missing_gpu_or_library_fails_loudly(monkeypatch):
    from densepose_torchscript_amd import lib
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libdensepose_hip.so")
    monkeypatch.setattr(lib, "_lib", None)
    with pytest.raises(lib.DensePoseHipError):
        lib.load()


def test_streams_and_graphs_do_not_change_results():
    """Sub-batches on several HIP streams and HIP-graph replay of the static part give bit-identical outputs."""
    from densepose_torchscript_amd.predictor import DensePosePredictor
    meta, z = load_golden("tiny_r50_s1x_a")
    cfg, state, img = golden_case_inputs(meta)
    rng = np.random.default_rng(6)
    imgs = [torch.from_numpy(rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)) for _ in range(6)]
    base = DensePosePredictor(cfg, state, dtype="fp32", num_streams=1).predict_batch(imgs)
    for streams, graphs in ((2, False), (3, True), (1, True)):
        pred = DensePosePredictor(cfg, state, dtype="fp32", num_streams=streams, use_graphs=graphs)
        first = pred.predict_batch(imgs)   # 1st call captures, later calls replay
        for _ in range(2):
            out = pred.predict_batch(imgs[::-1])   # other inputs through the same graph: must not disturb `first`
        out = pred.predict_batch(imgs)
        torch.cuda.synchronize()
        for res in (first, out):
            for a, b in zip(base, res):
                for k in a:
                    assert torch.equal(a[k].cpu(), b[k].cpu()), (streams, graphs, k)


def test_pipeline_lanes_do_not_change_results():
    """pipeline_depth = 2: consecutive batches alternate between two stream lanes (own HIP graph instance each) and the
    caller's stream only waits in join(); outputs are bit-identical to the plain one-stream calls."""
    from densepose_torchscript_amd.predictor import DensePosePredictor
    meta, z = load_golden("tiny_r50_s1x_a")
    cfg, state, img = golden_case_inputs(meta)
    rng = np.random.default_rng(7)
    batches = [[torch.from_numpy(rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)).cuda() for _ in range(4)] for _ in range(5)]
    plain = DensePosePredictor(cfg, state, dtype="fp32", resize="device")
    want = [plain.predict_batch(b) for b in batches]
    torch.cuda.synchronize()
    for graphs in (False, True):
        pred = DensePosePredictor(cfg, state, dtype="fp32", resize="device", use_graphs=graphs, pipeline_depth=2)
        got = [pred.predict_batch(b) for b in batches]   # lanes 0,1,0,1,0: graph capture on first use of a lane, replays after
        pred.join()
        torch.cuda.current_stream().synchronize()
        for w, g in zip(want, got):
            for a, b in zip(w, g):
                for k in a:
                    assert torch.equal(a[k].cpu(), b[k].cpu()), (graphs, k)


def test_run_cli_frame_sequence_equals_single_frames(tmp_path):
    """run.py on a [T,H,W,3] .npy (the reference's video loop, run.py:42-57): batched + pipelined == frame by frame."""
    import subprocess
    import sys
    import os
    from densepose_torchscript_amd import TINY_OPTS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(21)
    frames = rng.integers(0, 256, (5, 96, 160, 3), dtype=np.uint8)
    np.save(tmp_path / "clip.npy", frames)
    # a tiny yaml-free config: the CLI accepts a variant name + `--opts KEY VALUE ...` overrides
    def run(inp, out):
        subprocess.check_call([sys.executable, os.path.join(root, "run.py"), "densepose_rcnn_R_50_FPN_s1x", "synthetic:4", str(inp),
                               "--out", str(out), "--fp32", "--batch", "2", "--min_score", "0.05", "--opts"] + [str(o) for o in TINY_OPTS], cwd=root)
        return np.load(out)
    seq = run(tmp_path / "clip.npy", tmp_path / "seq.npz")["iuv"]
    assert seq.shape == (5, 3, 96, 160)
    for t in (0, 3, 4):
        np.save(tmp_path / "one.npy", frames[t])
        one = run(tmp_path / "one.npy", tmp_path / "one.npz")["iuv"]
        assert np.array_equal(seq[t], one), t
    assert seq.any()


def test_zero_and_many_detections():
    """R = 0 is legal (empty tensors with the right shapes/dtypes); a low threshold gives the DETECTIONS_PER_IMAGE cap."""
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    from oracle.ref_cpu import OracleModel
    img = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (100, 150, 3), dtype=np.uint8))
    for opts, expect in ((["MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.9999], "zero"),
                         (["MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.0, "TEST.DETECTIONS_PER_IMAGE", 20], "many")):
        cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS + ["MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION", 7] + opts)
        state = make_synthetic_state(cfg, 5)
        out = DensePosePredictor(cfg, state, dtype="fp32")(img)
        ref = OracleModel(cfg, state)(img)
        torch.cuda.synchronize()
        R = ref["scores"].shape[0]
        assert (R == 0) if expect == "zero" else (R == 20)
        for k in ref:
            assert tuple(out[k].shape) == tuple(ref[k].shape) and out[k].dtype == ref[k].dtype, (k, out[k].shape, ref[k].shape)
        if R:
            assert (out["pred_densepose_u"].cpu() - ref["pred_densepose_u"]).abs().max().item() <= IUV_ATOL
            # the side-stream decoder is pure scheduling: bit-identical results without it
            plain = DensePosePredictor(cfg, state, dtype="fp32")
            plain.engine.overlap_decoder = False
            out2 = plain(img)
            torch.cuda.synchronize()
            for k in out:
                assert torch.equal(out[k].cpu(), out2[k].cpu()), k


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_densepose_branch_slots_follow_the_detection_count(dtype):
    """The DensePose branch is launched before the host knows R. Its buffers are sized for a high-water mark of recent batches
    (Engine._dp_slots), not for n x DETECTIONS_PER_IMAGE slots (the reference default is 100 per image: 3.9 MB of fp32 IUV maps
    per slot whatever the scene holds); the device caps the compact ROI list at the slot count (dp_count_offsets_limited) and a
    batch with more boxes than slots runs the branch once more. Checked: (1) with the default 100 slots per frame and a scene of
    few boxes the IUV buffers hold <= 48 slots, not 200; (2) the results equal a run sized for every slot, bit for bit;
    (3) a batch that overflows the mark (quiet frames first, then a crowded one) still returns every detection, identical to a
    fresh predictor's result."""
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    rng = np.random.default_rng(21)
    imgs = [torch.from_numpy(rng.integers(0, 256, (100, 150, 3), dtype=np.uint8)) for _ in range(2)]
    base = TINY_OPTS + ["MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION", 7, "TEST.DETECTIONS_PER_IMAGE", 100]
    # a scene of few boxes under the default slot count: six proposals per frame survive the RPN, every one of them is kept
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", base + ["MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.0, "MODEL.RPN.POST_NMS_TOPK_TEST", 6])
    state = make_synthetic_state(cfg, 5)
    pred = DensePosePredictor(cfg, state, dtype=dtype)
    seen = []
    branch0 = pred.engine.densepose_branch

    def branch(*a, slots=None, **kw):
        seen.append(slots)
        return branch0(*a, slots=slots, **kw)
    pred.engine.densepose_branch = branch
    out = pred.predict_batch(imgs)
    torch.cuda.synchronize()
    R = sum(int(o["scores"].shape[0]) for o in out)
    assert 0 < R <= 32 and seen == [32]                     # 16 per frame until a count has been seen, never 2 x 100
    assert out[0]["pred_densepose_u"]._base is None or out[0]["pred_densepose_u"]._base.shape[0] <= 48
    full = DensePosePredictor(cfg, state, dtype=dtype)
    full.engine._dp_slots = lambda n, D, seen=None: None if seen is not None else n * D
    want = full.predict_batch(imgs)
    torch.cuda.synchronize()
    for a, b in zip(out, want):
        for k in a:
            assert torch.equal(a[k].cpu(), b[k].cpu()), k
    # overflow: the same predictor now meets a scene with (many) more boxes than its mark allows for
    crowded = get_config("densepose_rcnn_R_50_FPN_s1x", base + ["MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.0])
    pred2 = DensePosePredictor(crowded, state, dtype=dtype)
    pred2.engine._r_hwm[(2, 100)] = 3.0                     # as if quiet batches had come before
    seen2 = []
    b0 = pred2.engine.densepose_branch
    pred2.engine.densepose_branch = lambda *a, slots=None, **kw: (seen2.append(slots), b0(*a, slots=slots, **kw))[1]
    got = pred2.predict_batch(imgs)
    torch.cuda.synchronize()
    fresh = DensePosePredictor(crowded, state, dtype=dtype)
    fresh.engine._dp_slots = lambda n, D, seen=None: None if seen is not None else n * D
    want = fresh.predict_batch(imgs)
    torch.cuda.synchronize()
    assert len(seen2) == 2 and seen2[0] == 16 and seen2[1] >= sum(int(o["scores"].shape[0]) for o in want) > 16
    for a, b in zip(got, want):
        for k in a:
            assert torch.equal(a[k].cpu(), b[k].cpu()), k


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_fused_resize_preprocess_equals_two_launches(dtype):
    """SURVEY 8 f1 for frames whose scale is not 1 (1080 x 1920 video frames -> 749 x 1333, BASELINE.json configs[4]'s input):
    the vertical resize pass, (x - mean) / std, the padding and the stem's paired layout in ONE launch (dp_resize_preprocess_u8_batch)
    against the resize's two passes followed by the preprocess launch - the paired-layout tensor the stem reads, bit for bit, and the
    end-to-end results of a batch with HIP graphs and pipeline lanes."""
    import ctypes as C
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    from densepose_torchscript_amd.resize import FusedResize, resize_u8_device_batch
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS + ["INPUT.MIN_SIZE_TEST", 800, "INPUT.MAX_SIZE_TEST", 1333,
                                                                 "MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION", 7])
    state = make_synthetic_state(cfg, 2)
    rng = np.random.default_rng(5)
    frames = [torch.from_numpy(rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)).cuda() for _ in range(3)]
    two = DensePosePredictor(cfg, state, dtype=dtype, resize="device")
    two.fuse_resize = False
    one = DensePosePredictor(cfg, state, dtype=dtype, resize="device", use_graphs=True, pipeline_depth=2)
    assert one.fuse_resize
    # the tensor the stem reads
    e = one.engine
    k = one._scale(1080, 1920)
    u8 = resize_u8_device_batch(e, frames, k, src_hwc=True)
    assert tuple(u8.shape) == (3, 3, 749, 1333)
    want_x = e.preprocess(u8, 768, 1344)
    fr = FusedResize(e, frames, k, src_hwc=True)
    got_x = torch.full_like(want_x.t, 3.0)
    fr.run(e, got_x)
    torch.cuda.synchronize()
    assert torch.equal(got_x, want_x.t)
    # end to end, three times over (capture, replay, replay on the other lane)
    want = two.predict_batch(frames)
    torch.cuda.synchronize()
    for _ in range(3):
        got = one.predict_batch(frames)
        one.join()
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            for key in a:
                assert torch.equal(a[key].cpu(), b[key].cpu()), key
    assert sum(int(o["scores"].shape[0]) for o in want) > 0


def test_video_frame_geometry_and_chw_input():
    """1080x1920 frame (-> 749x1333, padded 768x1344) given as CHW; same result as HWC."""
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS + ["MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION", 7,
                                                                "INPUT.MIN_SIZE_TEST", 800, "INPUT.MAX_SIZE_TEST", 1333])
    state = make_synthetic_state(cfg, 6)
    pred = DensePosePredictor(cfg, state, dtype="fp32")
    img = torch.from_numpy(np.random.default_rng(4).integers(0, 256, (1080, 1920, 3), dtype=np.uint8))
    a = pred(img)
    b = pred(img.permute(2, 0, 1).contiguous())
    assert a["image_size"].tolist() == [1080, 1920]
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    with pytest.raises(AssertionError):
        pred(torch.zeros((10, 10, 4), dtype=torch.uint8))


def test_gpu_iuv_extract_matches_reference_visualizer_golden():
    """densepose_torchscript_amd.visualizer.extract_iuv (dp_iuv_extract) vs the labels/uv the reference's
    DensePoseResultExtractor produced (visualizer.py:46-56), on the engine's own fp32 outputs."""
    from densepose_torchscript_amd.visualizer import extract_iuv, iuv_image
    meta, z, cfg, pred, out = _run("tiny_r50_s1x_a", "fp32")
    dev_out = {k: (v.cuda() if k != "image_size" else v) for k, v in out.items()}
    results, xywh = extract_iuv(dev_out)
    assert len(results) == z["out/scores"].shape[0]
    for i, r in enumerate(results):
        ref_l, ref_uv = z["vis/labels_%d" % i], z["vis/uv_%d" % i]
        lab = r["labels"].cpu().numpy()
        assert lab.shape == ref_l.shape
        # bit-exact part index: dp_iuv_extract restates the CPU bilinear kernel operation by operation, and the fp32 maps it
        # reads agree with the reference's to ~1e-5, far inside the margin between the two largest resampled logits here
        np.testing.assert_array_equal(lab, ref_l)
        np.testing.assert_allclose(r["uv"].cpu().numpy(), ref_uv, atol=IUV_ATOL)
    img = iuv_image(results, xywh, int(out["image_size"][0]), int(out["image_size"][1]))
    assert img.shape == (3, int(out["image_size"][0]), int(out["image_size"][1])) and img.dtype == np.uint8


@pytest.mark.parametrize("name,dtype", [("tiny_r50_s1x_a", "fp32"), ("tiny_r50_s1x_a", "bf16"), ("tiny_r50_dl", "bf16"), ("tiny_r101_dl", "fp16")])
def test_replica_built_from_broadcast_weights_equals_rank0(name, dtype):
    """N ranks x the same frames == 1 rank. Rank 0 packs the real weights; every other rank builds its engine from
    a zeros / ones state (bench.py) and receives `PackedModel.parameter_tensors()` through parallel.broadcast_tensors.
    Here the collective is replaced by a copy of rank 0's flat buckets (the flatten / unflatten code is the real one), and the
    replica's outputs must be bit-identical to rank 0's."""
    from densepose_torchscript_amd import parallel
    from densepose_torchscript_amd.predictor import DensePosePredictor
    from densepose_torchscript_amd.weights import param_shapes
    meta, z = load_golden(name)
    cfg, state, img = golden_case_inputs(meta)
    rank0 = DensePosePredictor(cfg, state, dtype=dtype)
    blank = {k: np.zeros(s, dtype=np.float32) for k, s in param_shapes(cfg).items()}
    for k in blank:
        if k.endswith("running_var"):
            blank[k] += 1.0
    replica = DensePosePredictor(cfg, blank, dtype=dtype)
    src, dst = rank0.engine.model.parameter_tensors(), replica.engine.model.parameter_tensors()
    assert [(t.shape, t.dtype) for t in src] == [(t.shape, t.dtype) for t in dst]
    assert any(not torch.equal(a, b) for a, b in zip(src, dst))       # the replica really starts from different weights
    bucket = 1 << 16                                                     # small buckets: many messages, tensors > one bucket too
    wire = [torch.cat([b.reshape(-1) for b in bk]) for bk in parallel.plan_buckets(src, bucket)]
    parallel.broadcast_tensors(dst, src=0, bucket_bytes=bucket, transport=lambda flat: flat.copy_(wire.pop(0)))
    assert not wire
    for a, b in zip(src, dst):
        assert torch.equal(a, b)
    rng = np.random.default_rng(3)
    frames = [torch.from_numpy(img)] + [torch.from_numpy(rng.integers(0, 256, img.shape, dtype=np.uint8)) for _ in range(2)]
    want, got = rank0.predict_batch(frames), replica.predict_batch(frames)
    torch.cuda.synchronize()
    assert sum(int(w["scores"].shape[0]) for w in want) > 0
    for w, g in zip(want, got):
        for k in w:
            assert torch.equal(w[k].cpu(), g[k].cpu()), (name, dtype, k)


def test_host_resident_frames_equal_device_resident_frames():
    """Frames handed over in pageable HOST memory (the reference's boundary: defaults.py:65-80, run.py:34-36) go through the
    pinned staging ring + copy stream (predictor._HostFrameRing); the results are bit-identical to the same frames already
    resident on the device, ring slots are reused safely across calls (more calls than slots, pipeline lanes on)."""
    from densepose_torchscript_amd.predictor import DensePosePredictor
    meta, z = load_golden("tiny_r50_s1x_a")
    cfg, state, img = golden_case_inputs(meta)
    rng = np.random.default_rng(17)
    batches = [[torch.from_numpy(rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)) for _ in range(3)] for _ in range(5)]
    for mode in ("device", "host"):
        dev = DensePosePredictor(cfg, state, dtype="bf16", resize="device" if mode == "device" else "host")
        want = [dev.predict_batch([f.cuda() if mode == "device" else f for f in b]) for b in batches]
        torch.cuda.synchronize()
        want = [[{k: v.cpu() for k, v in r.items()} for r in res] for res in want]
        host = DensePosePredictor(cfg, state, dtype="bf16", resize=mode, use_graphs=(mode == "device"), pipeline_depth=2)
        got = []
        for rep in range(2):                       # second pass: every ring slot is being reused
            got = [host.predict_batch(b) for b in batches]
        host.join()
        torch.cuda.synchronize()
        assert host._host_ring is not None and len(host._host_ring.rings) >= 1
        for res_w, res_g in zip(want, got):
            for w, g in zip(res_w, res_g):
                for k in w:
                    assert torch.equal(w[k], g[k].cpu()), (mode, k)
    # CHW views and pinned inputs take the same path
    chw = [b.permute(2, 0, 1).contiguous().pin_memory() for b in batches[0]]
    got = host.predict_batch(chw)
    torch.cuda.synchronize()
    for w, g in zip(want[0], got):
        for k in w:
            assert torch.equal(w[k], g[k].cpu()), k


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_fused_launches_equal_layer_by_layer_end_to_end(dtype, policy):
    """The fused kernels only exist for 16-bit storage, so the fp32 goldens never run them: end to end in the throughput dtypes
    the default engine (stem + pool fused, res2 tails fused, RPN heads in the conv epilogue, weight-stationary 3x3) must equal,
    bit for bit, an engine with every fusion off and the 3x3 layers on the ring kernels."""
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 400, "TEST.DETECTIONS_PER_IMAGE", 5])
    state = make_synthetic_state(cfg, 4)
    rng = np.random.default_rng(23)
    imgs = [torch.from_numpy(rng.integers(0, 256, (256, 400, 3), dtype=np.uint8)).cuda() for _ in range(3)]
    fused = DensePosePredictor(cfg, state, dtype=dtype, resize="device")
    want = fused.predict_batch(imgs)
    torch.cuda.synchronize()
    want = [{k: v.cpu() for k, v in r.items()} for r in want]
    policy.set("conv_ws", "0")
    plain = DensePosePredictor(cfg, state, dtype=dtype, resize="device")
    plain.engine.fuse_stem_pool = plain.engine.fuse_bottleneck = plain.engine.fuse_rpn_head = False
    got = plain.predict_batch(imgs)
    torch.cuda.synchronize()
    assert sum(int(w["scores"].shape[0]) for w in want) > 0
    for w, g in zip(want, got):
        for k in w:
            assert torch.equal(w[k], g[k].cpu()), (dtype, k)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_stream_schedule_does_not_change_the_results(dtype):
    """Where the decoder's side stream forks (after the RPN heads, at the FPN's end, or not at all), with graphs and pipeline
    lanes or launch by launch on one stream: the same kernels on the same data, so the same bits."""
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 400, "TEST.DETECTIONS_PER_IMAGE", 5])
    state = make_synthetic_state(cfg, 4)
    rng = np.random.default_rng(29)
    batches = [[torch.from_numpy(rng.integers(0, 256, (256, 400, 3), dtype=np.uint8)).cuda() for _ in range(3)] for _ in range(3)]

    def run(after_heads, overlap, graphs, fork=0):
        from densepose_torchscript_amd.options import EngineOptions
        pred = DensePosePredictor(cfg, state, dtype=dtype, resize="device", use_graphs=graphs, pipeline_depth=2 if graphs else 1,
                                  options=EngineOptions(fork_levels=fork))
        pred.engine.decoder_after_rpn_heads, pred.engine.overlap_decoder = after_heads, overlap
        outs = [pred.predict_batch(b) for b in batches] + [pred.predict_batch(batches[0])]    # the 4th replays lane 0's graph
        pred.join()
        torch.cuda.synchronize()
        return [[{k: v.cpu() for k, v in r.items()} for r in o] for o in outs]

    want = run(False, False, False)
    assert sum(int(r["scores"].shape[0]) for o in want for r in o) > 0
    # (fork = EngineOptions.fork_levels: FPN output convolutions / RPN levels on forked streams - the default until round 6 was 2)
    for after_heads, overlap, graphs, fork in ((True, True, True, 0), (False, True, True, 0), (True, True, False, 0), (True, True, True, 3), (True, True, False, 2)):
        got = run(after_heads, overlap, graphs, fork)
        for wo, go in zip(want, got):
            for w, g in zip(wo, go):
                for k in w:
                    assert torch.equal(w[k], g[k]), (after_heads, overlap, graphs, fork, k)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_frames_of_the_test_size_skip_the_resize(dtype):
    """A frame whose shortest edge already is MIN_SIZE_TEST has scale 1 (defaults.py:84-89): the uint8 resize is the identity and
    the engine reads the frames as handed over (interleaved HWC, dp_preprocess_params.src_hwc) - device frames, host frames,
    with and without graphs; the results equal the reference-exact host resize path bit for bit."""
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS + ["INPUT.MIN_SIZE_TEST", 96, "INPUT.MAX_SIZE_TEST", 160])
    state = make_synthetic_state(cfg, 5)
    rng = np.random.default_rng(31)
    imgs = [torch.from_numpy(rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)) for _ in range(4)]
    host = DensePosePredictor(cfg, state, dtype=dtype, resize="host")
    assert host._scale(96, 160) == 1.0
    want = [{k: v.cpu() for k, v in r.items()} for r in host.predict_batch(imgs)]
    for graphs in (False, True):
        dev = DensePosePredictor(cfg, state, dtype=dtype, resize="device", use_graphs=graphs, pipeline_depth=2 if graphs else 1)
        for frames in ([im.cuda() for im in imgs], imgs, [im.permute(2, 0, 1).contiguous().cuda() for im in imgs]):
            for rep in range(2):
                got = dev.predict_batch(frames)
            dev.join()
            torch.cuda.synchronize()
            for w_, g_ in zip(want, got):
                for k in w_:
                    assert torch.equal(w_[k], g_[k].cpu()), (dtype, graphs, k)


def test_batch_64_full_size_frames_equal_single_calls():
    """BASELINE.json configs[3] / [4] hand a GPU up to 64 frames: at 800x1333 the res2 / res3 / p2-level tensors of such a batch
    exceed 2 GiB, so the kernels that address tensors with 32-bit offsets (fused res2 tail, the two-source conv3 + shortcut, the
    decoder's post-activation sums) run image chunk by image chunk - a frame's result must still be the single-call result."""
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 4])
    state = make_synthetic_state(cfg, 0)
    rng = np.random.default_rng(77)
    base = [torch.from_numpy(rng.integers(0, 256, (800, 1333, 3), dtype=np.uint8)).cuda() for _ in range(4)]
    frames = [base[i % 4] for i in range(64)]
    pred = DensePosePredictor(cfg, state, dtype="bf16", resize="device")
    got = pred.predict_batch(frames)
    torch.cuda.synchronize()
    for i in (0, 1, 63):
        want = pred(frames[i])
        torch.cuda.synchronize()
        assert int(want["scores"].shape[0]) > 0
        for k in want:
            assert torch.equal(want[k].cpu(), got[i][k].cpu()), (i, k)
