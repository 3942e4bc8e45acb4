"""Checkpoint schema, loader and seeded synthetic weights.

* ``param_shapes(cfg)``    - canonical (alias-free) key -> shape table of the Detectron2-zoo
  ``.pkl`` files the reference loads (/root/reference/detectron2/checkpoint/detection_checkpoint.py:49-56;
  module layout from backbone/resnet.py:609-689, fpn.py:72-100, rpn.py:97-113, box_head.py:70-75,
  fast_rcnn.py:200-203, densepose roi_head.py:42-68, v1convx.py:36-41, deeplab.py:33-60,117-137, chart.py:45-59).
* ``load_checkpoint(path, cfg)`` - ``.pkl`` (pickle, latin1, ``{"model": {name: ndarray}, "__author__": ...}``)
  or a torch ``.pth`` state dict; accepts the TorchScript fork's ModuleList alias spellings
   and the ``model.`` prefix of ``DefaultPredictor.state_dict()``; Caffe2 / Detectron1 blob
  pickles are renamed by ``c2_names.py`` (c2_model_loading.py:66-204).
* ``make_synthetic_state(cfg, seed)`` - seeded weights whose activations stay O(1) on 0..255 pixel
  input (there are no pretrained checkpoints and no network in the build/bench boxes).
"""
import pickle
import re
from collections import OrderedDict

import numpy as np


def _conv(shapes, name, cout, cin, k, bias):
    shapes[name + ".weight"] = (cout, cin, k, k)
    if bias:
        shapes[name + ".bias"] = (cout,)


def _bn(shapes, name, c):
    for s in ("weight", "bias", "running_mean", "running_var"):
        shapes[name + "." + s] = (c,)


def resnet_blocks(cfg):
    """Yield (stage 'res2'.., block idx, cin, cmid, cout, stride, has_shortcut)."""
    cin = cfg.stem_out
    cout = cfg.res2_out
    cmid = cfg.width_per_group
    out = []
    for si, nb in enumerate(cfg.blocks_per_stage):
        for b in range(nb):
            stride = 2 if (b == 0 and si > 0) else 1
            out.append(("res%d" % (si + 2), b, cin, cmid, cout, stride, cin != cout))
            cin = cout
        cout *= 2
        cmid *= 2
    return out


def decoder_layout(cfg):
    """[(level name, n convs)] - head_length = max(1, log2(stride) - log2(4)) (roi_head.py:45-47)."""
    return [("p2", 1), ("p3", 1), ("p4", 2), ("p5", 3)]


def param_shapes(cfg):
    s = OrderedDict()
    bu = "backbone.bottom_up."
    _conv(s, bu + "stem.conv1", cfg.stem_out, 3, 7, False)
    _bn(s, bu + "stem.conv1.norm", cfg.stem_out)
    for stage, b, cin, cmid, cout, stride, sc in resnet_blocks(cfg):
        p = "%s%s.%d." % (bu, stage, b)
        if sc:
            _conv(s, p + "shortcut", cout, cin, 1, False)
            _bn(s, p + "shortcut.norm", cout)
        _conv(s, p + "conv1", cmid, cin, 1, False)
        _bn(s, p + "conv1.norm", cmid)
        _conv(s, p + "conv2", cmid, cmid, 3, False)
        _bn(s, p + "conv2.norm", cmid)
        _conv(s, p + "conv3", cout, cmid, 1, False)
        _bn(s, p + "conv3.norm", cout)
    c = cfg.res2_out
    for lvl in (2, 3, 4, 5):
        _conv(s, "backbone.fpn_lateral%d" % lvl, cfg.fpn_out, c, 1, True)
        _conv(s, "backbone.fpn_output%d" % lvl, cfg.fpn_out, cfg.fpn_out, 3, True)
        c *= 2
    A = len(cfg.anchor_ratios)
    _conv(s, "proposal_generator.rpn_head.conv", cfg.fpn_out, cfg.fpn_out, 3, True)
    _conv(s, "proposal_generator.rpn_head.objectness_logits", A, cfg.fpn_out, 1, True)
    _conv(s, "proposal_generator.rpn_head.anchor_deltas", 4 * A, cfg.fpn_out, 1, True)
    fin = cfg.fpn_out * cfg.box_pool * cfg.box_pool
    for i in range(cfg.box_num_fc):
        s["roi_heads.box_head.fc%d.weight" % (i + 1)] = (cfg.box_fc_dim, fin)
        s["roi_heads.box_head.fc%d.bias" % (i + 1)] = (cfg.box_fc_dim,)
        fin = cfg.box_fc_dim
    s["roi_heads.box_predictor.cls_score.weight"] = (2, fin)
    s["roi_heads.box_predictor.cls_score.bias"] = (2,)
    s["roi_heads.box_predictor.bbox_pred.weight"] = (4, fin)
    s["roi_heads.box_predictor.bbox_pred.bias"] = (4,)
    if cfg.dp_decoder_on:
        for lvl, n in decoder_layout(cfg):
            for k in range(n):
                _conv(s, "roi_heads.decoder.%s.%d" % (lvl, 2 * k), cfg.dp_decoder_dims,
                      cfg.fpn_out if k == 0 else cfg.dp_decoder_dims, 3, True)
        _conv(s, "roi_heads.decoder.predictor", cfg.dp_decoder_classes, cfg.dp_decoder_dims, 1, True)
    # the DensePose head always takes FPN_OUT channels (roi_head.py:107: channels of in_features[0])
    cin = cfg.fpn_out
    hd = "roi_heads.densepose_head."
    if cfg.is_deeplab:
        a = hd + "ASPP."
        _conv(s, a + "convs.0.0", cin, cin, 1, False)
        s[a + "convs.0.1.weight"] = (cin,)
        s[a + "convs.0.1.bias"] = (cin,)
        for i in (1, 2, 3):
            _conv(s, a + "convs.%d.0" % i, cin, cin, 3, False)
            s[a + "convs.%d.1.weight" % i] = (cin,)
            s[a + "convs.%d.1.bias" % i] = (cin,)
        _conv(s, a + "convs.4.1", cin, cin, 1, False)
        s[a + "convs.4.2.weight"] = (cin,)
        s[a + "convs.4.2.bias"] = (cin,)
        _conv(s, a + "project.0", cin, 5 * cin, 1, False)
    c = cin
    for i in range(cfg.dp_num_convs):
        n = hd + "body_conv_fcn%d" % (i + 1)
        if cfg.is_deeplab:
            _conv(s, n, cfg.dp_head_dim, c, 3, False)
            s[n + ".norm.weight"] = (cfg.dp_head_dim,)
            s[n + ".norm.bias"] = (cfg.dp_head_dim,)
        else:
            _conv(s, n, cfg.dp_head_dim, c, 3, True)
        c = cfg.dp_head_dim
    pr = "roi_heads.densepose_predictor."
    for nm, co in (("ann_index_lowres", cfg.dp_coarse_ch), ("index_uv_lowres", cfg.dp_patches + 1),
                   ("u_lowres", cfg.dp_patches + 1), ("v_lowres", cfg.dp_patches + 1)):
        s[pr + nm + ".weight"] = (cfg.dp_head_dim, co, 4, 4)  # ConvTranspose2d: [Cin, Cout, kh, kw]
        s[pr + nm + ".bias"] = (co,)
    return s


# ---- alias spellings added by the TorchScript fork -> canonical names -------------------
_ALIAS_RULES = [
    (re.compile(r"^backbone\.bottom_up\.stages\.(\d+)\."), lambda m: "backbone.bottom_up.res%d." % (int(m.group(1)) + 2)),
    (re.compile(r"^backbone\.lateral_convs\.(\d+)\."), lambda m: "backbone.fpn_lateral%d." % (5 - int(m.group(1)))),
    (re.compile(r"^backbone\.output_convs\.(\d+)\."), lambda m: "backbone.fpn_output%d." % (5 - int(m.group(1)))),
    (re.compile(r"^roi_heads\.decoder\.scale_heads\.(\d+)\."), lambda m: "roi_heads.decoder.p%d." % (int(m.group(1)) + 2)),
    (re.compile(r"^roi_heads\.densepose_head\.stacked_convs\.(\d+)\."),
     lambda m: "roi_heads.densepose_head.body_conv_fcn%d." % (int(m.group(1)) + 1)),
]


def canonical_name(key):
    if key.startswith("model."):
        key = key[len("model."):]
    for rx, fn in _ALIAS_RULES:
        m = rx.match(key)
        if m:
            return fn(m) + key[m.end():]
    return key


def load_checkpoint(path, cfg=None):
    """-> OrderedDict canonical name -> float32 ndarray (pixel_mean/std and cell_anchors dropped).
    Caffe2 / Detectron1 pickles (no "model" + "__author__" pair: detection_checkpoint.py:53-64) need ``cfg`` to name
    the target variant; they are renamed by ``c2_names.convert_caffe2_blobs``."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        if "model" in data and "__author__" in data:
            data = data["model"]
        elif "model" in data and isinstance(data["model"], dict):
            data = data["model"]
        else:
            from .c2_names import convert_caffe2_blobs
            if cfg is None:
                raise ValueError("%s is a Caffe2/Detectron1 checkpoint: pass the model config so that its blobs can be renamed" % path)
            blobs = data["blobs"] if "blobs" in data else data
            blobs = {k: v for k, v in blobs.items() if not k.endswith("_momentum")}
            return convert_caffe2_blobs(blobs, param_shapes(cfg))
    else:
        import torch
        data = torch.load(path, map_location="cpu")
        if "model" in data and not hasattr(data["model"], "shape"):
            data = data["model"]
    out = OrderedDict()
    for k, v in data.items():
        ck = canonical_name(k)
        if ck in ("pixel_mean", "pixel_std") or "anchor_generator.cell_anchors" in ck:
            continue
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v, dtype=np.float32)
        if ck in out:
            if out[ck].shape != v.shape or not np.array_equal(out[ck], v):
                raise ValueError("alias %s disagrees with %s" % (k, ck))
            continue
        out[ck] = v
    return out


def check_state(cfg, state):
    """Strict check against the schema (the reference loads non-strictly and silently keeps
    random init for missing keys - detection_checkpoint.py:95-122; an inference engine must not)."""
    want = param_shapes(cfg)
    missing = [k for k in want if k not in state]
    bad = [(k, tuple(state[k].shape), want[k]) for k in want if k in state and tuple(state[k].shape) != tuple(want[k])]
    if missing or bad:
        raise ValueError("checkpoint does not match %s: missing=%s shape_mismatch=%s"
                         % (cfg.name, missing[:8], bad[:8]))
    return [k for k in state if k not in want]


def save_pkl(state, path, author="densepose_torchscript_amd synthetic"):
    with open(path, "wb") as f:
        pickle.dump({"model": {k: np.asarray(v) for k, v in state.items()}, "__author__": author}, f, protocol=2)


def make_synthetic_state(cfg, seed=0):
    """Seeded synthetic weights (numpy Generator PCG64 -> identical on every box).

    Scaling rules: convs followed by ReLU use He fan-in (std = sqrt(2/fan_in)), linear-output convs
    use std = sqrt(1/fan_in); the stem is divided by the std of mean-subtracted 0..255 pixels (~74);
    FrozenBN has mean 0 / var 1 / beta small, gamma 1 except the last BN of each bottleneck (0.25) so the
    residual trunk does not blow up; score/delta/IUV predictors are scaled so logits are O(1).
    """
    rng = np.random.default_rng(seed)
    shapes = param_shapes(cfg)
    st = OrderedDict()

    def normal(shape, std):
        return (rng.standard_normal(size=shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    for name, shp in shapes.items():
        leaf = name.rsplit(".", 1)[1]
        if ".norm." in name or re.search(r"ASPP\.convs\.\d\.[12]\.(weight|bias)$", name) and len(shp) == 1:
            c = shp[0]
            if leaf == "weight":
                g = 0.25 if ".conv3.norm." in name else 1.0
                st[name] = (np.float32(g) * (1.0 + 0.1 * rng.standard_normal(c, dtype=np.float32))).astype(np.float32)
            elif leaf == "bias":
                st[name] = normal((c,), 0.05)
            elif leaf == "running_mean":
                st[name] = normal((c,), 0.05)
            else:  # running_var
                st[name] = (1.0 + 0.2 * rng.random(c, dtype=np.float32)).astype(np.float32)
            continue
        if leaf == "bias":
            st[name] = normal(shp, 0.02)
            continue
        # weights
        if "densepose_predictor" in name:
            cin = shp[0]
            st[name] = normal(shp, (1.0 / (cin * 4.0)) ** 0.5)  # 2x2 taps contribute per output pixel
            continue
        fan_in = int(np.prod(shp[1:]))
        relu_after = True
        if any(t in name for t in ("fpn_lateral", "fpn_output", "objectness_logits", "anchor_deltas",
                                   "cls_score", "bbox_pred", "decoder.predictor", ".conv3.weight", ".shortcut.weight")):
            relu_after = False
        std = (2.0 / fan_in) ** 0.5 if relu_after else (1.0 / fan_in) ** 0.5
        if name.endswith("stem.conv1.weight"):
            std /= 74.0
        if "anchor_deltas" in name:
            std *= 0.3
        if "bbox_pred" in name:
            std *= 0.5
        if "cls_score" in name:
            std *= 1.0
        st[name] = normal(shp, std)
    return st


def state_checksum(state):
    """Order-independent fingerprint used by the golden fixtures to prove both sides generated equal weights."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k], dtype=np.float32).tobytes())
    return h.hexdigest()
