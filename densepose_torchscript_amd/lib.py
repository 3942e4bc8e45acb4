"""ctypes binding of the C-ABI library (include/densepose_hip.h -> libdensepose_hip.so).

The product path has NO fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DP_HIP_LIB") or os.path.join(_HERE, "libdensepose_hip.so")  # DP_HIP_LIB: A/B builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

DP_F32, DP_BF16, DP_F16 = 0, 1, 2
ABI_VERSION = 8   # == DP_ABI_VERSION of include/densepose_hip.h (bumped whenever a params struct or the symbol set changes)

# user-facing dtype names -> (enum, element size)
DTYPES = {"fp32": DP_F32, "float32": DP_F32, "bf16": DP_BF16, "bfloat16": DP_BF16, "fp16": DP_F16, "float16": DP_F16, "half": DP_F16}

c_void_p, c_int, c_i32, c_i64, c_float = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float


class PreprocessParams(C.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p),
                ("n_img", c_i32), ("h", c_i32), ("w", c_i32), ("Hp", c_i32), ("Wp", c_i32), ("dtype", c_i32),
                ("mean", c_float * 3), ("inv_std_unused", c_float * 3), ("std", c_float * 3), ("paired", c_i32), ("src_hwc", c_i32)]


class ConvParams(C.Structure):
    _fields_ = [("in_", c_void_p), ("weight", c_void_p), ("ktab", c_void_p), ("bias", c_void_p),
                ("residual", c_void_p), ("out", c_void_p),
                ("N", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32),
                ("Ho", c_i32), ("Wo", c_i32), ("Cout", c_i32),
                ("Cout_w", c_i32), ("Kpad", c_i32),
                ("stride", c_i32), ("ntaps", c_i32),
                ("osN", c_i64), ("osH", c_i64), ("osW", c_i64),
                ("rsN", c_i64), ("rsH", c_i64), ("rsW", c_i64),
                ("rshift", c_i32), ("relu", c_i32), ("dtype", c_i32), ("out_f32", c_i32),
                ("hi_off", c_i32), ("wi_off", c_i32), ("stride_w", c_i32),
                ("head_w", c_void_p), ("head_b", c_void_p), ("head_out", c_void_p),
                ("shared_chip", c_i32), ("post_mode", c_i32), ("post_res", c_void_p), ("n_dev", c_void_p),
                ("in2", c_void_p), ("H2", c_i32), ("W2", c_i32), ("Cin2", c_i32), ("stride2", c_i32),
                ("split_k", c_i32), ("split_ws", c_void_p),
                ("n_groups", c_i32), ("weight_g", c_void_p * 4), ("ktab_g", c_void_p * 4), ("out_g", c_void_p * 4),
                ("ring_order", c_i32)]


class BottleneckParams(C.Structure):
    _fields_ = [("t1", c_void_p), ("residual", c_void_p), ("out", c_void_p), ("next_t1", c_void_p),
                ("w2", c_void_p), ("w3", c_void_p), ("w1n", c_void_p), ("ktab2", c_void_p),
                ("b2", c_void_p), ("b3", c_void_p), ("b1n", c_void_p),
                ("N", c_i32), ("H", c_i32), ("W", c_i32),
                ("Cmid", c_i32), ("Cout", c_i32), ("Cmid_next", c_i32),
                ("Kpad2", c_i32), ("Kpad3", c_i32), ("Kpad1n", c_i32),
                ("ntaps2", c_i32), ("hi_off2", c_i32), ("wi_off2", c_i32), ("k_order2", c_i32),
                ("dtype", c_i32), ("Csc", c_i32), ("sc_in", c_void_p)]


class PairParams(C.Structure):
    _fields_ = [("t2", c_void_p), ("residual", c_void_p), ("out", c_void_p), ("next_t1", c_void_p),
                ("w3", c_void_p), ("w1n", c_void_p), ("b3", c_void_p), ("b1n", c_void_p),
                ("M", c_i64), ("Cmid", c_i32), ("Cout", c_i32), ("Cmid_next", c_i32), ("Kpad3", c_i32), ("Kpad1n", c_i32), ("dtype", c_i32)]


class StemPoolParams(C.Structure):
    _fields_ = [("in_", c_void_p), ("weight", c_void_p), ("bias", c_void_p), ("out", c_void_p),
                ("N", c_i32), ("Hp", c_i32), ("Wp", c_i32), ("Cout", c_i32), ("Kpad", c_i32), ("dtype", c_i32)]


class RpnLevelParams(C.Structure):
    _fields_ = [("head", c_void_p),
                ("n_img", c_i32), ("Hi", c_i32), ("Wi", c_i32), ("A", c_i32), ("head_c", c_i32),
                ("stride_px", c_i32),
                ("cell_anchors", (c_float * 4) * 3),
                ("level", c_i32), ("kmax", c_i32), ("slot_off", c_i32), ("slots_per_img", c_i32),
                ("clip_x", c_float), ("clip_y", c_float),
                ("cand_boxes", c_void_p), ("cand_scores", c_void_p), ("cand_level", c_void_p), ("cand_valid", c_void_p),
                ("workspace", c_void_p)]


class NmsParams(C.Structure):
    _fields_ = [("boxes", c_void_p), ("scores", c_void_p), ("group", c_void_p), ("valid", c_void_p),
                ("n_img", c_i32), ("n_slots", c_i32), ("iou_thr", c_float), ("max_out", c_i32), ("trick_max_numel", c_i32),
                ("out_boxes", c_void_p), ("out_scores", c_void_p), ("out_index", c_void_p), ("out_count", c_void_p),
                ("workspace", c_void_p)]


class RoiAlignParams(C.Structure):
    _fields_ = [("feat", c_void_p * 4), ("Hl", c_i32 * 4), ("Wl", c_i32 * 4), ("scale", c_float * 4),
                ("n_levels", c_i32), ("min_level", c_i32), ("C", c_i32), ("P", c_i32), ("sampling", c_i32),
                ("boxes", c_void_p), ("counts", c_void_p), ("n_img", c_i32), ("max_rois", c_i32),
                ("out", c_void_p), ("dtype", c_i32), ("compact", c_i32), ("roi_offsets", c_void_p)]


class BoxDecodeParams(C.Structure):
    _fields_ = [("logits", c_void_p), ("ld", c_i32), ("prop_boxes", c_void_p), ("prop_counts", c_void_p),
                ("n_img", c_i32), ("max_rois", c_i32),
                ("wx", c_float), ("wy", c_float), ("ww", c_float), ("wh", c_float), ("score_thresh", c_float),
                ("cand_boxes", c_void_p), ("cand_scores", c_void_p), ("cand_group", c_void_p), ("cand_valid", c_void_p)]


class PostprocessParams(C.Structure):
    _fields_ = [("boxes", c_void_p), ("counts", c_void_p), ("n_img", c_i32), ("max_dets", c_i32),
                ("scale_xy", c_void_p), ("out_hw", c_void_p), ("out_boxes", c_void_p), ("keep", c_void_p)]


class IuvParams(C.Structure):
    _fields_ = [("in_", c_void_p), ("R", c_i32), ("Hs", c_i32), ("Ws", c_i32), ("in_c", c_i32),
                ("n_coarse", c_i32), ("n_fine", c_i32),
                ("coarse", c_void_p), ("fine", c_void_p), ("u", c_void_p), ("v", c_void_p), ("r_dev", c_void_p)]


class GroupNormParams(C.Structure):
    _fields_ = [("x", c_void_p), ("R", c_i32), ("HW", c_i32), ("C", c_i32), ("c_stride", c_i32), ("c_off", c_i32),
                ("groups", c_i32), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("relu", c_i32), ("dtype", c_i32),
                ("r_dev", c_void_p)]


class ResizeParams(C.Structure):
    _fields_ = [("src", c_void_p), ("tmp", c_void_p), ("dst", c_void_p),
                ("H", c_i32), ("W", c_i32), ("oh", c_i32), ("ow", c_i32), ("src_hwc", c_i32),
                ("xtab", c_void_p), ("ytab", c_void_p), ("xprec", c_i32), ("yprec", c_i32)]


class IuvExtractParams(C.Structure):
    _fields_ = [("coarse", c_void_p), ("fine", c_void_p), ("u", c_void_p), ("v", c_void_p),
                ("R", c_i32), ("S", c_i32), ("n_coarse", c_i32), ("n_fine", c_i32),
                ("box_xywh", c_void_p), ("out_offset", c_void_p), ("labels", c_void_p), ("uv", c_void_p), ("max_hw", c_i32)]


class PackParams(C.Structure):
    _fields_ = [("Cout", c_i32), ("ntaps", c_i32), ("Cin", c_i32), ("cin_alloc", c_i32), ("dtype", c_i32), ("tap_major", c_i32)]


class PackInfo(C.Structure):
    _fields_ = [("cout", c_i32), ("cout_w", c_i32), ("kpad", c_i32), ("n_ktab", c_i32), ("plane_major", c_i32)]


# every symbol include/densepose_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dp_abi_version": (c_int, []),
    "dp_last_error": (C.c_char_p, []),
    "dp_preprocess_u8": (c_int, [C.POINTER(PreprocessParams), c_void_p]),
    "dp_conv2d_nhwc": (c_int, [C.POINTER(ConvParams), c_void_p]),
    "dp_conv2d_kernel_class": (c_int, [C.POINTER(ConvParams)]),
    "dp_conv2d_tile_rows": (c_int, [C.POINTER(ConvParams)]),
    "dp_bottleneck_tail_supported": (c_int, [C.POINTER(BottleneckParams)]),
    "dp_bottleneck_tail_nhwc": (c_int, [C.POINTER(BottleneckParams), c_void_p]),
    "dp_stem_pool_supported": (c_int, [C.POINTER(StemPoolParams)]),
    "dp_stem_pool_nhwc": (c_int, [C.POINTER(StemPoolParams), c_void_p]),
    "dp_maxpool3x3s2_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dp_subsample2_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dp_upsample_bilinear2x_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dp_merge_upsample2x_nhwc": (c_int, [c_void_p, C.POINTER(c_void_p), c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dp_rpn_topk_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "dp_rpn_topk_decode": (c_int, [C.POINTER(RpnLevelParams), c_void_p]),
    "dp_rpn_topk_decode_levels": (c_int, [C.POINTER(RpnLevelParams), c_int, c_void_p]),
    "dp_nms_workspace_bytes": (c_i64, [c_int, c_int]),
    "dp_batched_nms": (c_int, [C.POINTER(NmsParams), c_void_p]),
    "dp_roi_align_nhwc": (c_int, [C.POINTER(RoiAlignParams), c_void_p]),
    "dp_box_decode_score": (c_int, [C.POINTER(BoxDecodeParams), c_void_p]),
    "dp_postprocess_boxes": (c_int, [C.POINTER(PostprocessParams), c_void_p]),
    "dp_iuv_upsample_split": (c_int, [C.POINTER(IuvParams), c_void_p]),
    "dp_groupnorm_relu_nhwc": (c_int, [C.POINTER(GroupNormParams), c_void_p]),
    "dp_global_avgpool_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dp_broadcast_hw_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dp_count_offsets": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dp_count_offsets_limited": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dp_fold_frozen_bn": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "dp_conv_taps": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dp_pack_conv_info": (c_int, [C.POINTER(PackParams), C.POINTER(PackInfo)]),
    "dp_pack_conv_weights": (c_int, [C.POINTER(PackParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dp_resize_u8_bilinear": (c_int, [C.POINTER(ResizeParams), c_void_p]),
    "dp_resize_u8_bilinear_batch": (c_int, [C.POINTER(ResizeParams), C.POINTER(c_void_p), c_int, c_void_p]),
    "dp_resize_preprocess_u8_batch": (c_int, [C.POINTER(ResizeParams), C.POINTER(c_void_p), c_int, C.POINTER(PreprocessParams), c_void_p]),
    "dp_preprocess_u8_frames": (c_int, [C.POINTER(PreprocessParams), C.POINTER(c_void_p), c_int, c_void_p]),
    "dp_iuv_extract": (c_int, [C.POINTER(IuvExtractParams), c_void_p]),
    "dp_bottleneck_pair_supported": (c_int, [C.POINTER(PairParams)]),
    "dp_bottleneck_pair_nhwc": (c_int, [C.POINTER(PairParams), c_void_p]),
    "dp_set_policy": (c_int, [C.c_char_p, c_i64]),
    "dp_get_policy": (c_int, [C.c_char_p, C.POINTER(c_i64)]),
    "dp_reset_policy": (None, []),
    "dp_policy_num_keys": (c_int, []),
    "dp_policy_key": (C.c_char_p, [c_int]),
}


class DensePoseHipError(RuntimeError):
    pass


def _source_digest():
    """sha256 over every source the library is built from (kernels, headers, Makefile), in a fixed order."""
    import hashlib
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile"]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "densepose_hip.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_library(force=False):
    """hipcc --offload-arch=gfx950 -shared (cross-compiles without a GPU). Built IN-TREE so it travels.
    Up to date = the digest of the sources equals the one recorded beside the .so when it was built (file times mean
    nothing on a fresh checkout or after the tree was copied to the GPU box)."""
    stamp = LIB_PATH + ".sha256"
    digest = _source_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB_PATH
    subprocess.check_call(["make", "-C", CSRC, "-B", "-j%d" % min(8, os.cpu_count() or 1)])
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return LIB_PATH


_lib = None


if __name__ == "__main__":   # `python lib.py stamp [LIB]`: record the source digest beside a library built by hand (make)
    import sys
    if sys.argv[1:2] == ["stamp"]:
        with open((sys.argv[2] if len(sys.argv) > 2 else LIB_PATH) + ".sha256", "w") as f:
            f.write(_source_digest() + "\n")


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DensePoseHipError(
            "HIP kernel library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback in the product path)" % LIB_PATH)
    # A library built from other sources than the ones in this tree may lay its params structs out differently while exporting
    # the same symbols (the version number only moves when somebody remembers to bump it): the digest recorded at build time
    # must match the sources this binding was written against. DP_SKIP_STAMP_CHECK=1 is for hand-made experiment builds.
    if os.environ.get("DP_SKIP_STAMP_CHECK", "0") != "1":
        stamp = LIB_PATH + ".sha256"
        have = open(stamp).read().strip() if os.path.exists(stamp) else None
        if have != _source_digest():
            raise DensePoseHipError(
                "%s was not built from the sources in this tree (digest %s, sources %s): rebuild it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`" % (LIB_PATH, have and have[:12], _source_digest()[:12]))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.dp_abi_version() != ABI_VERSION:
        raise DensePoseHipError("ABI version mismatch: library %d, binding %d" % (lib.dp_abi_version(), ABI_VERSION))
    _lib = lib
    _apply_env_policy(lib)
    return lib


def policy_keys():
    lib = load()
    return [lib.dp_policy_key(i).decode() for i in range(lib.dp_policy_num_keys())]


def set_policy(key, value):
    """Kernel-policy override (include/densepose_hip.h dp_set_policy): tests pin a kernel class, tools A/B a schedule."""
    check(load().dp_set_policy(key.encode(), int(value)), "dp_set_policy(%s)" % key)


def get_policy(key):
    v = c_i64(0)
    check(load().dp_get_policy(key.encode(), C.byref(v)), "dp_get_policy(%s)" % key)
    return v.value


def reset_policy():
    load().dp_reset_policy()


class policy:
    """with policy(conv_big=1, conv_tp=6): ...   - overrides for the duration of the block, previous values restored afterwards."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_policy(k)
            set_policy(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_policy(k, v)
        return False


def apply_env_policy():
    """tools/: os.environ changed at run time -> policy table (every key back to its default first). The library never looks itself."""
    lib = load()
    lib.dp_reset_policy()
    _apply_env_policy(lib)


def _apply_env_policy(lib):
    """The library itself never reads the environment. For the command-line tools (tools/*.sh A/B runs: `DP_CONV_WS=0 python bench.py`)
    the HOST side applies DP_<KEY> variables to the policy table once, when the library is loaded."""
    for i in range(lib.dp_policy_num_keys()):
        key = lib.dp_policy_key(i).decode()
        v = os.environ.get("DP_" + key.upper())
        if v is not None and v.strip() != "":
            if key == "tail_kernel" and not v.lstrip("-").isdigit():
                v = "1" if v.startswith("t") else "0"     # historical spelling: DP_TAIL_KERNEL=tile
            if lib.dp_set_policy(key.encode(), int(v)) != 0:
                raise DensePoseHipError("bad policy override DP_%s=%s" % (key.upper(), v))


def check(rc, what=""):
    if rc != 0:
        msg = load().dp_last_error()
        raise DensePoseHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
