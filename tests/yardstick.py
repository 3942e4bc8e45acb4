"""Distances of a low-precision run to the reference's fp32 golden of the same case - one definition for the reference's own low-precision
runs (tests/golden/<case>__bf16.npz / __half.npz, recorded by oracle/make_goldens.py --bf16 / --half) and for the engine's 16-bit modes."""
import numpy as np

IUV_KEYS = ("pred_densepose_coarse_segm", "pred_densepose_fine_segm", "pred_densepose_u", "pred_densepose_v")
BOX_TOL = 1.5      # a detection of the fp32 run counts as found when the nearest box of the other run is within this many pixels


def reference_lowp_distance(name, mode):
    """The REFERENCE run in `mode` ("bf16": predictor.bfloat16(), "half": predictor.half(), run.py:20-29 / export.py:36-37) against its own
    fp32 run, from the committed fixtures alone -> dict(box_match_rate, max_box_err_px, max_score_err, iuv (largest deviation on the matched
    detections relative to the largest fp32 value of the same map), label_agreement, label_agreement_foreground, label_pixels)."""
    from conftest import load_golden
    _, z = load_golden(name)
    _, zl = load_golden(name + "__" + mode)
    return lowp_distance_from_fixtures(z, zl)


def lowp_distance_from_fixtures(z, zl):
    """... on the loaded fixtures: z = the fp32 golden, zl = the low-precision one (bench.py calls this too)."""
    match = zl["match/fp32_to_this"]
    R = len(z["out/scores"])
    assert len(match) == R
    hits, box_err, score_err, iuv = 0, 0.0, 0.0, 0.0
    eq = tot = eq_fg = tot_fg = 0
    for i in range(R):
        j = int(match[i])
        if j < 0:
            continue
        hits += 1
        box_err = max(box_err, float(np.abs(zl["out/pred_boxes"][j] - z["out/pred_boxes"][i]).max()))
        score_err = max(score_err, float(abs(zl["out/scores"][j] - z["out/scores"][i])))
        for k in IUV_KEYS:
            ref = z["out/" + k][i]
            iuv = max(iuv, float(np.abs(zl["out/" + k][j] - ref).max()) / max(float(np.abs(ref).max()), 1e-6))
        want, got = z["vis/labels_%d" % i], zl["vis/labels_on_fp32_box_%d" % i]
        assert want.shape == got.shape
        eq, tot = eq + int((want == got).sum()), tot + want.size
        fg = want > 0
        eq_fg, tot_fg = eq_fg + int((want[fg] == got[fg]).sum()), tot_fg + int(fg.sum())
    return dict(detections=int(len(zl["out/scores"])), ref_detections=R, box_match_rate=hits / max(R, 1), max_box_err_px=box_err,
                max_score_err=score_err, iuv=iuv, label_agreement=eq / max(tot, 1), label_agreement_foreground=eq_fg / max(tot_fg, 1),
                label_pixels=tot)


def engine_distance(out, z, s, extract_iuv):
    """The same quantities for an engine output dict `out` (CPU tensors, full-resolution maps) against the fp32 golden `z` whose maps are
    subsampled by `s`; extract_iuv = oracle.ref_cpu.extract_iuv (the visualiser's label rule, visualizer.py:10-17) run on THIS run's maps
    over the fp32 golden's box of each matched detection."""
    import torch
    gb, gs, rb, rs = out["pred_boxes"].numpy(), out["scores"].numpy(), z["out/pred_boxes"], z["out/scores"]
    R = len(rs)
    hits, box_err, score_err, iuv = 0, 0.0, 0.0, 0.0
    eq = tot = eq_fg = tot_fg = 0
    for i in range(R):
        if len(gb) == 0:
            break
        d = np.abs(gb - rb[i]).max(axis=1)
        j = int(d.argmin())
        if d[j] >= BOX_TOL:
            continue
        hits += 1
        box_err, score_err = max(box_err, float(d[j])), max(score_err, float(abs(gs[j] - rs[i])))
        for k in IUV_KEYS:
            ref = z["out/" + k][i]
            iuv = max(iuv, float(np.abs(out[k][j].numpy()[:, ::s, ::s] - ref).max()) / max(float(np.abs(ref).max()), 1e-6))
        sub = {k: out[k][j:j + 1] for k in IUV_KEYS}
        sub["pred_boxes"] = torch.from_numpy(rb[i:i + 1].copy())
        (labels, _), = extract_iuv(sub)
        want, got = z["vis/labels_%d" % i], labels.numpy().astype(np.uint8)
        eq, tot = eq + int((want == got).sum()), tot + want.size
        fg = want > 0
        eq_fg, tot_fg = eq_fg + int((want[fg] == got[fg]).sum()), tot_fg + int(fg.sum())
    return dict(detections=int(len(gs)), ref_detections=R, box_match_rate=hits / max(R, 1), max_box_err_px=box_err, max_score_err=score_err,
                iuv=iuv, label_agreement=eq / max(tot, 1), label_agreement_foreground=eq_fg / max(tot_fg, 1), label_pixels=tot)
