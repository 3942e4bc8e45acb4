#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/sec of the DensePose hot path on R_50_FPN_s1x, 800x1333 frames,
batch 8 per GPU, bf16 operands / fp32 accumulate, detections pinned to R = 8 per image.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one pass of the whole path (device resize -> backbone -> RPN -> box head -> DensePose head -> IUV maps)
over one batch of synthetic frames that are already resident in HBM. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE = 2.5e15   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: "~2.5 PF dense")
PEAK_F32_MATRIX = 157.3e12
HBM_PEAK = 8.0e12


def make_frames(n, start, hw, device):
    h, w = hw
    return [torch.from_numpy(np.random.default_rng(1234 + start + i).integers(0, 256, (h, w, 3), dtype=np.uint8)).to(device)
            for i in range(n)]


def cpu_baseline(cfg, state, hw, budget_s):
    """The oracle (a CPU port of the reference path, oracle/ref_cpu.py) timed on this box's host cores on a bounded
    sample of the same workload (same weights, same frames, same R)."""
    from oracle.ref_cpu import OracleModel
    # torch's CPU convolutions stop scaling (and then collapse) beyond ~32 threads on this class of host
    # (measured on the 2x64-core EPYC GPU box: 3x3 conv 256->256 @200x336: 61 ms at 32 threads, 149 ms at 128, 595 ms at 256)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    model = OracleModel(cfg, state)
    frames = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)) for i in range(4)]
    t0 = time.time()
    model(frames[0])  # warm-up (also bounds the sample: if one frame is already slow, time fewer)
    warm = time.time() - t0
    n_timed = max(1, min(3, int(budget_s / max(warm, 1e-3))))
    times = []
    for i in range(n_timed):
        t0 = time.time()
        model(frames[1 + i % 3])
        times.append(time.time() - t0)
    return {"value": round(len(times) / sum(times), 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d frame(s) 800x1333 R_50_FPN_s1x fp32 after 1 warm-up, oracle/ref_cpu.py (torch %s CPU + C roi_align/nms), p50 %.0f ms/img"
                      % (len(times), torch.__version__, 1e3 * float(np.median(times)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--config", default="densepose_rcnn_R_50_FPN_s1x")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--dets", type=int, default=8, help="detections per image (R), pinned via TEST.DETECTIONS_PER_IMAGE")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--streams", type=int, default=1, help="sub-batches of a step run concurrently on this many HIP streams")
    ap.add_argument("--pipeline", type=int, default=2, help="stream lanes consecutive batches alternate between (1 = off)")
    ap.add_argument("--no-overlap", action="store_true", help="decoder in line instead of on a side stream")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-roofline", action="store_true", help="skip the event-instrumented roofline pass (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    args = ap.parse_args()

    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd import parallel
    from densepose_torchscript_amd.predictor import DensePosePredictor
    from densepose_torchscript_amd.weights import param_shapes
    import torch.distributed as dist

    rank, local_rank, world = parallel.init_distributed()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank

    cfg = get_config(args.config, ["TEST.DETECTIONS_PER_IMAGE", args.dets])
    # weights: generated on rank 0 only, then ONE coalesced RCCL broadcast of the packed device tensors (xGMI)
    if rank == 0:
        state = make_synthetic_state(cfg, 0)
    else:
        state = {k: np.zeros(s, dtype=np.float32) for k, s in param_shapes(cfg).items()}
        for k in state:
            if k.endswith("running_var"):
                state[k] += 1.0
    pred = DensePosePredictor(cfg, state, dtype=args.dtype, device=device, resize="device", num_streams=args.streams, use_graphs=not args.no_graphs)
    if world > 1:
        parallel.broadcast_tensors(pred.engine.model.parameter_tensors(), src=0)
    eng = pred.engine
    # `--streams 1 --no-graphs` is the fully serialized configuration (one kernel at a time on the chip) that profiles/ and
    # the roofline pass use; otherwise the decoder also runs beside the RPN / box branch on a side stream
    overlap = not (args.streams == 1 and args.no_graphs) and not args.no_overlap
    eng.overlap_decoder = overlap
    hw = (args.height, args.width)
    frames = make_frames(args.batch, rank * args.batch, hw, device)  # weak scaling: every rank owns `batch` frames
    torch.cuda.synchronize()

    def step():
        return pred.predict_batch(frames)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: consecutive batches alternate between `--pipeline` stream lanes, so the DensePose-head phase of batch i
    # (launched once the host has read batch i's detection counts) runs beside the backbone / RPN phase of batch i+1.
    # Every batch is complete when the closing device synchronize returns; nothing is skipped or carried over.
    pipelined = not (args.streams == 1 and args.no_graphs)
    pred.pipeline_depth = args.pipeline if pipelined else 1
    for _ in range(max(args.warmup, 2 * pred.pipeline_depth)):   # at least one capture + one replay per lane
        out = step()
    barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    pred.join()
    barrier()
    elapsed = time.perf_counter() - t_begin
    pred.pipeline_depth = 1   # the latency loop and the roofline pass below run batch after batch on one stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dets = [int(o["scores"].shape[0]) for o in out]
    # p50 latency of one batch (synchronised per step), reported per image
    step_times = []
    for _ in range(max(5, args.steps)):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        step_times.append(time.perf_counter() - t0)
    flops_step = eng.flops_last

    # ---- roofline of the dominant kernel: K further steps of the same workload, every conv launch bracketed by HIP events
    # on the stream it is launched on. This pass is fully serialized - one stream, no graph replay, no side stream, no
    # pipeline lane - so each kernel has the chip to itself and the event delta is its own duration (in the timed region the
    # decoder and the next batch share the chip with the launch being timed, which inflates per-launch durations).
    # profiles/ holds the rocprofv3 summary of the same serialized configuration (`--streams 1 --no-graphs`), whose
    # per-kernel averages must agree with these.
    if args.no_roofline:
        if rank == 0:
            print(json.dumps({"value": round(args.batch * world * args.steps / elapsed, 3), "ms_per_step": round(1e3 * elapsed / args.steps, 3)}), flush=True)
        return
    pred.num_streams, eng.use_graphs, eng.overlap_decoder = 1, False, False
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    eng.prof = []
    t_ser = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t_ser = (time.perf_counter() - t_ser) / args.steps
    agg = {}
    for cls, flops, e0, e1, name, _ in eng.prof:
        a = agg.setdefault(cls, [0, 0.0, 0])
        a[0] += flops
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
    eng.prof = None
    pred.num_streams, eng.use_graphs, eng.overlap_decoder = args.streams, not args.no_graphs, overlap
    dom = max(agg, key=lambda c: agg[c][1])
    dflops, dsec, dcalls = agg[dom]
    peak = PEAK_F32_MATRIX if args.dtype == "fp32" else PEAK_BF16_DENSE  # fp16 and bf16 MFMA share the dense peak
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")
    if os.path.exists(tpath) and args.dtype == "bf16" and args.config == "densepose_rcnn_R_50_FPN_s1x" and args.batch == 8:
        k = json.load(open(tpath))["kernels"].get(dom + "[bf16]")
        if k:
            traffic = {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "source": "profiles/r1_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"}
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(dflops / dsec / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(dflops / dsec / peak, 4), "traffic": traffic,
                "launches_per_step": dcalls // args.steps, "avg_launch_us": round(1e6 * dsec / dcalls, 2),
                "alg_gflop_per_launch": round(dflops / dcalls / 1e9, 3),
                "share_of_serialized_step": round(dsec / args.steps / t_ser, 3),
                # all convolution FLOPs of a step over the step time of the timed region (every kernel included, not a kernel roofline)
                "whole_step_tflops": round(flops_step / (elapsed / args.steps) / 1e12, 1),
                "measured": "K event-instrumented steps on one stream (kernels alone on the chip), %.2f ms/step serialized" % (1e3 * t_ser),
                "all_conv_classes": {c: {"tflops": round(v[0] / v[1] / 1e12, 2), "calls_per_step": v[2] // args.steps,
                                         "ms_per_step": round(1e3 * v[1] / args.steps, 3)} for c, v in agg.items()}}
    # the 256-cout ring kernel is one source with one template instance (= one rocprofv3 kernel name) per tile height
    fam = [v for c, v in agg.items() if c.startswith("conv_ring_kernel<") and c.endswith("x256>")]
    if fam:
        ff, ft, fc = sum(v[0] for v in fam), sum(v[1] for v in fam), sum(v[2] for v in fam)
        roofline["ring256_family"] = {"tflops": round(ff / ft / 1e12, 2), "frac": round(ff / ft / peak, 4), "calls_per_step": fc // args.steps,
                                      "ms_per_step": round(1e3 * ft / args.steps, 3)}

    result = None
    if rank == 0:
        images = args.batch * world * args.steps
        result = {
            "metric": "images/sec at 1/2/4/8 MI355X, R_50_FPN_s1x 800x1333; p50 ms/img" if args.config == "densepose_rcnn_R_50_FPN_s1x"
                      else "images/sec, %s 800x1333" % args.config,
            "value": round(images / elapsed, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.dtype], "data": "synthetic",
            "p50_ms_per_img": round(1e3 * float(np.median(step_times)) / args.batch, 3),
            "config": {"workload": "%s batch=%d/GPU %dx%d uint8 frames resident in HBM, R=%d detections/img (measured %s), synthetic seeded weights"
                                   % (args.config, args.batch, hw[0], hw[1], args.dets, dets),
                       "global_batch": args.batch * world, "parallelism": "frame-sharded dp%d, no hot-loop collective" % world,
                       "alg_gflop_per_image": round(flops_step / args.batch / 1e9, 1),
                       "schedule": "%d batch stream(s), %s, decoder %s, %d pipeline lane(s)"
                                   % (args.streams, "eager launches" if args.no_graphs else "HIP-graph replay of the static part",
                                      "on a side stream" if overlap else "in line", args.pipeline if pipelined else 1)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, state, hw, args.cpu_budget_s)
            result["config"]["gpu_over_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
