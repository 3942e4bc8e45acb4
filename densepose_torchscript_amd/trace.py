"""Optional tracing hooks of the engine (off by default, zero cost when off).

* per-stage HIP-event timers: ``Engine.trace = StageTrace(device)`` brackets every stage of the path (preprocess, stem,
  res2..res5, fpn, rpn, box head, decoder, DensePose head, predictor, postprocess) with events recorded on the stream
  the stage is launched on; ``summary()`` returns milliseconds and algorithmic FLOPs per stage. Used by ``bench.py``
  for the backbone roofline and by ``tools/prof_layers.py``.
* roctx ranges: with ``roctx=True`` (or ``DP_ROCTX=1``) the same brackets also push/pop a named range through
  libroctx64, so a ``rocprofv3 --marker-trace`` timeline shows the stages above the kernels.

The reference has no counterpart (it relies on torch's profiler around the TorchScript call); SURVEY.md §5 row 1.
"""
import contextlib
import ctypes
import os

import torch


def _load_roctx():
    for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            return lib
        except (OSError, AttributeError):
            continue
    return None


class StageTrace:
    def __init__(self, device, roctx=False, timers=True):
        self.device = torch.device(device)
        self.roctx = _load_roctx() if roctx else None
        self.timers = timers
        self.records = []   # (stage name, start event, end event, flops)

    @contextlib.contextmanager
    def stage(self, name, engine=None):
        if self.roctx is not None:
            self.roctx.roctxRangePushA(name.encode())
        e0 = e1 = None
        f0 = engine.flops_last if engine is not None else 0
        if self.timers:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        try:
            yield
        finally:
            if self.timers:
                e1.record(torch.cuda.current_stream(self.device))
                self.records.append((name, e0, e1, (engine.flops_last - f0) if engine is not None else 0))
            if self.roctx is not None:
                self.roctx.roctxRangePop()

    def reset(self):
        self.records = []

    def summary(self):
        """{stage: {"ms": total milliseconds, "gflop": algorithmic GFLOP, "calls": n}} (synchronises the device)."""
        torch.cuda.synchronize(self.device)
        out = {}
        for name, e0, e1, flops in self.records:
            r = out.setdefault(name, {"ms": 0.0, "gflop": 0.0, "calls": 0})
            r["ms"] += e0.elapsed_time(e1)
            r["gflop"] += flops / 1e9
            r["calls"] += 1
        return out
