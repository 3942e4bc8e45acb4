"""Drop-in counterpart of the reference's predictor callable.

    predictor(original_image: uint8 Tensor[H,W,3] | [3,H,W], bgr: bool = True) -> Dict[str, Tensor]

mirrors ``DefaultPredictor.forward`` (/root/reference/detectron2/engine/defaults.py:65-97) and returns the dict
of ``detector_postprocess`` (/root/reference/detectron2/modeling/postprocessing.py:52-61): same 8 keys, dtypes,
shapes and value semantics; tensors live on the GPU like the reference's CUDA mode (``run.py:22-29``).
Construction mirrors ``export.py:22-34``: a config name / yaml (+ override list) and a ``.pkl`` checkpoint.

Extras the reference does not have: ``predict_batch`` (N frames -> N dicts, equal to N single calls),
frame sharding over ranks (parallel.py) and a GPU-side resize (bit-exact restatement of the CPU uint8 kernel).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .config import ModelConfig, get_config
from .engine import Engine
from .options import EngineOptions
from .resize import resize_u8_device_batch
from .weights import check_state, load_checkpoint


class _HostFrameRing:
    """Pinned staging slots for HOST-resident frames (the reference's boundary hands over CPU tensors: defaults.py:65-80,
    run.py:34-36). A pageable tensor cannot be copied asynchronously - `.to(device, non_blocking=True)` on it stages through
    the driver's own bounce buffer on the caller's thread and stream. Here the frames of a batch are gathered into one
    page-locked buffer (one host memcpy each, torch's multi-threaded copy), ONE H2D transfer per batch runs on a dedicated
    copy stream, and the compute lane only waits for its event: the upload of batch i + 1 overlaps the kernels of batch i.
    `slots` buffers per frame geometry rotate; a slot is rewritten only after its previous upload has completed (host wait on
    its event, normally long past) and its device landing buffer only after the resize kernel that read it has run."""

    def __init__(self, device, copy_stream, slots=3, workers=None):
        if workers is None:      # a rank of a multi-GPU run gets its share of the host, not the single-process default
            from .parallel import host_workers
            workers = host_workers(4)
        self.workers = workers
        self.device, self.n_slots = device, slots
        self.copy_stream = copy_stream
        self.rings = {}      # (n, frame shape) -> [slot dicts]
        self.cursor = {}
        # the gather into the pinned slot is plain memcpy work: a few worker threads (ctypes.memmove releases the GIL) instead of
        # torch's copy_, whose OpenMP region over every core of a 256-thread host costs milliseconds per 3 MB frame
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=workers)

    @staticmethod
    def _gather_one(dst, src):
        if src.is_contiguous():
            import ctypes
            ctypes.memmove(dst.data_ptr(), src.data_ptr(), src.numel())
        else:
            dst.copy_(src)

    def upload(self, frames, consumer_stream):
        """frames: list of equal-shape CPU uint8 tensors (any strides). -> (device tensor [n, *shape], slot); the caller records
        slot["consumed"] on the stream that read the device tensor once its kernel is launched."""
        for f in frames:      # the landing buffers are uint8 and a contiguous frame is copied byte-wise: anything else would arrive truncated
            if f.dtype != torch.uint8:
                raise TypeError("host frames must be uint8 tensors (defaults.py:76-80), got %s" % f.dtype)
        key = (len(frames), tuple(frames[0].shape))
        ring = self.rings.get(key)
        if ring is None:
            ring = self.rings[key] = [
                {"pinned": torch.empty((len(frames),) + tuple(frames[0].shape), dtype=torch.uint8, pin_memory=True),
                 "dev": torch.empty((len(frames),) + tuple(frames[0].shape), dtype=torch.uint8, device=self.device),
                 "h2d_done": torch.cuda.Event(), "consumed": torch.cuda.Event(), "used": False}
                for _ in range(self.n_slots)]
            self.cursor[key] = 0
        slot = ring[self.cursor[key]]
        self.cursor[key] = (self.cursor[key] + 1) % self.n_slots
        if slot["used"]:
            slot["h2d_done"].synchronize()                  # the previous upload out of this pinned buffer has finished
            self.copy_stream.wait_event(slot["consumed"])   # ... and the kernel that read the landing buffer has run
        # host memcpy into the page-locked slot (a strided view is laid out contiguously by copy_ instead)
        list(self.pool.map(self._gather_one, [slot["pinned"][i] for i in range(len(frames))], frames))
        with torch.cuda.stream(self.copy_stream):
            slot["dev"].copy_(slot["pinned"], non_blocking=True)
            slot["h2d_done"].record(self.copy_stream)
        consumer_stream.wait_event(slot["h2d_done"])
        slot["used"] = True
        return slot["dev"], slot


class DensePosePredictor:
    def __init__(self, cfg, weights, dtype="bf16", device="cuda:0", resize="host", num_streams=1, use_graphs=False, check_keep=False,
                 pipeline_depth=1, nms_reference="cpu", options=None):
        """cfg: ModelConfig | variant name | yaml path. weights: path to .pkl/.pth or a canonical state dict.
        options: options.EngineOptions (the A/B switches; default = what the product runs; nothing here reads the environment)."""
        if not isinstance(cfg, ModelConfig):
            cfg = get_config(cfg)
        self.cfg = cfg
        state = load_checkpoint(weights, cfg) if isinstance(weights, str) else weights
        check_state(cfg, state)
        assert cfg.input_format in ("RGB", "BGR"), cfg.input_format
        self.input_format = cfg.input_format
        self.min_size = cfg.min_size
        self.max_size = cfg.max_size
        self.options = options if options is not None else EngineOptions()
        self.engine = Engine(cfg, state, dtype=dtype, device=device, options=self.options)
        self.device = self.engine.device
        self.resize_mode = resize  # "host": torch CPU uint8 kernel exactly as the reference (Q4) ; "device": HIP kernel
        self.num_streams = num_streams  # sub-batches of a batch run concurrently on this many HIP streams
        self.engine.use_graphs = use_graphs  # replay the static part of the path (backbone .. detection select) as a HIP graph
        # HIP graphs are for fixed-geometry streams (a video, a benchmark): one graph per (batch shape, lane), at most
        # engine.MAX_GRAPHS kept (least recently used dropped); a stream of ever-changing frame sizes should run eagerly.
        # nms_reference: "cpu" reproduces torchvision's batched_nms strategy switch of the reference's CPU mode (what the golden
        # vectors were recorded with; default), "cuda" the one of its CUDA mode (run.py:22-29) - see engine.NMS_TRICK_MAX_NUMEL
        assert nms_reference in ("cpu", "cuda"), nms_reference
        self.engine.nms_reference = nms_reference
        # postprocessing.py:51 drops boxes with negative extent after rescaling. That cannot happen on this path: decoded
        # boxes have w = exp(dw) * w_src >= 0 (box_regression.py:104-105), clipping and the positive rescale are monotonic,
        # non-finite boxes were filtered before NMS. The flags are still computed on the device; reading them back costs a
        # device synchronisation per batch, so it is opt-in (tests run with check_keep=True).
        self.check_keep = check_keep
        # > 1: consecutive predict_batch calls alternate between this many streams without joining the caller's stream
        # (see predict_batch / join); 1 = every call is ordered on the caller's stream like the reference's module call
        self.pipeline_depth = pipeline_depth
        self._lanes, self._next_lane = [], 0
        self._last_done = []   # completion events of the most recent predict_batch call (one per frame group)
        self._host_ring = None  # _HostFrameRing, created on the first host-resident frame
        self.fuse_resize = bool(self.options.fuse_resize)            # device resize at scale != 1: fused with the preprocess
        self.identity_resize = bool(self.options.identity_resize)    # frames that already have the test size skip the two resize passes

    # -- defaults.py:76-89 ---------------------------------------------------------------------------------
    def _to_chw(self, original_image, bgr):
        if original_image.shape[2] == 3:
            original_image = original_image.permute(2, 0, 1)
        else:
            assert original_image.shape[0] == 3, (
                "Only 3 channels expected either in HWC or CHW format, got {}".format(original_image.shape))
        if self.input_format == "RGB" and bgr:
            original_image = original_image.flip(0)
        return original_image

    def _scale(self, height, width):
        return min(self.min_size / min(height, width), self.max_size / max(height, width))

    def _resize_group(self, chws):
        """chws: CHW views of ONE geometry -> (uint8 [n,3,oh,ow] on the device, False) (defaults.py:89, once per frame there) - or,
        for frames that already have the test size (scale 1: the uint8 bilinear resize is the identity, resize.axis_table gives
        weights (1, 0)), the frames themselves as ([n,h,w,3] tensor or list of [h,w,3] device frames, True): no resize launch."""
        from .resize import output_size
        height, width = int(chws[0].shape[1]), int(chws[0].shape[2])
        k = self._scale(height, width)
        cur = torch.cuda.current_stream(self.device)
        if self.resize_mode == "device":
            # permuted views of contiguous HWC frames are resized straight from HWC
            hwc = all(c.stride(0) == 1 and c.stride(2) == 3 for c in chws)
            views = [(c.permute(1, 2, 0) if hwc else c) for c in chws]
            identity = hwc and k == 1.0 and output_size(height, width, k) == (height, width) and self.identity_resize
            if all(not v.is_cuda for v in views):
                # host-resident frames (the reference's boundary): pinned ring + one H2D per batch on the copy stream
                if self._host_ring is None:
                    self._host_ring = _HostFrameRing(self.device, self.engine.new_stream())
                dev, slot = self._host_ring.upload(views, cur)
                if identity:
                    out = dev.clone()      # (the engine keeps reading its batch after this slot has been handed out again)
                else:
                    out = resize_u8_device_batch(self.engine, [dev[i] for i in range(len(views))], k, src_hwc=hwc)
                slot["consumed"].record(cur)
                return out, identity
            frames = [v.to(self.device, non_blocking=True) for v in views]
            if identity:
                return [f.contiguous() for f in frames], True
            if self.fuse_resize and self.num_streams == 1 and len(frames) <= 64:
                # scale != 1 (every video frame): vertical pass + normalise + pad + layout in one launch, straight into the stem's input
                from .resize import FusedResize
                return FusedResize(self.engine, frames, k, src_hwc=hwc), False
            return resize_u8_device_batch(self.engine, frames, k, src_hwc=hwc), False
        # "host": torch's CPU uint8 kernel exactly as the reference runs it; the resized frames go up through the same ring
        small = [F.interpolate(c.cpu()[None], scale_factor=k, mode="bilinear", align_corners=False)[0] for c in chws]
        if self._host_ring is None:
            self._host_ring = _HostFrameRing(self.device, self.engine.new_stream())
        dev, slot = self._host_ring.upload(small, cur)
        out = dev.clone()                    # the batch tensor outlives the slot (graph replay copies from it later)
        slot["consumed"].record(cur)
        return out, False

    @torch.no_grad()
    def __call__(self, original_image, bgr=True):
        return self.predict_batch([original_image], bgr)[0]

    forward = __call__

    @torch.no_grad()
    def predict_batch(self, images, bgr=True):
        """N frames -> N dicts. Frames whose resized size is equal are run as one batch through the kernels.

        With ``pipeline_depth > 1`` consecutive calls alternate between that many HIP streams ("lanes") and the caller's
        stream is NOT made to wait for the lane: the DensePose-head phase of one batch (launched after the host read the
        detection counts) then runs beside the backbone / RPN phase of the next batch, which fills the partial last
        rounds of its launches. The returned tensors are valid once ``join()`` (or a device synchronize) has been called."""
        chws = [self._to_chw(im if torch.is_tensor(im) else torch.from_numpy(np.asarray(im)), bgr) for im in images]
        groups = {}
        for i, c in enumerate(chws):   # frames of one geometry (a video: run.py:42-57) share the resize launch and the batch
            groups.setdefault((tuple(c.shape), c.stride(0) == 1 and c.stride(2) == 3), []).append(i)
        out = [None] * len(images)
        cur = torch.cuda.current_stream(self.device)
        self._last_done = []
        for idxs in groups.values():
            lane, stream = 0, cur
            if self.pipeline_depth > 1:
                if len(self._lanes) < self.pipeline_depth:
                    self._lanes = [self.engine.new_stream() for _ in range(self.pipeline_depth)]
                lane = self._next_lane
                self._next_lane = (lane + 1) % self.pipeline_depth
                stream = self._lanes[lane]
                ready = torch.cuda.Event()
                ready.record(cur)          # the frames are ready on the caller's stream
                stream.wait_event(ready)
                for i in idxs:
                    if chws[i].is_cuda:
                        chws[i].record_stream(stream)   # the caller may drop its frames as soon as this call returns
            with torch.cuda.stream(stream):
                batch, hwc = self._resize_group([chws[i] for i in idxs])
                orig = [(int(chws[i].shape[1]), int(chws[i].shape[2])) for i in idxs]
                res = self.engine.forward_batch(batch, orig, num_streams=self.num_streams if len(idxs) >= 4 else 1, slot=lane, hwc=hwc)
                if self.check_keep:
                    res = self.engine.apply_keep_filter(res)
            if stream is not cur:
                done = torch.cuda.Event()
                done.record(stream)
                self._last_done.append(done)
                for r in res:
                    for t in r.values():
                        if t.is_cuda:
                            t.record_stream(cur)   # the caller will read them on its own stream after join()
            for i, r in zip(idxs, res):
                out[i] = r
        return out

    def join(self):
        """Make the caller's stream wait for every pipeline lane (no host synchronisation)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self._lanes:
            cur.wait_stream(s)

    def completion(self):
        """Handle for the batch the most recent predict_batch call submitted: pass it to wait() to make the caller's stream wait
        for THAT batch only - unlike join(), later batches on the other lanes keep running (run.py: post-processing of batch
        i-1 beside batch i)."""
        return list(self._last_done)

    def wait(self, handle):
        cur = torch.cuda.current_stream(self.device)
        for ev in handle:
            cur.wait_event(ev)
