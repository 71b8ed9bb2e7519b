"""Throughput mode: a software-pipelined detector.

The post-forward stage (64-bit-key top-k, decode, NMS, row selection -- lib/rpn_util.py:1442-1555 in the reference)
is latency-bound and leaves the chip mostly idle for ~0.35 ms per batch.  ``PipelinedDetector`` overlaps it with
the forward of the NEXT batch: one captured hipGraph per input shape whose two branches are

    branch A (main stream):  forward(batch k)  [the launches in front of the first head that writes the planar staging]
    branch B (side stream):  detect(batch k-1) on the planar head outputs + sort keys the previous replay left
    join, then the heads of batch k and ``m3d_score_keys_planar`` -> what branch B of the next replay will read

so each ``step(x)`` returns the detections of the PREVIOUS batch (one batch of pipeline latency) and ``flush()``
drains the last one.  Results are identical to ``lib.rpn_util.detect_batch`` (tests/test_gpu_detect.py).

The detection stage needs 3000 decoded rows per image, not the bundled ``cls / prob / bbox_2d / bbox_3d`` tensors
(M3d_inference_align.py:280-301: 38 MB per image written and read back, 2.5 GB per step at bs = 64): ``planar=True`` (default)
replaces ``m3d_bundle_outputs`` by the key-only pass and decodes from the planar staging (``m3d_topk_decode_planar``, same
bits); ``planar=False`` (or M3D_PIPE_PLANAR=0) is the bundled form: detect(k-1) beside forward(k), join, bundle(k).  In the
planar form ``plan.named["prob" / "bbox_2d" / "bbox_3d" / "cls"]`` are NOT written by a replay.

``u8_frame=(h, w)`` is the fed-input form (the reference pays ``im.cuda()`` per frame, lib/rpn_util.py:1427-1429,
and preprocesses on the host, lib/dataloader.py:934-950, lib/augmentations.py:472-501): raw uint8 BGR frames [B, h, w, 3] sit in
PINNED host memory; the graph of batch k carries, on its side branch, an upload KERNEL (``m3d_upload_indirect``) that streams
the frames of batch k + 1 over PCIe into the second of TWO device buffers while the forward of batch k runs on the first, and the
stem reads uint8 frames directly (``m3d_stem_conv7x7_u8``: padding to the crop size, /255, -mean, /stds, BGR->RGB in its loads).
Two graphs are captured, one per input buffer; which host buffer a replay uploads is named by an 8-byte word in pinned host
memory that the host writes before the replay (no copy engine, no copy stream, no events: a hipMemcpyAsync next to the graph
cost more step time than a serial copy on this platform, tools/feed_probe.py).

``refine=True`` appends the post-NMS 3-D refinement of ``test_kitti_3d`` (lib/rpn_util.py:1796-1847: back to the original image
scale, clipping, alpha -> ry, hill climbing, back-projection; ``m3d_refine_3d_ex``) to branch B, reading the selected rows where
``m3d_select_post`` left them: a replay then yields the rows of the KITTI result files of batch k-1.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _hip
from .host.detect import detect_from_outputs, detect_from_planar, score_keys_planar, select_block, unwrap
from .host.refine import p2_arrays


class PipelinedDetector:
    def __init__(self, net, conf, batch, height, width, refine=False, score_thresh=0.75, step_r_init=0.3 * math.pi, r_lim=0.01,
                 u8_frame=None, planar=None):
        net = unwrap(net)                               # (a DataParallel / DDP wrapper of the module, as the reference's scripts pass it)
        self.net, self.conf = net, conf
        self.planar = (os.environ.get("M3D_PIPE_PLANAR", "1") != "0") if planar is None else bool(planar)
        self.u8_frame = None if u8_frame is None else (int(u8_frame[0]), int(u8_frame[1]))
        self.refine = bool(refine)
        self._rargs = (float(score_thresh), 1 if bool(getattr(conf, "hill_climbing", True)) else 0, float(step_r_init), float(r_lim))
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise NotImplementedError("PipelinedDetector needs the module on a ROCm device")
        self.dev = dev
        self.eng = net.engine()
        self.plan = self.eng.plan_for(batch, height, width)
        if self.u8_frame is None:
            self.input = torch.zeros(batch, 3, height, width, device=dev, dtype=torch.float32)
        else:
            fh, fw = self.u8_frame
            if fh > height or fw > width:
                raise RuntimeError("u8_frame %dx%d does not fit the padded size %dx%d" % (fh, fw, height, width))
            nbytes = batch * fh * fw * 3
            self._u8_bytes = nbytes
            self._u8_flat = [torch.zeros(-(-nbytes // 16) * 16, device=dev, dtype=torch.uint8) for _ in range(2)]
            self.inputs_u8 = [t[:nbytes].view(batch, fh, fw, 3) for t in self._u8_flat]
            self.input = self.inputs_u8[0]
            # slot i: address of the pinned-host frames that the graph reading buffer i uploads into buffer i ^ 1 (0 = nothing)
            self._slots = torch.zeros(2, dtype=torch.int64).pin_memory()
            self._done = [torch.cuda.Event(), torch.cuda.Event()]      # the last replay of graph i finished (main stream)
            self._used = [False, False]
            self._queue = []                                           # fed, not yet submitted frame sets (kept alive here)
            self._cur = 0                                              # buffer the next submitted batch is read from
            self._primed = False                                       # buffer _cur holds the head of the queue
            self._inflight = [None, None]                              # frames a replay of graph i is uploading (kept alive)
            self._prime_ev = None                                      # behind feed()'s own copy of a batch nothing in flight uploads
        self.n_fwd = len(self.plan.ops) - 1
        assert self.plan.ops[-1][0] == "bundle_outputs"
        n = self.plan.named
        # planar form: the side branch must be done before the first launch that overwrites the planar staging
        # bundled form: detect(k-1) reads prob / bbox_* / score_bits; bundle_outputs (the last op) writes the first three, but with
        # SELECT_KEYS anchor_select of batch k already writes score_bits in the middle of the forward (that write ran
        # unordered against the side branch's top-k): join in front of the FIRST op that writes anything the side branch reads
        self.n_join = int(n["planar_first_op"]) if self.planar else min(self.n_fwd, int(n.get("score_bits_first_write_op", self.n_fwd)))
        self._check_no_write_beside_detect()
        # where the side branch forks off (experiment switch M3D_PIPE_FORK = op index; default: at the first launch).  In the bf16 graph
        # the persistent front end is stretched from 0.83 to 1.09 ms by detect(k-1) beside it (tools/graph_timeline.py), but forking
        # behind it moves the cost, it does not remove it: 10.48 / 10.58 / 10.61 / 10.52 ms for forks at op 0 / 1 / 2 / 3 on one lease
        # (fp32: 6.00 / 6.01 / 5.99 / 6.04, later forks worse) -- detect's 0.3 ms of mask arithmetic is work, not latency.
        fork = os.environ.get("M3D_PIPE_FORK")
        self.n_fork = min(int(fork) if fork is not None else int(n.get("pipe_fork_op", 0)), self.n_join)
        self._outs = (n["prob"], n["bbox_2d"], n["bbox_3d"])
        self._rois = net.rois.to(dev)
        self._pending = False
        if self.refine:
            # calibration / scale / image size of the batch whose detections the NEXT replay refines; `_meta_next` (host) holds
            # those of the batch submitted last
            self._p2 = torch.zeros(batch, 16, device=dev, dtype=torch.float64)
            self._p2_inv = torch.zeros(batch, 16, device=dev, dtype=torch.float64)
            self._p2[:, 0::5] = 1.0
            self._p2_inv[:, 0::5] = 1.0
            self._scale = torch.ones(batch, device=dev, dtype=torch.float32)
            self._clip = torch.zeros(batch, 2, device=dev, dtype=torch.float32)
            self._meta_next = None
        self._build()

    def _check_no_write_beside_detect(self):
        """No op of forward(k) in [0, n_join) may write a buffer that detect(k-1) reads: the ops that do are known by name
        (the heads and anchor_select write the planar staging / the sort keys, bundle_outputs the bundled tensors)."""
        n = self.plan.named
        first_writer = {"planar": int(n["planar_first_op"]), "keys": int(n.get("score_bits_first_write_op", self.n_fwd)),
                        "bundled": self.n_fwd}
        reads = ("planar", "keys") if self.planar else ("keys", "bundled")
        limit = min(first_writer[r] for r in reads)
        if self.n_join > limit:
            raise AssertionError("PipelinedDetector: join at op %d, but op %d of the next forward writes what detect reads (%s)"
                                 % (self.n_join, limit, first_writer))

    def _detect(self):
        prob, b2, b3 = self._outs
        # refine mode carries the frames' test-time scale factors: the boxes are divided by them inside the decode, before the NMS
        # (lib/rpn_util.py:1504-1506), not in the refinement behind it
        scale = self._scale if self.refine else None
        if self.planar:
            rows = detect_from_planar(self.eng, self.plan, self._rois, self.conf, scale)
        else:
            rows = detect_from_outputs(self.eng, self.plan, prob, b2, b3, self._rois, self.conf, scale)
        block, counts = select_block(*rows, self.conf)
        if not self.refine:
            return block, counts, None
        B, K1, _ = block.shape                          # K1 = kept rows + the count row (past counts[b]: refined to zeros)
        out = torch.empty(B, K1, 16, device=self.dev, dtype=torch.float64)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        _hip.check(_hip.lib().m3d_refine_3d_ex(block.data_ptr(), counts.data_ptr(), B, K1, self._p2.data_ptr(),
                                               self._p2_inv.data_ptr(), None, self._clip.data_ptr(),
                                               *self._rargs, out.data_ptr(), st))
        return block, counts, out

    def _forward(self, start, end, buf=0):
        if self.u8_frame is None:
            self.plan.named["input_ptr"][0] = self.input.data_ptr()
            self.eng.run_plan(self.plan, start, end)
            return
        # the stem launch reads (pointer, h, w) of the uint8 frames when it is ISSUED: capture bakes buffer `buf` into the graph
        self.plan.named["input_u8"][:] = [self.inputs_u8[buf].data_ptr(), self.u8_frame[0], self.u8_frame[1]]
        try:
            self.eng.run_plan(self.plan, start, end)
        finally:
            self.plan.named["input_u8"][0] = 0

    def _finish(self, buf=0):
        """The launches behind the join: the rest of the forward, then the key-only pass (planar) or the output bundling."""
        if self.planar:
            if self.n_join < self.n_fwd:
                self._forward(self.n_join, self.n_fwd, buf)
            if not self.plan.named.get("keys_by_select"):    # else anchor_select wrote them on the way (engine: SELECT_KEYS)
                score_keys_planar(self.eng, self.plan)
        else:
            self._forward(self.n_join, None, buf)

    def _capture_step(self, cap, side, buf):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            if self.n_fork:                          # launches of batch k in front of the fork (they do not share the chip well)
                self._forward(0, self.n_fork, buf)
            side.wait_stream(cap)                    # fork
            with torch.cuda.stream(side):
                outs = self._detect()                # batch k-1
                if self.u8_frame is not None:        # frames of batch k+1: pinned host -> the OTHER input buffer, as a kernel
                    _hip.check(_hip.lib().m3d_upload_indirect(
                        ctypes.c_void_p(self._slots.data_ptr() + 8 * buf), ctypes.c_void_p(self._u8_flat[buf ^ 1].data_ptr()),
                        self._u8_bytes, ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
            self._forward(self.n_fork, self.n_join, buf)   # batch k, everything in front of the first write of what the side branch reads
            cap.wait_stream(side)                    # join: outputs may now be overwritten
            self._finish(buf)
        return g, outs

    def _build(self):
        cap, side = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        cap.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(cap), torch.no_grad():
            self._forward(0, self.n_join)                # warm-up outside capture: plan buffers, allocator pools,
            self._finish()                               # kernel attributes, zero page
            self._detect()
            torch.cuda.synchronize(self.dev)
            self.graph, (self._block, self._counts, self._refined) = self._capture_step(cap, side, 0)
            if self.u8_frame is not None:                # second input buffer: its own graph, its own result tensors
                self._graphs = [self.graph, None]
                self._results = [(self._block, self._counts, self._refined), None]
                self._graphs[1], self._results[1] = self._capture_step(cap, side, 1)
            # tail graph for flush(): detect only
            self.tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.tail, stream=cap):
                self._tblock, self._tcounts, self._trefined = self._detect()
        torch.cuda.current_stream(self.dev).wait_stream(cap)

    def _upload_meta(self):
        """Calibration of the batch submitted LAST -> the device buffers the refinement of the next replay reads."""
        m = self._meta_next
        if m is None:
            return
        B = self.input.shape[0]
        p2, p2_inv = p2_arrays(m["p2"], B)
        self._p2.copy_(torch.from_numpy(p2.reshape(B, 16)))
        self._p2_inv.copy_(torch.from_numpy(p2_inv.reshape(B, 16)))
        scale = m.get("scale")
        self._scale.copy_(torch.from_numpy(np.ones(B, np.float32) if scale is None else np.asarray(scale, np.float32).reshape(B)))
        clip = m.get("clip_wh")
        self._clip.copy_(torch.from_numpy(np.zeros((B, 2), np.float32) if clip is None
                                          else np.asarray(clip, np.float32).reshape(B, 2)))

    # ---- fed-input form ---------------------------------------------------------------------------------------------------------
    def feed(self, frames):
        """Queue one batch of uint8 BGR frames [B, h, w, 3] in PINNED host memory (a device tensor works too).  Nothing is copied
        here: the graph of the batch submitted BEFORE this one uploads it (the first batch is uploaded right away) -- an
        asynchronous read of this tensor.  The tensor must stay unchanged until the step_fed() that SUBMITS it has returned:
        that call waits (on the host, behind its own replay launch, so the device never idles) for the replay / copy that
        uploaded the batch.  A ring of two pinned buffers written in the order feed, step_fed, feed, step_fed, ... is therefore
        safe.  At most two batches can be queued."""
        if self.u8_frame is None:
            raise RuntimeError("feed() needs PipelinedDetector(..., u8_frame=(h, w))")
        if len(self._queue) >= 2:
            raise RuntimeError("feed(): two batches are queued already; call step_fed()")
        if frames.dtype != torch.uint8 or tuple(frames.shape) != tuple(self.inputs_u8[0].shape) or not frames.is_contiguous():
            raise RuntimeError("feed(): contiguous uint8 frames of shape %s expected, got %s %s"
                               % (tuple(self.inputs_u8[0].shape), frames.dtype, tuple(frames.shape)))
        if not (frames.is_cuda or frames.is_pinned()):
            raise RuntimeError("feed(): the frames must be in pinned host memory (tensor.pin_memory()) or on the device")
        self._queue.append(frames)
        if not self._primed:                                     # nothing in flight names this batch: upload it now
            self.inputs_u8[self._cur].copy_(frames, non_blocking=True)
            self._prime_ev = torch.cuda.Event()                  # the copy reads the pinned tensor asynchronously
            self._prime_ev.record(torch.cuda.current_stream(self.dev))
            self._primed = True

    def step_fed(self, as_block=False):
        """Submit the oldest queued batch; its graph also uploads the next queued batch (if any) into the other buffer.  Returns
        (dets, counts) of the batch submitted before it (None for the first call).  The returned tensors belong to the graph of
        that input buffer and are overwritten two steps later."""
        if not self._queue:
            raise RuntimeError("step_fed(): no batch was fed")
        i = self._cur
        self._queue.pop(0)
        nxt = self._queue[0] if self._queue else None
        if self._used[i]:
            self._done[i].synchronize()          # (host) the previous replay of this graph has read its slot
        self._slots[i] = nxt.data_ptr() if nxt is not None else 0
        self._inflight[i] = nxt
        main = torch.cuda.current_stream(self.dev)
        had = self._pending
        self._graphs[i].replay()
        self._done[i].record(main)
        # The batch submitted here was uploaded by the PREVIOUS replay (graph i ^ 1, side branch) or by feed()'s own copy: wait
        # for that read of the caller's pinned tensor -- behind this replay's launch, so one replay is always queued on the
        # device -- and the documented lifetime ("unchanged until the step_fed() that submits it has returned") holds for a
        # two-deep ring as well (_done[i] above is two replays back and did not cover it).
        if self._used[i ^ 1]:
            self._done[i ^ 1].synchronize()
        if self._prime_ev is not None:
            self._prime_ev.synchronize()
            self._prime_ev = None
        self._used[i] = True
        self._pending = True
        self._cur ^= 1
        self._primed = nxt is not None           # the replay uploads it into the buffer the next step reads
        if not had:
            return None
        block, counts, _ = self._results[i]
        return (block if as_block else block[:, :-1], counts)

    def step(self, x=None, as_block=False, meta=None):
        """Submit batch k (copied into ``self.input`` unless x is None = already written there); returns
        (dets, counts) of batch k-1, or None for the first call.  Returned tensors are overwritten by the next step.
        as_block: return the [B, nms_topN_post + 1, 14] gather block (m3dssd_amd.dist.gather_block) instead of dets.
        refine mode: meta = {"p2": [B, 4, 4] (or [4, 4]), "scale": [B] or None, "clip_wh": [B, 2] or None} of batch k; the
        return value gains a third element, the refined rows [B, nms_topN_post, 16] (float64) of batch k-1."""
        if self.u8_frame is not None:
            if self.refine:
                raise NotImplementedError("refine=True with fed uint8 frames: use step() with float frames")
            if x is not None:
                self.feed(x)
            return self.step_fed(as_block)
        if x is not None:
            self.input.copy_(x)
        had = self._pending
        if self.refine:
            if meta is None:
                raise RuntimeError("PipelinedDetector(refine=True).step needs the batch's meta (p2, scale, clip_wh)")
            self._upload_meta()                      # of batch k-1: what this replay's refinement reads
            self._meta_next = meta
        self.graph.replay()
        self._pending = True
        if not had:
            return None
        res = (self._block if as_block else self._block[:, :-1], self._counts)
        return res + (self._refined[:, :-1],) if self.refine else res

    def flush(self, as_block=False):
        """Detections of the last submitted batch."""
        if not self._pending:
            return None
        if self.refine:
            self._upload_meta()
            self._meta_next = None
        self.tail.replay()
        self._pending = False
        res = (self._tblock if as_block else self._tblock[:, :-1], self._tcounts)
        return res + (self._trefined[:, :-1],) if self.refine else res
