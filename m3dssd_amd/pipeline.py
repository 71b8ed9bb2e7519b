"""Throughput mode: a software-pipelined detector.

The post-forward stage (64-bit-key top-k, decode, NMS, row selection -- lib/rpn_util.py:1442-1555 in the reference)
is latency-bound and leaves the chip mostly idle for ~0.35 ms per batch.  ``PipelinedDetector`` overlaps it with
the forward of the NEXT batch: one captured hipGraph per input shape whose two branches are

    branch A (main stream):  forward(batch k)  [all launches except the final output bundling]
    branch B (side stream):  detect(batch k-1) on the output buffers written by the previous replay
    join, then bundle_outputs(batch k)  -> the output buffers branch B of the next replay will read

so each ``step(x)`` returns the detections of the PREVIOUS batch (one batch of pipeline latency) and ``flush()``
drains the last one.  Results are identical to ``lib.rpn_util.detect_batch`` (tests/test_gpu_parity.py).
"""
import torch

from .host.detect import detect_from_outputs, select_block


class PipelinedDetector:
    def __init__(self, net, conf, batch, height, width):
        self.net, self.conf = net, conf
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise NotImplementedError("PipelinedDetector needs the module on a ROCm device")
        self.dev = dev
        self.eng = net.engine()
        self.plan = self.eng.plan_for(batch, height, width)
        self.input = torch.zeros(batch, 3, height, width, device=dev, dtype=torch.float32)
        self.n_fwd = len(self.plan.ops) - 1
        assert self.plan.ops[-1][0] == "bundle_outputs"
        n = self.plan.named
        self._outs = (n["prob"], n["bbox_2d"], n["bbox_3d"])
        self._rois = net.rois.to(dev)
        self._pending = False
        self._build()

    def _detect(self):
        prob, b2, b3 = self._outs
        return select_block(*detect_from_outputs(self.eng, self.plan, prob, b2, b3, self._rois, self.conf), self.conf)

    def _forward(self, start, end):
        self.plan.named["input_ptr"][0] = self.input.data_ptr()
        self.eng.run_plan(self.plan, start, end)

    def _build(self):
        cap, side = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        cap.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(cap), torch.no_grad():
            self._forward(0, None)                       # warm-up outside capture: plan buffers, allocator pools,
            self._detect()                               # kernel attributes, zero page
            torch.cuda.synchronize(self.dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=cap):
                side.wait_stream(cap)                    # fork
                with torch.cuda.stream(side):
                    self._block, self._counts = self._detect()      # batch k-1
                self._forward(0, self.n_fwd)             # batch k, everything but the bundling
                cap.wait_stream(side)                    # join: outputs may now be overwritten
                self._forward(self.n_fwd, None)
            # tail graph for flush(): detect only
            self.tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.tail, stream=cap):
                self._tblock, self._tcounts = self._detect()
        torch.cuda.current_stream(self.dev).wait_stream(cap)

    def step(self, x=None, as_block=False):
        """Submit batch k (copied into ``self.input`` unless x is None = already written there); returns
        (dets, counts) of batch k-1, or None for the first call.  Returned tensors are overwritten by the next step.
        as_block: return the [B, nms_topN_post + 1, 14] gather block (m3dssd_amd.dist.gather_block) instead of dets."""
        if x is not None:
            self.input.copy_(x)
        had = self._pending
        self.graph.replay()
        self._pending = True
        if not had:
            return None
        return (self._block if as_block else self._block[:, :-1], self._counts)

    def flush(self, as_block=False):
        """Detections of the last submitted batch."""
        if not self._pending:
            return None
        self.tail.replay()
        self._pending = False
        return (self._tblock if as_block else self._tblock[:, :-1], self._tcounts)
