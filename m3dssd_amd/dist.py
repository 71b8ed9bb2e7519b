"""Multi-GPU data parallelism for the detection path: one process per GPU, batch sharded by image,
ONE collective per batch.

Replaces the reference's single-process ``nn.DataParallel`` (scripts/test_rpn_3d.py:50-51,
lib/core.py:73-74: scatter -> replicate -> parallel_apply -> gather on GPU 0).  Images are independent
units, so every rank runs forward -> decode -> top-k -> NMS on its own shard with no data-path
communication; the only exchange is an ``all_gather_into_tensor`` of the fixed-size detection blocks
``[B/G, nms_topN_post, 14]`` + ``[B/G]`` counts (about 72 KB per rank at 32 images/GPU): latency-bound on
xGMI, so no bucketing / overlap machinery is warranted.  Backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist

ROW = 14  # x1,y1,x2,y2,score,cls,x3d,y3d,z3d,w3d,h3d,l3d,ry3d,anchor  (lib/rpn_util.py:1550)


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank).

    Backend: the argument, else $M3D_DIST_BACKEND, else "nccl" (= RCCL over xGMI) when a GPU is visible, "gloo" otherwise.
    With nccl every rank needs its own device: LOCAL_RANK >= device_count is a launch error and is reported as one (RCCL
    rejects or hangs on duplicate devices in one communicator); gloo (CPU tests, or two ranks sharing the one GPU of a test
    lease) maps ranks onto the visible devices round-robin."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("M3D_DIST_BACKEND")
            if backend is not None and backend not in ("nccl", "gloo"):      # only the env var is policed: an explicit argument
                raise ValueError("M3D_DIST_BACKEND must be 'nccl' or 'gloo' (got %r)" % (backend,))   # may be anything torch knows
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            n_dev = torch.cuda.device_count()
            if local >= n_dev:
                raise RuntimeError("LOCAL_RANK %d but only %d visible GPU(s): the nccl (RCCL) backend needs one device per "
                                   "rank (set M3D_DIST_BACKEND=gloo to share a device in tests)" % (local, n_dev))
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def backend_name():
    return dist.get_backend() if dist.is_initialized() else None


def shard_range(n_images, rank, world):
    """Contiguous block partition of the batch: rank r owns [lo, hi)."""
    if n_images % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_images, world))
    per = n_images // world
    return rank * per, (rank + 1) * per


def shard_batch(batch, rank, world):
    lo, hi = shard_range(batch.shape[0], rank, world)
    return batch[lo:hi]


_GATHER_OUT = {}


def gather_block(block, group=None):
    """block [b, P + 1, 14] float32 as written by ``m3d_select_post`` (row P = (count, 0, ...)), this rank's shard ->
    (all_dets [world*b, P, 14], all_counts [world*b] int32) identical on every rank, in global image order.
    ONE collective, no staging copy; the receive buffer is cached per shape (the returned tensors are views of it and
    are overwritten by the next call with the same shape)."""
    if block.dim() != 3 or block.shape[2] != ROW or not block.is_contiguous():
        raise ValueError("gather_block: block must be contiguous [b, P + 1, 14]")
    b, p1, _ = block.shape
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return block[:, :p1 - 1], block[:, p1 - 1, 0].to(torch.int32)
    world = dist.get_world_size(group)
    key = (block.device, b, p1, world, id(group))
    out = _GATHER_OUT.get(key)
    if out is None:
        out = _GATHER_OUT[key] = torch.empty(world * b, p1, ROW, device=block.device, dtype=torch.float32)
    dist.all_gather_into_tensor(out, block, group=group)
    return out[:, :p1 - 1], out[:, p1 - 1, 0].to(torch.int32)


def gather_detections(dets, counts, group=None):
    """dets [b, P, 14] float32, counts [b] int32 (this rank's shard) ->
    (all_dets [world*b, P, 14], all_counts [world*b]) identical on every rank, in global image order.
    (Callers that already hold the [b, P + 1, 14] block of ``select_block`` use ``gather_block`` and skip the staging copy.)"""
    if dets.dim() != 3 or dets.shape[2] != ROW or counts.shape[0] != dets.shape[0]:
        raise ValueError("gather_detections: dets must be [b, P, 14] and counts [b]")
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dets, counts
    b, p, _ = dets.shape
    # counts travel in the same message as the boxes (one collective, one latency): last row of the block
    block = torch.zeros(b, p + 1, ROW, device=dets.device, dtype=torch.float32)
    block[:, :p] = dets
    block[:, p, 0] = counts.to(torch.float32)
    return gather_block(block, group)
