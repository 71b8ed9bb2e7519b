"""world_size-2 gloo test (CPU) of the N>1 path: batch sharding + the single all-gather of detection blocks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from m3dssd_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    B, P = 6, 40
    all_dets = torch.randn(B, P, 14, generator=g)
    all_counts = torch.tensor([0, 3, 40, 7, 1, 12], dtype=torch.int32)
    for i in range(B):
        all_dets[i, all_counts[i]:] = 0
    lo, hi = mdist.shard_range(B, rank, world)
    assert torch.equal(mdist.shard_batch(all_dets, rank, world), all_dets[lo:hi])
    dets, counts = mdist.gather_detections(all_dets[lo:hi].clone(), all_counts[lo:hi].clone())
    ok = torch.equal(dets, all_dets) and torch.equal(counts, all_counts)
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_detections_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r))[0] == 1


def test_shard_range_and_single_process_passthrough():
    assert mdist.shard_range(256, 3, 8) == (96, 128)
    try:
        mdist.shard_range(10, 0, 4)
        assert False
    except ValueError:
        pass
    d, c = torch.zeros(2, 40, 14), torch.zeros(2, dtype=torch.int32)
    d2, c2 = mdist.gather_detections(d, c)
    assert d2 is d and c2 is c


def _worker8(rank, world, port, out_dir):
    """BASELINE.json configs[3]: bs 256 = 8 ranks x 32 images; every rank contributes its [32, 41, 14] block (40 kept rows + the
    count row, as m3d_select_post writes it) to ONE all_gather_into_tensor and reads [256, 40, 14] + [256] back."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    r, w, _ = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(7)
    B, P = 256, 40
    counts = torch.randint(0, P + 1, (B,), generator=g, dtype=torch.int32)
    counts[::37] = 0
    counts[5::41] = P
    full = torch.zeros(B, P + 1, 14)
    full[:, :P] = torch.randn(B, P, 14, generator=g)
    for i in range(B):
        full[i, int(counts[i]):P] = 0
    full[:, P, 0] = counts.float()
    lo, hi = mdist.shard_range(B, rank, world)
    assert (lo, hi) == (32 * rank, 32 * rank + 32)
    block = full[lo:hi].contiguous()
    assert tuple(block.shape) == (32, 41, 14) and block.numel() * 4 == 73472
    ok = True
    for _ in range(2):                           # second call: the cached receive buffer
        dets, cnt = mdist.gather_block(block)
        ok = ok and tuple(dets.shape) == (256, 40, 14) and tuple(cnt.shape) == (256,) and cnt.dtype == torch.int32
        ok = ok and torch.equal(dets, full[:, :P]) and torch.equal(cnt, counts)
    np.save(os.path.join(out_dir, "ok8_%d.npy" % rank), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_block_world8_configs3_block_gloo(tmp_path):
    world, port = 8, _free_port()
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(os.path.join(str(tmp_path), "ok8_%d.npy" % r))[0] == 1
