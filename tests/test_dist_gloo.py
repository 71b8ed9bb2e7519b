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
