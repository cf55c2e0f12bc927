"""N>1 path on CPU: world_size-2 gloo processes exercise the weight broadcast and the
patch-shard partition/gather that bench.py and the multi-GPU driver rely on."""
import os
import socket

import numpy as np
import pytest

from mri_inr_amd import dist as mdist
from mri_inr_amd import synthetic as syn


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 400, 25600):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        mdist.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd0 = syn.make_state_dict(seed=7) if rank == 0 else None
        sd = mdist.broadcast_state_dict(sd0, src=0)
        ref = syn.make_state_dict(seed=7)
        same = set(sd) == set(ref) and all(np.array_equal(sd[k], ref[k]) and sd[k].shape == ref[k].shape for k in ref)
        # stand-in "model": a deterministic per-patch function, so the gather order is checkable
        tiles = np.random.default_rng(3).random((401, 32, 32), dtype=np.float32)
        fn = lambda t: t[:, 4:28, 4:28] * 2.0 + 1.0
        full = mdist.sharded_forward(fn, tiles)
        ok_gather = True
        if rank == 0:
            ok_gather = full.shape == (401, 24, 24) and np.array_equal(full, fn(tiles))
        else:
            ok_gather = full is None
        local, (lo, hi) = mdist.sharded_forward(fn, tiles, gather=False)
        q.put((rank, same, ok_gather, lo, hi, local.shape[0]))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharded_forward_world2():
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] and r[2] for r in res)
    assert (res[0][3], res[0][4], res[1][3], res[1][4]) == (0, 201, 201, 401)
    assert res[0][5] + res[1][5] == 401
