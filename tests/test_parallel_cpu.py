"""world_size-2 gloo tests of the multi-GPU host logic (sharding, noise slicing, image gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from maskbit_amd.parallel import gather_images, shard_range, slice_noise


def test_shard_range_partitions_the_batch():
    for B in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(r, world, B)
                cover += list(range(lo, hi))
                assert hi - lo in (B // world, B // world + 1)
            assert cover == list(range(B))
    with pytest.raises(ValueError):
        shard_range(2, 2, 8)


def test_slice_noise_is_the_batch_rows():
    steps, B, n, m, C = 3, 6, 4, 2, 5
    e = torch.arange(steps * B * n * m * C, dtype=torch.float32).reshape(steps, B * n * m, C)
    c = torch.arange(steps * B * n * m, dtype=torch.float32).reshape(steps, B, n, m)
    es, cs = slice_noise(e, c, 2, 5, n * m)
    assert torch.equal(es, e.reshape(steps, B, n * m, C)[:, 2:5].reshape(steps, 3 * n * m, C))
    assert torch.equal(cs, c[:, 2:5])
    parts = [slice_noise(e, c, *shard_range(r, 3, B), n * m) for r in range(3)]
    assert torch.equal(torch.cat([p[0] for p in parts], 1), e) and torch.equal(torch.cat([p[1] for p in parts], 1), c)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ragged, q, equal=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 5 if ragged else 6
        lo, hi = shard_range(rank, world, B)
        full = (torch.arange(B * 4 * 4 * 3) % 251).to(torch.uint8).reshape(B, 4, 4, 3)
        import unittest.mock as um
        calls = []
        real = dist.all_gather
        with um.patch.object(dist, "all_gather", lambda *a, **k: (calls.append(1), real(*a, **k))[1]):
            got = gather_images(full[lo:hi].clone(), equal=equal)
        # equal=True: exactly one collective (all_gather_into_tensor), no size exchange
        q.put((rank, bool(torch.equal(got, full)) and (len(calls) == 0 if equal else len(calls) >= 1), tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ragged,equal", [(False, None), (True, None), (False, True)])
def test_gather_images_gloo_world2(ragged, equal):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ragged, q, equal)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
