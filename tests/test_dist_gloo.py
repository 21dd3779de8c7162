"""CPU: the N>1 logic (sharding, flat gradient bucket, loss all-reduce) with world_size-2 gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from geometrics_amd import dist as gdist


def test_shard_ranges_partition_the_batch():
    for total, world in ((64, 8), (16, 8), (10, 4), (3, 8), (8, 1)):
        spans = [gdist.shard_range(total, r, world) for r in range(world)]
        assert sum(c for _, c in spans) == total
        pos = 0
        for first, count in spans:
            assert first == pos
            pos += count
    assert gdist.shard_range(64, 3, 8) == (24, 8)          # config 5: 8 contiguous groups of 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                                   # replicated parameters
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    bucket = gdist.GradBucket(model.parameters(), extra=2)
    first, count = gdist.shard_range(8, rank, world)
    g = torch.Generator().manual_seed(123)
    data = torch.randn(8, 6, generator=g)                  # the "meshes": every rank sees the same global batch
    loss = model(data[first:first + count]).pow(2).sum(1).mean()   # mean over the local shard
    loss.backward()
    # one cat + ONE all-reduce(SUM) carries the gradients and the [loss_sum, count] tail
    views = bucket.pack_all_reduce(loss.detach() * count, torch.tensor(float(count)))
    assert [tuple(v.shape) for v in views] == [tuple(p.shape) for p in model.parameters()]
    assert all(v.data_ptr() >= bucket.flat.data_ptr() for v in views)
    mean_loss = bucket.extra[0] / bucket.extra[1]
    assert float(bucket.extra[1]) == 8.0
    bucket.flat[:bucket.numel].div_(world)
    gdist.barrier()
    if rank == 0:
        out.put((bucket.flat[:bucket.numel].clone(), float(mean_loss)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_bucket_and_loss_match_single_process():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    flat, mean_loss = out.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    g = torch.Generator().manual_seed(123)
    data = torch.randn(8, 6, generator=g)
    loss = model(data).pow(2).sum(1).mean()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-6)          # equal shards: mean of means is exact
    assert abs(mean_loss - float(loss)) < 1e-5


def test_single_process_helpers_are_noops():
    t = torch.ones(3)
    assert gdist.all_reduce_sum_(t) is t
    assert float(gdist.global_mean_loss(torch.tensor(6.0), 3)) == 2.0
    gdist.barrier()
