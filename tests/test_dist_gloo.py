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
    # an optimiser step inside backward() would use the LOCAL gradients: refused while more than one rank is active
    from geometrics_amd import optim
    with pytest.raises(RuntimeError, match="before the all-reduce"):
        optim._InBackward(object()).__enter__()
    gdist.barrier()
    if rank == 0:
        out.put((bucket.flat[:bucket.numel].numpy().copy(), float(mean_loss)))     # by value: a tensor would be fetched from this process later
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
    flat = torch.from_numpy(flat)
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


# ------------------------------------------------- BatchNorm(verts) under data parallelism: global-batch statistics ----
def _bn_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    gdist.init_from_env(backend="gloo")
    from geometrics_amd import models
    torch.manual_seed(3)
    bn = models.VertexBatchNorm(11)
    assert bn.sync_across_ranks is False                    # opt-in: the default is local-shard statistics (DDP semantics)
    bn.sync_across_ranks = True
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(9)
    x_all, go_all, res_all = (torch.randn(6, 11, 5, generator=g) for _ in range(3))
    x_all = x_all + 30.0                                    # a mean far above the spread: E[x^2] - mean^2 would lose ~3 digits here
    first, count = gdist.shard_range(6, rank, world)
    x = x_all[first:first + count].clone().requires_grad_(True)
    res = res_all[first:first + count].clone().requires_grad_(True)
    y = bn(x, relu=True, residual=res)                      # CPU tensors + 2 ranks: only the synchronised path can serve this
    y.backward(go_all[first:first + count])
    gw, gb = bn.weight.grad.clone(), bn.bias.grad.clone()
    dist.all_reduce(gw), dist.all_reduce(gb)                # what the flat gradient bucket does for every parameter
    gdist.barrier()
    by_value = lambda t: t.detach().numpy().copy()          # a tensor would be fetched from this process after it has gone
    out.put((rank, by_value(y), by_value(x.grad), by_value(res.grad), by_value(gw), by_value(gb), by_value(bn.running_mean),
             by_value(bn.running_var), int(bn.state_dict()["num_batches_tracked"])))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_vertex_batchnorm_uses_global_batch_statistics_across_ranks():
    """Two ranks with 3 meshes each must normalise exactly as ONE process with all 6 (the reference is single-GPU:
    BatchNorm1d(verts) sees the whole batch, models.py:237-297): outputs, input / residual gradients, parameter
    gradients (summed over ranks) and running statistics against torch.nn.BatchNorm1d on the full batch."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, out), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([out.get(timeout=90) for _ in range(2)], key=lambda t: t[0])
    got = [tuple(torch.from_numpy(v) if hasattr(v, "dtype") else v for v in item) for item in got]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    torch.manual_seed(3)
    ref = torch.nn.BatchNorm1d(11)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(9)
    x_all, go_all, res_all = (torch.randn(6, 11, 5, generator=g) for _ in range(3))
    x_all = x_all + 30.0
    x, res = x_all.clone().requires_grad_(True), res_all.clone().requires_grad_(True)
    y = (res + torch.relu(ref(x))) * 0.5
    y.backward(go_all)
    y2 = torch.cat([got[0][1], got[1][1]])
    assert torch.allclose(y2, y.detach(), rtol=2e-5, atol=5e-6)
    assert torch.allclose(torch.cat([got[0][2], got[1][2]]), x.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(torch.cat([got[0][3], got[1][3]]), res.grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(got[0][4], ref.weight.grad, rtol=1e-4, atol=1e-6) and torch.allclose(got[0][5], ref.bias.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(got[0][6], ref.running_mean, rtol=1e-5, atol=1e-7) and torch.allclose(got[0][7], ref.running_var, rtol=1e-5, atol=1e-6)
    assert got[0][8] == 1
