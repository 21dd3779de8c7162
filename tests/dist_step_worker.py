"""Workers of tests/test_dist_step_gpu.py (spawned; must be importable): run bench.Workload -- the REAL step, HIP
graphs included -- and report parameters, the averaged gradient Adam consumed and the mean loss of every step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _flat(tensors):
    import torch
    return torch.cat([t.detach().reshape(-1) for t in tensors]).cpu().numpy()


def run(rank, world, port, total_meshes, steps, out, activation="relu", lr=1e-4):
    """One rank of a `world`-rank job over `total_meshes` meshes, exactly as bench.py runs it (activation "elu": the
    smooth-activation variant of the same step, for the comparison that ReLU's unit flips would blur; lr = 0 keeps the
    parameters where they are, so that every step's gradient can be compared across job shapes)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), GEOM_DIST_BACKEND="gloo")     # both ranks share cuda:0; gloo carries the all-reduce
    import torch
    import bench
    from geometrics_amd import dist as gdist, gemm_tuning
    r, w, _ = gdist.init_from_env()
    assert (r, w) == (rank, world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    gemm_tuning.enable()
    first, count = gdist.shard_range(total_meshes, rank, world)
    import torch.nn.functional as F
    wl = bench.Workload(dev, first, count, activation={"relu": F.relu, "elu": F.elu}[activation], lr=lr)
    wl.capture()                       # N=1: one graph; N>1: graph A / all-reduce beside graph B, as bench.py runs it
    losses = []
    for _ in range(steps):
        wl.run()
        torch.cuda.synchronize()
        losses.append(wl.mean_loss())
    wl.finish()                        # N > 1: the Adam step of the last iteration is still owed (it opens the next one)
    if world > 1:                      # the reduced bucket of the last step, scaled as Adam consumed it
        grads = (wl.bucket.flat[:wl.bucket.numel] / world).cpu().numpy()
    else:
        grads = _flat([p.grad for p in wl.stack.parameters()])
    gdist.barrier()
    if rank == 0:
        out.put({"params": _flat(wl.stack.parameters()), "grads": grads, "losses": losses,
                 "steps_taken": wl.opt.step_count})
    if world > 1:
        torch.distributed.destroy_process_group()


def run_rccl_single(port, total_meshes, steps, out, sequence="two_graphs"):
    """ONE rank, backend "nccl" (= RCCL) on cuda:0, bench.Workload forced onto its N > 1 sequence: graph A (Adam of the
    previous step, forward, backward, reduction launch -> bucket) -> RCCL all-reduce of the flat bucket from the side
    stream, beside graph B (the first layer's postponed input gradient)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch
    import bench
    from geometrics_amd import gemm_tuning
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    assert torch.distributed.get_backend() == "nccl"
    gemm_tuning.enable()
    wl = bench.Workload(dev, 0, total_meshes, force_dp=True, dp_sequence=sequence)
    assert wl.bucket is not None and wl.bucket.force
    wl.capture()
    assert wl.dp_sequence == sequence and wl.pending                  # (a captured sequence that fell back would say so here)
    if sequence == "two_graphs":
        assert len(wl.graphs) == 2 and wl.graphs[1] is not None       # graph A opens with the Adam step still owed; B = the postponed product
    else:
        assert len(wl.graphs) == 1                                    # the collective is a node of the step's one graph
    losses = []
    for _ in range(steps):
        wl.run()
        torch.cuda.synchronize()
        losses.append(wl.mean_loss())
    wl.finish()
    out.put({"overlap": not wl.packed_late, "params": _flat(wl.stack.parameters()), "grads": wl.bucket.flat[:wl.bucket.numel].cpu().numpy(),
             "losses": losses, "steps_taken": wl.opt.step_count})
    torch.distributed.destroy_process_group()


def run_serial(shards, total_meshes, steps, warm, out):
    """ONE process emulating `shards` data-parallel ranks in turn: every shard is a bench.Workload of its own (same
    kernels, same GEMM shapes as a real rank), gradients are summed in rank order, and every replica applies Adam with
    grad_scale = 1/shards.  What a correct N-rank job must reproduce bit for bit."""
    import torch
    import bench
    from geometrics_amd import dist as gdist, gemm_tuning, ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    gemm_tuning.enable()
    wls = [bench.Workload(dev, *gdist.shard_range(total_meshes, r, shards)) for r in range(shards)]
    losses = []
    for it in range(warm + steps):
        for wl in wls:
            ops.set_rng_state(wl.rng)
            wl.forward_backward()
        summed = [sum(g[1:], g[0]) for g in zip(*[[p.grad for p in wl.stack.parameters()] for wl in wls])]
        tail = sum((wl.loss.detach() * wl.batch for wl in wls[1:]), wls[0].loss.detach() * wls[0].batch)
        for wl in wls:
            wl.opt.step(summed, grad_scale=1.0 / shards)
        torch.cuda.synchronize()
        if it >= warm:
            losses.append(float(tail) / sum(wl.batch for wl in wls))
    out.put({"params": _flat(wls[0].stack.parameters()), "grads": _flat(summed) / shards, "losses": losses,
             "steps_taken": wls[0].opt.step_count})


# ---- the deformation block under data parallelism with GLOBAL-batch BatchNorm statistics (VertexBatchNorm.sync_across_ranks) ----
def _block_case(total, nv_level=2, width=40):
    import torch
    from geometrics_amd import meshgen
    V, Fc = meshgen.icosphere(nv_level)
    g = torch.Generator().manual_seed(77)
    feats = torch.randn(total, V.shape[0], 3, generator=g)
    pooled = torch.randn(total, V.shape[0], 189 + width, generator=g) + 3.0      # a mean well away from zero
    g_f = torch.randn(total, V.shape[0], 192, generator=g)
    g_c = torch.randn(total, V.shape[0], 3, generator=g)
    return V, Fc, feats, pooled, g_f, g_c


def _block_pass(block, adj, feats, pooled, g_f, g_c, scale):
    import torch
    f, p = feats.clone().requires_grad_(True), pooled.clone().requires_grad_(True)
    out_f, coords = block(f, p, adj)
    (((out_f * g_f).sum() + (coords * g_c).sum()) * scale).backward()
    return out_f.detach(), coords.detach(), f.grad, p.grad


def run_sync_bn_block(rank, world, port, total, out):
    """`world` ranks (gloo, sharing cuda:0), each with total / world meshes, VertexBatchNorm.sync_across_ranks = True; rank 0
    reports outputs / input gradients of all ranks (gathered) + its parameter gradients all-reduced, and the running statistics."""
    import os
    import torch
    import torch.distributed as dist
    from geometrics_amd import models, utils
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    V, Fc, feats, pooled, g_f, g_c = _block_case(total)
    adj = utils.adj_init(torch.from_numpy(Fc).to(dev))["adj"]
    torch.manual_seed(5)
    block = models.BatchMeshDeformationBlock(feats.shape[-1] + pooled.shape[-1], V.shape[0]).to(dev).train()
    models.VertexBatchNorm.sync_across_ranks = True
    per = total // world
    sl = slice(rank * per, (rank + 1) * per)
    res = _block_pass(block, adj, *(t[sl].to(dev) for t in (feats, pooled, g_f, g_c)), 1.0)
    grads = torch.cat([p.grad.flatten() for n, p in block.named_parameters() if p.grad is not None])
    if world > 1:
        dist.all_reduce(grads)                       # the job's gradient all-reduce (sum over the shards)
        gathered = []
        for t in res:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t.contiguous())
            gathered.append(torch.cat(parts))
        res = gathered
    if rank == 0:
        stats = torch.cat([torch.cat((getattr(block, "bn%d" % i).running_mean, getattr(block, "bn%d" % i).running_var)) for i in range(1, 14)])
        out.put({"res": [t.cpu().numpy() for t in res], "grads": grads.cpu().numpy(), "stats": stats.cpu().numpy()})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_sync_bn_capture(port, total, out):
    """1-rank RCCL group on cuda:0 with the synchronised route forced on: the block's forward + backward (two small all-reduces
    per layer inside) captured into ONE HIP graph and replayed, against the same pass run eagerly."""
    import os
    import torch
    import torch.distributed as dist
    from geometrics_amd import models, utils
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    V, Fc, feats, pooled, g_f, g_c = _block_case(total)
    adj = utils.adj_init(torch.from_numpy(Fc).to(dev))["adj"]
    torch.manual_seed(5)
    block = models.BatchMeshDeformationBlock(feats.shape[-1] + pooled.shape[-1], V.shape[0]).to(dev).train()
    models.VertexBatchNorm.sync_across_ranks = True
    models.VertexBatchNorm._sync_single_rank_groups = True
    args = [t.to(dev) for t in (feats, pooled, g_f, g_c)]
    f, p = args[0].clone().requires_grad_(True), args[1].clone().requires_grad_(True)
    assert block.bn1._synchronised()
    keep = {}

    def step():
        for q in block.parameters():
            q.grad = None
        f.grad = p.grad = None
        out_f, coords = block(f, p, adj)
        ((out_f * args[2]).sum() + (coords * args[3]).sum()).backward()
        keep["out"] = (out_f.detach(), coords.detach())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = [t.clone() for t in keep["out"]] + [f.grad.clone(), p.grad.clone()] + [q.grad.clone() for q in block.parameters() if q.grad is not None]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
        step()
    g.replay()
    torch.cuda.synchronize()
    replay = list(keep["out"]) + [f.grad, p.grad] + [q.grad for q in block.parameters() if q.grad is not None]
    same = all(torch.equal(a, b) for a, b in zip(eager, replay))
    out.put({"same": bool(same), "count": len(eager), "finite": bool(all(torch.isfinite(t).all() for t in replay))})
    dist.destroy_process_group()
