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
