"""-m gpu: ONE step of the workload bench.py times (BASELINE config 5's shard: 8 meshes of 2562 vertices / 5120 faces,
963-192-192-192 0N-GCN stack -> positions -> surface loss (both Chamfer routes) -> backward) against a CPU restatement of
the reference formulation on the same parameters, features and -- through ops.scan_capture -- the very draws the step made:
layers.py:107-116 (dense row-normalised adjacency, torch.cat, bias, ReLU) in torch on the host, utils.py:441-502 through
oracle.ref_ops with the arg-min stages from the C oracle.  Positions and loss 1e-5 with the bench's ReLU.  The gradients of EVERY parameter tensor and
of the input features in max-norm 1e-4 of their scale with ELU in its place: a pre-activation that the two summation
orders put on different sides of zero (within an ulp of it) switches a whole unit's term under ReLU -- measured 2e-4 of
scale on the hidden weights and 4e-3 on single feature rows, the disagreement of ANY two fp32 evaluations of the kinked
function, the reference's CPU and CUDA paths included -- so the smooth activation is what the gradient chain (every
launch of the backward, the deferred reductions, the postponed products) is held against; under ReLU the last layer's
gradients, which no kink precedes, still have to agree (1e-5).  Per-element bounds belong to the isolated products and
the surface gradient: tests/test_dense_gpu.py, tests/test_ops_parity_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle  # noqa: F401
from oracle import ref_ops
from geometrics_amd import layers, ops

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4


@pytest.fixture
def route(request):
    """The Chamfer route of the step: "plain" (the headline since round 6: brute-force tiles, nothing precomputed from the
    ground truth) or "gt_index" (culled tiles on an index built outside the step)."""
    import bench
    old = bench.CULLED_CHAMFER
    bench.CULLED_CHAMFER = request.param == "gt_index"
    yield request.param
    bench.CULLED_CHAMFER = old


@pytest.mark.parametrize("act,route", [("relu", "plain"), ("relu", "gt_index"), ("elu", "plain")], indirect=["route"])
def test_one_bench_step_against_the_cpu_restatement(gpu, act, route):
    import bench
    activation = F.relu if act == "relu" else F.elu
    w = bench.Workload(gpu, 0, 8, activation=activation)
    assert (w.gt_index is not None) == (route == "gt_index")
    seen = {}
    ops.scan_capture = seen
    try:
        w.opt.zero_grad()
        w.feat.grad = None
        with layers.deferred_parameter_gradients():
            pos = w.positions()
            loss = bench.utils.batch_point_to_surface(pos, w.info, w.gt, num=bench.S_PTS, gt_index=w.gt_index)
            loss.backward()
    finally:
        ops.scan_capture = None
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu()
    adj = cpu(w.info["adj"])
    feat = cpu(w.feat).requires_grad_(True)
    params = [(cpu(l.weight1).requires_grad_(True), cpu(l.bias).requires_grad_(True)) for l in w.stack]
    h = feat
    for wt, b in params:
        h = ref_ops.zero_n_layer(h, adj, wt, b, 3, activation)
    pos_c = cpu(w.base) + 0.01 * h[..., :3]
    assert float((cpu(pos) - pos_c).abs().max()) <= 1e-5 * float(pos_c.abs().max())
    ref = ref_ops.point_to_surface(pos_c, cpu(w.faces), cpu(w.gt), cpu(seen["choices"]), cpu(seen["u"]), cpu(seen["v"]))
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    got = [w.feat.grad] + [g for l in w.stack for g in (l.weight1.grad, l.bias.grad)]
    want = [feat.grad] + [g for wt, b in params for g in (wt.grad, b.grad)]
    names = ["features"] + ["%s of layer %d" % (n, i) for i in range(3) for n in ("weight1", "bias")]
    errs = {name: float((cpu(g).view_as(r) - r).abs().max()) / float(r.abs().max()) for name, g, r in zip(names, got, want)}
    for name, err in errs.items():
        if act == "relu" and "layer 2" not in name:
            continue
        tol = GRAD_TOL if act == "elu" else 1e-5
        assert err <= tol, "%s: gradient differs from the CPU restatement by %.2e of its scale (all: %s)" % (name, err, errs)


def test_a_replay_of_the_captured_step_on_other_ground_truth_clouds_matches_the_oracle(gpu):
    """The reference hands the step a FRESH ground-truth subset every fetch (utils.py:180-182: shuffle, take sample_num), so
    everything that depends on the ground truth has to live inside the step.  The headline step (bench.Workload, one HIP
    graph) reads its ground-truth tensor as an ordinary input: overwrite it with other clouds, replay the SAME graph, and
    the arg-min stages of that replay must be the C oracle's on the new clouds bit for bit (indices, region codes, squared
    distances, sampled points) and its loss the CPU restatement's (1e-5) -- at the positions and draws of that very replay."""
    import bench
    from geometrics_amd import meshgen
    assert not bench.CULLED_CHAMFER, "the headline default: no index of the ground truth outside the step"
    w = bench.Workload(gpu, 0, 8)
    assert w.gt_index is None
    seen = {}
    ops.scan_capture = seen
    try:
        w.capture()
    finally:
        ops.scan_capture = None
    assert w.graphs is not None and len(w.graphs) == 1
    w.run()
    torch.cuda.synchronize()
    first_loss = w.mean_loss()
    other = torch.from_numpy(np.ascontiguousarray(meshgen.gt_cloud(8, bench.G_PTS, first=900))).to(gpu)
    assert not torch.equal(other, w.gt)
    w.gt.copy_(other)                                  # what a data loader does between two steps
    with torch.no_grad():
        pos = w.positions().detach().clone()           # the positions the replay is about to compute (parameters as they are now)
    w.run()
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu().numpy()
    verts, gt, faces = cpu(pos), cpu(other), cpu(w.faces)
    ch, u, v = cpu(seen["choices"]), cpu(seen["u"]), cpu(seen["v"])
    pred = ref_ops.sample_points(*(torch.from_numpy(a) for a in (verts, faces, ch, u, v))).numpy()
    d_gt, i_gt, d_pred, i_pred = oracle.chamfer_nn(gt, pred)
    t_d, t_opt, t_idx = oracle.tri_scan_indexed(gt, verts, faces)
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    assert np.array_equal(bits(cpu(seen["points"])), bits(pred))
    assert np.array_equal(cpu(seen["idx_gt"]), i_gt) and np.array_equal(cpu(seen["idx_pred"]), i_pred)
    assert np.array_equal(cpu(seen["tri_index"]), t_idx) and np.array_equal(cpu(seen["tri_option"]), t_opt)
    assert np.array_equal(bits(cpu(seen["sq_pred"])), bits(d_pred)) and np.array_equal(bits(cpu(seen["tri_dist"])), bits(t_d))
    ref = ref_ops.point_to_surface(torch.from_numpy(verts), torch.from_numpy(faces), torch.from_numpy(gt), torch.from_numpy(ch),
                                   torch.from_numpy(u), torch.from_numpy(v))
    loss = w.mean_loss()
    assert abs(loss - float(ref)) <= 1e-5 * abs(float(ref)), (loss, float(ref))
    assert abs(loss - first_loss) > 1e-3 * abs(first_loss), "the replay did not see the new clouds"
