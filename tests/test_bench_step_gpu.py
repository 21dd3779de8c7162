"""-m gpu: ONE step of the workload bench.py times (BASELINE config 5's shard: 8 meshes of 2562 vertices / 5120 faces,
963-192-192-192 0N-GCN stack -> positions -> surface loss on the gt-index route -> backward) against a CPU restatement of
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


@pytest.mark.parametrize("act", ["relu", "elu"])
def test_one_bench_step_against_the_cpu_restatement(gpu, act):
    import bench
    activation = F.relu if act == "relu" else F.elu
    w = bench.Workload(gpu, 0, 8, activation=activation)
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
