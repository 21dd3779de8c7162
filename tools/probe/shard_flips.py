"""How many surface samples land on another face when the SAME 16 meshes are stepped as one 16-mesh batch and as two
8-mesh shards (other GEMM kernel selections -> positions differ by an ulp here and there -> the face-area CDF moves by
round-off), and how far the parameter gradients move with them.  ELU, lr = 0, one eager step each.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                   # noqa: E402
import torch.nn.functional as F                # noqa: E402
import bench                                   # noqa: E402
from geometrics_amd import gemm_tuning, ops    # noqa: E402

dev = torch.device("cuda:0")
gemm_tuning.enable()
seen = {}
orig = ops.draw_samples


def spy(*a, **k):
    out = orig(*a, **k)
    seen.setdefault("draws", []).append((out[0].clone(), a[0].detach().clone()))
    return out


ops.draw_samples = spy
for culled in (True, False):
    bench.CULLED_CHAMFER = culled
    seen.clear()
    whole = bench.Workload(dev, 0, 16, activation=F.elu, lr=0.0)
    parts = [bench.Workload(dev, 0, 8, activation=F.elu, lr=0.0), bench.Workload(dev, 8, 8, activation=F.elu, lr=0.0)]
    ops.set_rng_state(whole.rng)              # the sampler's stream state is per device: every workload brings its own
    whole.forward_backward()
    gw = torch.cat([p.grad.reshape(-1) for p in whole.stack.parameters()])
    for w in parts:
        ops.set_rng_state(w.rng)
        w.forward_backward()
    gp = sum(torch.cat([p.grad.reshape(-1) for p in w.stack.parameters()]) for w in parts) / 2
    (cw, pw), (c0, p0), (c1, p1) = seen["draws"]
    cp, pp = torch.cat([c0, c1]), torch.cat([p0, p1])
    print("culled route %-5s: positions differing %d of %d (max |diff| %.2e); samples on another face %d of %d; "
          "parameter gradient moves by %.2e of its scale; losses %.6f / %.6f"
          % (culled, int((pw != pp).sum()), pw.numel(), float((pw - pp).abs().max()), int((cw != cp).sum()), cw.numel(),
             float((gw - gp).abs().max() / gw.abs().max()), float(whole.loss), float((parts[0].loss + parts[1].loss) / 2)))
