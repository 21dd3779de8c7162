"""auto_encoder.py encoder step on a ragged batch: the reference formulation (python loop over meshes, dense
adjacency mm per layer, torch ops on the same GPU) against MeshEncoder.encode_batch (block-diagonal CSR, one GEMM +
one aggregation launch per layer, segmented max).  fwd+bwd, 16 meshes of mixed size.

    python tools/time_encoder.py
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import meshgen, models, ragged, utils  # noqa: E402


def reference_formulation(params, names, verts, adjs):
    lat = []
    for v, adj in zip(verts, adjs):                     # auto_encoder.py:71-76
        x = v
        for n in names:                                 # layers.py:34-41
            s = torch.mm(x, params[n + ".weight"])
            k = s.shape[1] // 10
            x = F.elu(torch.cat((torch.mm(adj, s[:, :k]), s[:, k:]), dim=1) + params[n + ".bias"])
        s = torch.mm(x, params["reduce.weight_Ws.0"])   # layers.py:61-79
        k = s.shape[1] // 10
        s = F.elu(torch.cat((torch.mm(adj, s[:, :k]), s[:, k:]), dim=1) + params["reduce.weight_Bs.0"])
        lat.append(torch.max(s, dim=0)[0])
    return torch.stack(lat)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    levels = [2, 3, 4, 3, 3, 4, 2, 3, 4, 3, 3, 2, 4, 3, 3, 4]          # 162 / 642 / 2562 vertices
    verts, faces = [], []
    for i, lv in enumerate(levels):
        V, Fc = meshgen.icosphere(lv)
        verts.append(torch.from_numpy(meshgen.jittered_batch(V, 1, first=i)[0]).to(dev))
        faces.append(torch.from_numpy(Fc).to(dev))
    adjs = [utils.normalize_adj(utils.calc_adj(f)) for f in faces]
    enc = models.MeshEncoder(50).to(dev)
    with torch.no_grad():                                                # keep 17 ELU layers in a sane range
        for p in enc.parameters():
            if p.dim() == 2:
                p.mul_(1 / 3.0)
    names = [n for n, _, _ in enc._WIDTHS]
    params = dict(enc.named_parameters())

    def ref_step():
        enc.zero_grad(set_to_none=True)
        reference_formulation(params, names, verts, adjs).square().mean().backward()

    batch = ragged.RaggedMeshBatch.from_faces(verts, faces)

    def ragged_step():
        enc.zero_grad(set_to_none=True)
        enc.encode_batch(batch).square().mean().backward()

    def ragged_step_with_build():
        enc.zero_grad(set_to_none=True)
        enc.encode_batch(ragged.RaggedMeshBatch.from_faces(verts, faces)).square().mean().backward()

    graphed = enc.graphed_encode(batch)

    def graphed_step():
        enc.zero_grad(set_to_none=True)
        graphed(batch.verts).square().mean().backward()

    with torch.no_grad():
        a = reference_formulation(params, names, verts, adjs)
        b = enc.encode_batch(batch)
    print("meshes %d, vertices %d, max |latent diff| %.2e (max |latent| %.2e)"
          % (len(verts), batch.total, float((a - b).abs().max()), float(a.abs().max())))
    ragged_step()
    eager_grads = [p.grad.clone() for p in enc.parameters()]
    graphed_step()
    worst = max(float((p.grad - g).abs().max()) for p, g in zip(enc.parameters(), eager_grads))
    print("graphed vs eager: max |parameter gradient diff| %.2e" % worst)
    t_ref, t_new, t_build, t_graph = timed(ref_step), timed(ragged_step), timed(ragged_step_with_build), timed(graphed_step)
    print("ragged batch, forward + backward as two HIP graphs %8.3f ms  (%.1fx)" % (t_graph, t_ref / t_graph))
    print("reference formulation (per-mesh loop, dense adj)  %8.3f ms" % t_ref)
    print("ragged batch, prebuilt CSR                        %8.3f ms  (%.1fx)" % (t_new, t_ref / t_new))
    print("ragged batch incl. CSR assembly from faces        %8.3f ms  (%.1fx)" % (t_build, t_ref / t_build))


if __name__ == "__main__":
    main()
