"""CPU restatement (torch, any float dtype) of the differentiable python stages of the
GEOMetrics hot path -- TEST INFRASTRUCTURE ONLY, never imported by geometrics_amd.

Each function states the reference lines it follows.  They are pinned against fixtures
emitted by the imported reference itself (tests/golden/make_golden.py ->
tests/test_oracle_pin.py) and then serve as the checker at sizes where no fixture is
stored (2562-vertex meshes), with the arg-min stages taken from the C oracle.
"""
import numpy as np
import torch

import oracle


def sample_points(verts, faces, choices, u, v):
    """utils.py:615-631: gather the three corners of each chosen face, then
    ((1-u)*x + (u*(1-v))*y) + (u*v)*z.  choices [B,S] face ids; u is already sqrt'ed."""
    b = verts.shape[0]
    sel = faces[choices.reshape(-1)].view(b, -1, 3)                      # [B,S,3] vertex ids
    x, y, z = (torch.gather(verts, 1, sel[..., k:k + 1].expand(-1, -1, 3)) for k in range(3))
    u = u.unsqueeze(-1)
    v = v.unsqueeze(-1)
    return (1 - u) * x + (u * (1 - v)) * y + u * v * z


def face_areas(verts, faces):
    """utils.py:596-602 (un-normalised)."""
    x = verts[:, faces[:, 0]] - verts[:, faces[:, 1]]
    y = verts[:, faces[:, 1]] - verts[:, faces[:, 2]]
    a = (x[..., 1] * y[..., 2] - x[..., 2] * y[..., 1]) ** 2
    b = (x[..., 2] * y[..., 0] - x[..., 0] * y[..., 2]) ** 2
    c = (x[..., 0] * y[..., 1] - x[..., 1] * y[..., 0]) ** 2
    return torch.sqrt(a + b + c) / 2


def _nn(gt, pred, flags=0):
    """flags = oracle.FLAG_REF_TAIL_TRUNC: the shipped CUDA kernel's tile structure (chamfer_distance.cu:6-55) instead of the
    full sequential scan."""
    _, idx_p, _, idx_g = oracle.chamfer_nn(gt.detach().float().numpy(), pred.detach().float().numpy(), flags)
    return torch.from_numpy(idx_p).long(), torch.from_numpy(idx_g).long()


def _f1(pred_counters, gt_counters, pred, gt, num):
    """utils.py:424-436 / 489-500."""
    to_pred = torch.sqrt(((.57 * pred_counters - .57 * gt) ** 2).sum(-1))
    to_gt = torch.sqrt(((.57 * gt_counters - .57 * pred) ** 2).sum(-1))
    score = 0.0
    for i in range(to_pred.shape[0]):
        recall = float((to_pred[i] <= 1e-2).sum()) / float(num)
        precision = float((to_gt[i] <= 1e-2).sum()) / float(num)
        score += 2 * (precision * recall) / (precision + recall + 1e-8)
    return score / to_pred.shape[0]


def point_to_point(verts, faces, gt, choices, u, v, f1=False, nn_flags=0):
    """utils.py:393-438 with the draws replayed."""
    pred = sample_points(verts, faces, choices, u, v)
    idx_p, idx_g = _nn(gt, pred, nn_flags)
    pred_counters = torch.gather(pred, 1, idx_p.unsqueeze(-1).expand(-1, -1, 3))
    gt_counters = torch.gather(gt, 1, idx_g.unsqueeze(-1).expand(-1, -1, 3))
    dist_1 = ((gt_counters - pred) ** 2).sum(-1).mean()
    dist_2 = ((pred_counters - gt) ** 2).sum(-1).mean()
    loss = (dist_1 + dist_2) * 3000
    return (loss, _f1(pred_counters, gt_counters, pred, gt, choices.shape[1])) if f1 else loss


def closest_point(p, a, b, c, option):
    """utils.py:506-548: candidate selected by option (1,2,3 corners; 4,5,6 edge points with the
    CORRECT deltas; 0 plane projection).  All [N,3]; option [N]."""
    def proj(org, delta):
        return ((p - org) * delta).sum(-1) / (delta ** 2).sum(-1)

    uab, ubc, uca = proj(a, b - a), proj(b, c - b), proj(c, a - c)
    n = torch.cross(a - b, a - c, dim=-1)
    n = n / torch.sqrt((n ** 2).sum(-1)).unsqueeze(-1)
    plane = p - ((p - a) * n).sum(-1, keepdim=True) * n
    cands = [plane, a, b, c, a + uab.unsqueeze(-1) * (b - a), b + ubc.unsqueeze(-1) * (c - b),
             c + uca.unsqueeze(-1) * (a - c)]
    out = torch.zeros_like(p)
    for code, cand in enumerate(cands):
        out = torch.where((option == code).unsqueeze(-1), cand, out)
    return out


def point_to_line(p, a, b, c, option):
    """utils.py:549: mean squared distance to the selected candidate."""
    return ((closest_point(p, a, b, c, option) - p) ** 2).sum(-1).mean()


def point_to_surface(verts, faces, gt, choices, u, v, f1=False, tri_flags=0, nn_flags=0):
    """utils.py:441-502 with the draws replayed; the tri scan is the C oracle."""
    pred = sample_points(verts, faces, choices, u, v)
    idx_p, idx_g = _nn(gt, pred, nn_flags)
    pred_counters = torch.gather(pred, 1, idx_p.unsqueeze(-1).expand(-1, -1, 3))
    gt_counters = torch.gather(gt, 1, idx_g.unsqueeze(-1).expand(-1, -1, 3))
    dist_1 = ((gt_counters - pred) ** 2).sum(-1).mean()
    _, opt, idx = oracle.tri_scan_indexed(gt.detach().float().numpy(), verts.detach().float().numpy(),
                                          faces.numpy(), tri_flags)
    idx = torch.from_numpy(idx).long()
    corners = [torch.gather(verts[:, faces[:, k]], 1, idx.unsqueeze(-1).expand(-1, -1, 3)).reshape(-1, 3)
               for k in range(3)]
    dist_2 = point_to_line(gt.reshape(-1, 3), *corners, torch.from_numpy(opt).reshape(-1))
    loss = (dist_1 + dist_2) * 3000
    return (loss, _f1(pred_counters, gt_counters, pred, gt, choices.shape[1])) if f1 else loss


def calc_adj(faces):
    """utils.py:115-131."""
    n = int(faces.max()) + 1
    adj = torch.eye(n)
    for i, j in ((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)):
        adj[faces[:, i], faces[:, j]] = 1
    return adj


def normalize_adj(mx):
    """utils.py:96-101."""
    r_inv = 1.0 / mx.sum(1)
    r_inv[r_inv != r_inv] = 0.0
    return torch.diag(r_inv) @ mx


def zero_n_layer(x, adj, weight, bias, split, activation):
    """layers.py:34-41 / 107-116 / 143-152: support = x W; first C//split columns multiplied by
    the dense adjacency; concat; + bias; activation."""
    if weight.dim() == 3:
        weight = weight[0]
    support = x @ weight
    k = support.shape[-1] // split
    out = torch.cat((adj @ support[..., :k], support[..., k:]), dim=-1)
    if bias is not None:
        out = out + bias
    return activation(out)


def gcn_max(x, adj, weight, bias, activation, batched):
    """layers.py:61-79 (max of the activation) / 175-189 (max of the PRE-activation)."""
    v = zero_n_layer(x, adj, weight, bias, 10, lambda t: t)
    return torch.max(v, dim=1)[0] if batched else torch.max(activation(v), dim=0)[0]


def as_np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def lap_info(positions, adj_orig):
    """utils.py:654-662: dense neighbour sum minus self, divided by (degree), subtracted from positions."""
    neighbour_sum = torch.matmul(adj_orig, positions) - positions
    scaler = (1.0 / (adj_orig.sum(1) - 1)).view(-1, 1)
    return positions - neighbour_sum * scaler


def calc_edge(verts, faces):
    """utils.py:636-651."""
    p1, p2, p3 = (verts[:, faces[:, k]] for k in range(3))
    return (((p2 - p1) ** 2).sum(-1).mean() + ((p3 - p1) ** 2).sum(-1).mean() + ((p2 - p3) ** 2).sum(-1).mean()) / 3.0


ENCODER_LAYERS = ("h1", "h21", "h22", "h23", "h24", "h3", "h4", "h41", "h5", "h6", "h7", "h8", "h81", "h9", "h10", "h11")


def mesh_encoder(params, positions, adj):
    """models.py:324-348 for one mesh: 16 ELU 0N-GCN layers (split 10), then GCNMax (layers.py:61-79).
    params: {name: tensor} with the reference's state_dict keys."""
    elu = torch.nn.functional.elu
    x = positions
    for name in ENCODER_LAYERS:
        x = zero_n_layer(x, adj, params[name + ".weight"], params[name + ".bias"], 10, elu)
    return gcn_max(x, adj, params["reduce.weight_Ws.0"], params["reduce.weight_Bs.0"], elu, batched=False)


def segment_max(x, sizes):
    """Per-mesh column max of rows concatenated along dim 0 (layers.py:78 applied mesh by mesh)."""
    return torch.stack([part.max(dim=0)[0] for part in torch.split(x, list(sizes), dim=0)])
