"""A 0N-GCN layer boundary in one launch (csrc/zn_stack.hip) -- thin host wrappers over `geom_zn_layer_*_f32`.

The reference runs `support = matmul(input, weight1)`, `matmul(adj, support[..., :k])`, `cat`, `+ bias`, activation
(layers.py:107-116) layer by layer; between two consecutive layers the aggregation of the first is the operand load of the
second's product (forward), and the aggregation backward of a layer is the operand load of its input-gradient product.
Shapes: 192-wide layers, k = 64, a neighbour table of width 8 without long rows -- the hidden layers of the BASELINE stack;
`supported()` says whether a layer pair qualifies, everything else takes the two separate operators.
"""
import os

import torch

from . import _lib


def supported(csr, c, k, n_out):
    return bool(csr.ell_w == 8 and csr.over is None and csr.over_t is None and c == 192 and k == 64
                and 0 < n_out <= 192 and n_out % 12 == 0)


# tests / tools: {"fwd": bool, "bwd": bool} overrides `plan`
force = None


def plan(rows):
    """Which boundaries take the single launch.  Launch against launch in a loop (tools/time_fused_layer.py,
    profiles/r05_fused_boundary.txt) the boundary is ahead of aggregation + library product from ~6 meshes of 2562 vertices
    forward and ~10 backward; INSIDE the step (tools/probe/fused_plan_sweep.sh: the whole step replayed with the plan off /
    forward / both) it is not -- its weight slice and tables arrive cold and its 27 us there lose to 10 + 14.4 us at the
    8-mesh shard (+3 us per boundary), it breaks even at 12-16 meshes, and wins from 32 (1512 vs 1567 us per step; at 64
    meshes, both directions: 2777 vs 2996 us, -7.3 %).  A finer sweep (20 / 24 / 28 / 32 / 40 / 48 meshes, both directions against
    none): 1020 vs 1110, 1209 vs 1212, 1370 vs 1411, 1535 vs 1584, 1883 vs 2067, 2282 vs 2466 us per step -- ahead or level from
    20 meshes on.  The threshold follows the in-step measurement: both directions from 50 000 rows."""
    if force is not None:
        return dict(force)
    env = os.environ.get("GEOM_FUSED_PLAN")       # tools: "off" | "fwd" | "all"
    if env:
        return {"fwd": env in ("fwd", "all"), "bwd": env == "all"}
    return {"fwd": rows >= 50000, "bwd": rows >= 50000}


def partial_rows(b, nv):
    return int(_lib.lib().geom_zn_layer_partial_rows(b, nv))


def layer_forward(s_prev, bias_prev, csr, k, act, w, x_out=None, mask=None, s_out=None, wt_out=None):
    """x = act([A . s_prev[..., :k] | s_prev[..., k:]] + bias_prev); s = x @ w.  s_prev [B, V, 192], w [192, n_out].
    Returns (x, s); `mask` (int16 [B*V*16]) receives the ReLU sign words, `wt_out` [n_out, 192] the transposed weight."""
    b, nv, c = s_prev.shape
    n_out = w.shape[1]
    x_out = torch.empty_like(s_prev) if x_out is None else x_out
    s_out = torch.empty(b, nv, n_out, dtype=torch.float32, device=s_prev.device) if s_out is None else s_out
    with torch.cuda.device(s_prev.device):
        _lib.call("geom_zn_layer_fwd_f32", b, nv, c, k, csr.ell_w, csr.ell_col.data_ptr(), csr.ell_val.data_ptr(),
                  s_prev.data_ptr(), _lib.ptr(bias_prev), act, w.data_ptr(), n_out, x_out.data_ptr(), _lib.ptr(mask),
                  s_out.data_ptr(), _lib.ptr(wt_out))
    return x_out, s_out


def layer_backward(grad_out, out, mask, csr, k, act, wt, g_out=None, grad_in=None, colsum_partial=None, grad_pos=None,
                   head_scale=0.0, shape=None):
    """g = [A^T . g'[..., :k] | g'[..., k:]], g' = grad_out * act'(out);  grad_in = g @ wt  (wt [192, n_in] = W^T).
    Head mode: grad_pos [B, V, 3] instead of grad_out (which is [head_scale * grad_pos | 0] by construction)."""
    b, nv, c = shape if shape is not None else grad_out.shape
    n_in = wt.shape[1]
    dev = wt.device
    g_out = torch.empty(b, nv, c, dtype=torch.float32, device=dev) if g_out is None else g_out
    grad_in = torch.empty(b, nv, n_in, dtype=torch.float32, device=dev) if grad_in is None else grad_in
    with torch.cuda.device(dev):
        _lib.call("geom_zn_layer_bwd_f32", b, nv, c, k, csr.ell_w, csr.ell_col_t.data_ptr(), csr.ell_val_t.data_ptr(),
                  _lib.ptr(grad_out), _lib.ptr(out), _lib.ptr(mask), act, _lib.ptr(grad_pos), float(head_scale), wt.data_ptr(),
                  n_in, g_out.data_ptr(), grad_in.data_ptr(), _lib.ptr(colsum_partial))
    return g_out, grad_in
