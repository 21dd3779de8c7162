"""Dense products of the 0N-GCN layers on the fp32 matrix cores (csrc/dense_gemm.hip) -- thin host wrappers over the
C ABI (`geom_dense_*`, include/geom_hip.h).  The reference computes `support = torch.matmul(input, weight)`
(layers.py:30, 107, 140) and leaves its two gradients to autograd's library calls; these entry points are the same three
products, exact fp32, with the layer's own epilogue / prologue folded in (see the kernel file for why).

Everything here takes 2-D row-major views: x [rows, cin] (rows = b*V), w [cin, c].
"""
import ctypes

import torch

from . import _lib


def supported(cin, c, rows=1):
    """Shapes the matrix-core kernels take (anything else: the library product)."""
    return 0 < c <= 192 and c % 16 == 0 and cin > 0 and rows * max(cin, c) < 2 ** 30


def plan(rows, cin, c):
    """Which implementation runs each of a layer's three products, from the timings on MI355X (tools/time_dense.py,
    profiles/): {"fwd", "dx", "dw"} -> "mfma" | "lib", plus "pair" = input and weight gradient share one launch.
    * weight gradient: the split kernel + the shared end-of-pass reduction beat the library's split-K product + its own
      reduction launch at every shape of the path;
    * input gradient of a layer with cin <= 192: free inside the pair launch (two workgroups per CU); for the 963-wide
      first layer the library's product is ahead;
    * forward: the library for plain products (its many small tiles overlap their memory phases; one 80-row tile per
      CU pays ~8 us of prologue + epilogue per launch)."""
    ok = supported(cin, c, rows)
    pair = ok and cin <= 192 and cin % 4 == 0 and rows >= 512
    ok = ok and c % 48 == 0        # the weight-gradient kernel: every wave owns whole 12-column groups of the output
    pair = pair and ok
    return {"fwd": "lib", "dx": "mfma" if pair else "lib", "dw": "mfma" if ok and rows >= 512 else "lib", "pair": pair}


def forward(x, w, out=None):
    """support = x @ w."""
    rows, cin = x.shape
    c = w.shape[1]
    out = torch.empty(rows, c, dtype=torch.float32, device=x.device) if out is None else out
    with torch.cuda.device(x.device):
        _lib.call("geom_dense_fwd_f32", rows, cin, c, x.data_ptr(), w.data_ptr(), 0, None, out.data_ptr(), None, None)
    return out


def forward_split(x, w, bias, ksplit, out, sup, mask):
    """out[:, ksplit:] = relu(x @ w[:, ksplit:] + bias[ksplit:]) (finished), sup = x @ w[:, :ksplit] (raw, compact), the
    sign bits of the finished columns into mask [rows, c/16] (uint16 as int16)."""
    rows, cin = x.shape
    c = w.shape[1]
    with torch.cuda.device(x.device):
        _lib.call("geom_dense_fwd_f32", rows, cin, c, x.data_ptr(), w.data_ptr(), ksplit, _lib.ptr(bias), out.data_ptr(),
                  sup.data_ptr(), _lib.ptr(mask))


def backward_input(g, w, out=None):
    """grad_x = g @ w.T   (g [rows, c], w [cin, c])."""
    rows, c = g.shape
    cin = w.shape[0]
    out = torch.empty(rows, cin, dtype=torch.float32, device=g.device) if out is None else out
    with torch.cuda.device(g.device):
        _lib.call("geom_dense_bwd_input_f32", rows, cin, c, g.data_ptr(), w.data_ptr(), out.data_ptr())
    return out


def weight_workspace(rows, cin, c, device):
    n = int(_lib.lib().geom_dense_bwd_weight_workspace_floats(rows, cin, c))
    return torch.empty(n, dtype=torch.float32, device=device)


# bench.py: a list; while it is set every weight-partial launch is bracketed by HIP events on its stream and appends
# (rows, cin, c, start, end) -- the launch timed where it runs, inside a step
launch_probe = None


def backward_weight_partials(x, g, workspace, want_colsum=False):
    """Per-split partial tiles of x.T @ g (and of g's column sums) into `workspace`; `reduce` finishes them."""
    rows, cin = x.shape
    c = g.shape[1]
    probe = launch_probe
    with torch.cuda.device(x.device):
        if probe is not None:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        _lib.call("geom_dense_bwd_weight_f32", rows, cin, c, x.data_ptr(), g.data_ptr(), workspace.data_ptr(),
                  1 if want_colsum else 0)
        if probe is not None:
            end.record()
            probe.append((rows, cin, c, start, end))


def backward_pair(x, g, w, grad_x, workspace, want_colsum=False):
    """grad_x = g @ w.T and the partial tiles of x.T @ g (+ column sums of g) -- ONE launch where the shape allows."""
    rows, cin = x.shape
    c = g.shape[1]
    with torch.cuda.device(x.device):
        _lib.call("geom_dense_bwd_f32", rows, cin, c, x.data_ptr(), g.data_ptr(), w.data_ptr(), grad_x.data_ptr(),
                  workspace.data_ptr(), 1 if want_colsum else 0)


def reduce(jobs, stream=None):
    """jobs = [(rows, cin, c, workspace, grad_w, grad_bias or None)]: all pending weight / bias gradients in ONE launch."""
    n = len(jobs)
    if n == 0:
        return
    if n > _lib.DENSE_MAX_LAYERS:
        for i in range(0, n, _lib.DENSE_MAX_LAYERS):
            reduce(jobs[i:i + _lib.DENSE_MAX_LAYERS], stream)
        return
    ints = lambda k: (ctypes.c_int * n)(*[j[k] for j in jobs])
    ptrs = lambda k: (ctypes.c_void_p * n)(*[None if j[k] is None else j[k].data_ptr() for j in jobs])
    dev = jobs[0][3].device
    with torch.cuda.device(dev):
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().geom_dense_reduce_f32(n, ints(0), ints(1), ints(2), ptrs(3), ptrs(4), ptrs(5), s),
                   "geom_dense_reduce_f32")


def backward_weight(x, g, want_bias=False):
    """grad_w = x.T @ g (and grad_bias = column sums of g): partial launch + reduction launch."""
    rows, cin = x.shape
    c = g.shape[1]
    ws = weight_workspace(rows, cin, c, x.device)
    backward_weight_partials(x, g, ws, want_bias)
    gw = torch.empty(cin, c, dtype=torch.float32, device=x.device)
    gb = torch.empty(c, dtype=torch.float32, device=x.device) if want_bias else None
    reduce([(rows, cin, c, ws, gw, gb)])
    return gw, gb


# ---- any-shape products (csrc/dense_any.hip): the layers whose widths the 192-column kernels do not take -------------------
def any_supported(rows, cin, c):
    """Every 2-D fp32 product below 2 GiB per operand; worth it from a few hundred rows (launch-bound below)."""
    return rows >= 256 and cin > 0 and c > 0 and rows * max(cin, c) * 4 < 2 ** 31 - 1


def _row_major(t):
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]


def gemm(a, b, trans_a=False, trans_b=False, out=None, workspace=None):
    """out = op(a) @ op(b), exact fp32 on the matrix cores; a, b 2-D with unit stride along their rows (any row pitch).
    trans_a: a is [k, m] (x^T . g);  trans_b: b is [n, k] (g . w^T).  `workspace`: geom_gemm_workspace_floats floats for the
    split over the summed index (allocated here when the plan wants one and none is given)."""
    if not (_row_major(a) and _row_major(b)):
        raise ValueError("gemm operands must be 2-D with contiguous rows")
    for name, t in (("a", a), ("b", b), ("out", out), ("workspace", workspace)):     # raw pointers go to the kernel: no silent casts
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32:
            raise ValueError("gemm: %s must be a float32 tensor on a HIP device (got %s on %s)" % (name, t.dtype, t.device))
        if t.device != a.device:
            raise ValueError("gemm: %s is on %s, a on %s" % (name, t.device, a.device))
    k, m = (a.shape if trans_a else a.shape[::-1])
    n, kb = (b.shape if trans_b else b.shape[::-1])
    if k != kb:
        raise ValueError("gemm: summed extents differ (%d vs %d)" % (k, kb))
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    elif not _row_major(out) or tuple(out.shape) != (m, n):
        raise ValueError("gemm: out must be [%d, %d] with contiguous rows" % (m, n))
    need = int(_lib.lib().geom_gemm_workspace_floats(m, n, k))
    if need and (workspace is None or workspace.numel() < need):
        workspace = torch.empty(need, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.call("geom_gemm_f32", m, n, k, a.data_ptr(), a.stride(0), 1 if trans_a else 0, b.data_ptr(), b.stride(0),
                  0 if trans_b else 1, out.data_ptr(), out.stride(0), _lib.ptr(workspace) if need else None, need)
    return out


# ---- EXPERIMENT: exact fp32 products on the bf16 matrix cores (csrc/dense_split_bf16.hip); on no default route -----------------
def split_bf16_planes(w):
    """w [k, 192] fp32 -> planes [3, 192, kpad] (bf16 bit patterns as int16): w == plane 0 + plane 1 + plane 2 exactly."""
    if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()):
        raise ValueError("split_bf16_planes: a contiguous 2-D float32 tensor on a HIP device")
    k, n = w.shape
    kpad = int(_lib.lib().geom_split_bf16_kpad(k))
    planes = torch.empty(3, n, kpad, dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.call("geom_split_bf16_planes_f32", k, n, w.data_ptr(), planes.data_ptr())
    return planes


def gemm_split_bf16(a, planes, terms=6, out=None):
    """a [m, k] @ w [k, 192] with w given as split_bf16_planes(w): exact products, fp32 accumulation, on the bf16 matrix cores."""
    if not (a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.is_contiguous()):
        raise ValueError("gemm_split_bf16: a contiguous 2-D float32 tensor on a HIP device")
    m, k = a.shape
    n = planes.shape[1]
    if planes.shape[2] != int(_lib.lib().geom_split_bf16_kpad(k)):
        raise ValueError("gemm_split_bf16: the planes belong to another k")
    out = torch.empty(m, n, dtype=torch.float32, device=a.device) if out is None else out
    with torch.cuda.device(a.device):
        _lib.call("geom_gemm_split_bf16_f32", m, k, n, a.data_ptr(), planes.data_ptr(), out.data_ptr(), int(terms))
    return out
