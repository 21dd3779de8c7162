"""Dev helper: the aggregation launches at the reference's training shape (16 x 482 vertices with two 33-entry pole rows,
C = 192, k = 64) beside a pole-free mesh of about the same number of rows (12 x 642-vertex icosphere)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import layers, meshgen, utils, _lib
dev = torch.device("cuda:0")


def t(fn, it=200):
    for _ in range(10): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for name, (V, F), B in (("uv_sphere 482 x16", meshgen.uv_sphere(), 16), ("icosphere 642 x12", meshgen.icosphere(3), 12)):
    adj = utils.adj_init(torch.from_numpy(F).to(dev))["adj"]
    csr = layers.adjacency_csr(adj)
    nv, C, k = V.shape[0], 192, 64
    sup = torch.randn(B, nv, C, device=dev); bias = torch.randn(C, device=dev); out = torch.empty_like(sup)
    gout = torch.randn_like(sup); gsup = torch.empty_like(sup); gb = torch.empty(C, device=dev)
    scr = torch.empty(_lib.lib().geom_zn_gcn_bwd_scratch_floats(B, nv, C), device=dev)
    over, over_t = csr.over or (None, None, None), csr.over_t or (None, None, None)
    fwd = lambda: _lib.call("geom_zn_gcn_aggregate_ell_fwd_f32", B, nv, C, k, csr.ell_w, csr.ell_col.data_ptr(), csr.ell_val.data_ptr(),
                            _lib.ptr(over[0]), _lib.ptr(over[1]), _lib.ptr(over[2]), sup.data_ptr(), bias.data_ptr(), 0, out.data_ptr(), None)
    bwd = lambda: _lib.call("geom_zn_gcn_aggregate_ell_bwd_f32", B, nv, C, k, csr.ell_w, csr.ell_col_t.data_ptr(), csr.ell_val_t.data_ptr(),
                            _lib.ptr(over_t[0]), _lib.ptr(over_t[1]), _lib.ptr(over_t[2]), gout.data_ptr(), None, None, 0,
                            gsup.data_ptr(), gb.data_ptr(), scr.data_ptr())
    print("%-20s ell_w %d tail %s   fwd %.1f us   bwd + colsum %.1f us" % (name, csr.ell_w, bool(csr.over), t(fwd), t(bwd)))
