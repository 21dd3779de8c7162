import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import layers, meshgen, utils, _lib
dev = torch.device("cuda:0")
V, F = meshgen.icosphere(4)
adj = utils.adj_init(torch.from_numpy(F).to(dev))["adj"]
csr = layers.adjacency_csr(adj)
B, C = 8, 192
sup = torch.randn(B, V.shape[0], C, device=dev); bias = torch.randn(C, device=dev); out = torch.empty_like(sup)
def run(k, act=0):
    _lib.call("geom_zn_gcn_aggregate_fwd_f32", B, V.shape[0], C, k, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.val.data_ptr(), sup.data_ptr(), bias.data_ptr(), act, out.data_ptr())
def run_ell(k, act=0):
    _lib.call("geom_zn_gcn_aggregate_ell_fwd_f32", B, V.shape[0], C, k, csr.ell_w, csr.ell_col.data_ptr(), csr.ell_val.data_ptr(), None, None, None, sup.data_ptr(), bias.data_ptr(), act, out.data_ptr(), mask.data_ptr() if (act == 1 and use_mask[0]) else None)
mask = torch.empty(_lib.lib().geom_zn_gcn_relu_mask_words(B, V.shape[0], C, 64), dtype=torch.int16, device=dev)
use_mask = [False]
gout = torch.randn_like(sup); gsup = torch.empty_like(sup); gb = torch.empty(C, device=dev)
scr = torch.empty(_lib.lib().geom_zn_gcn_bwd_scratch_floats(B, V.shape[0], C), device=dev)
def bwd(ell, act=1):
    if ell:
        _lib.call("geom_zn_gcn_aggregate_ell_bwd_f32", B, V.shape[0], C, 64, csr.ell_w, csr.ell_col_t.data_ptr(), csr.ell_val_t.data_ptr(), None, None, None, gout.data_ptr(), out.data_ptr(), mask.data_ptr() if use_mask[0] else None, act, gsup.data_ptr(), gb.data_ptr(), scr.data_ptr())
    else:
        _lib.call("geom_zn_gcn_aggregate_bwd_f32", B, V.shape[0], C, 64, csr.rowptr_t.data_ptr(), csr.col_t.data_ptr(), csr.val_t.data_ptr(), gout.data_ptr(), out.data_ptr(), act, gsup.data_ptr(), gb.data_ptr(), scr.data_ptr())
def t(fn, it=200):
    for _ in range(10): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
print("clone        ", t(lambda: out.copy_(sup)))
print("add bias     ", t(lambda: torch.add(sup, bias, out=out)))
for k in (0, 16, 64, 128, 192):
    print("agg k=%3d    " % k, t(lambda: run(k)), " relu", t(lambda: run(k, 1)))
print("ELL fwd k=64 ", t(lambda: run_ell(64)), " relu", t(lambda: run_ell(64, 1)))
print("bwd+colsum CSR", t(lambda: bwd(False)), "  ELL", t(lambda: bwd(True)))
use_mask[0] = True
run_ell(64, 1)
print("ELL relu fwd writing the sign mask", t(lambda: run_ell(64, 1)), "  ELL bwd from the mask", t(lambda: bwd(True)))
