"""Generate the golden fixtures in tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the fixtures it writes are pure
data (inputs + expected outputs) and are what travels to the GPU box.

How the reference is executed here
  * python stages (utils.batch_sample / batch_point_to_point / batch_point_to_surface /
    calc_point_to_line / calc_adj / normalize_adj, every class of layers.py): the
    reference modules are IMPORTED from /root/reference and run on CPU.  Modules the image
    lacks or that need nvcc are replaced in sys.modules before the import:
      torchvision(.transforms/.models)  -> inert placeholders (never called by these functions)
      chamfer_distance                  -> the reference's own CPU nnsearch (oracle/_ref, built
                                           from old_GEOMetrics/chamfer_distance/src/my_lib.c:4-26)
      tri_distance                      -> the C restatement of tri_distance.cu (oracle/)
    and torch.Tensor.cuda is made the identity.
  * NN vectors: emitted by that reference nnsearch binary directly.
  * random draws are captured by wrapping torch.multinomial and Uniform.sample_n, so a
    fixture holds (inputs, draws) -> outputs.

    python tests/golden/make_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import oracle  # noqa: E402
from geometrics_amd import meshgen  # noqa: E402

oracle.build()
assert oracle.have_ref(), "reference nnsearch (oracle/_ref) must be built first"


# ----------------------------------------------------------- import the reference ----
def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


tv = _placeholder("torchvision")
tv.transforms = _placeholder("torchvision.transforms", Normalize=_Inert, Compose=_Inert, Resize=_Inert, ToTensor=_Inert)
tv.models = _placeholder("torchvision.models")


class _RefChamfer(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        a, b = xyz1.detach().numpy(), xyz2.detach().numpy()
        _, i1, _, i2 = oracle.chamfer_nn(a, b, use_ref=True)
        return torch.from_numpy(i1), torch.from_numpy(i2)


class _OracleTri(torch.nn.Module):
    def forward(self, xyz1, tri1, tri2, tri3):
        d, p, i = oracle.tri_scan(*(t.detach().numpy() for t in (xyz1, tri1, tri2, tri3)))
        return torch.from_numpy(d), torch.from_numpy(p), torch.from_numpy(i)


_placeholder("chamfer_distance", ChamferDistance=_RefChamfer)
_placeholder("tri_distance", TriDistance=_OracleTri)
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
import utils as ref_utils  # noqa: E402  (the reference's utils.py)
import layers as ref_layers  # noqa: E402  (the reference's layers.py)

assert ref_utils.__file__.startswith(REF) and ref_layers.__file__.startswith(REF)


class CaptureDraws:
    """Record (choices, U1, U2) consumed by one reference batch_sample call."""

    def __enter__(self):
        self.choices, self.uniform = [], []
        self._m = torch.multinomial
        self._s = torch.distributions.Uniform.sample_n

        def multinomial(*a, **k):
            r = self._m(*a, **k)
            self.choices.append(r.clone())
            return r

        def sample_n(dist, n):
            r = self._s(dist, n)
            self.uniform.append(r.clone())
            return r

        torch.multinomial = multinomial
        torch.distributions.Uniform.sample_n = sample_n
        return self

    def __exit__(self, *exc):
        torch.multinomial = self._m
        torch.distributions.Uniform.sample_n = self._s

    def draws(self, batch):
        choices = torch.stack(self.choices).numpy().astype(np.int64)          # [B,num]
        u = torch.sqrt(self.uniform[0]).view(batch, -1).numpy()               # post-sqrt, as the kernel takes it
        v = self.uniform[1].view(batch, -1).numpy()
        return choices, u, v


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %6.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.requires_grad_(True) if grad else x


# ------------------------------------------------------------------ NN vectors ----
def nn_case(a, b):
    _, i1, _, i2 = oracle.chamfer_nn(a, b, use_ref=True)
    d1, _, d2, _ = oracle.chamfer_nn(a, b, use_ref=True)
    return dict(xyz1=a, xyz2=b, idx1=i1, idx2=i2, dist1=d1, dist2=d2)


def make_nn():
    V2, F2 = meshgen.icosphere(2)
    verts = meshgen.jittered_batch(V2, 2)
    ch, u, v = meshgen.sampling_draws(verts, F2, 500)
    x, y, z = (np.take_along_axis(verts, F2[ch][..., k][..., None].repeat(3, -1), 1) for k in range(3))
    pred = ((1 - u)[..., None] * x + (u * (1 - v))[..., None] * y + (u * v)[..., None] * z).astype(np.float32)
    save("nn_config1", **nn_case(meshgen.gt_cloud(2, 500), pred))           # 162-icosphere samples vs 500 GT
    rng = np.random.default_rng(7)
    a = rng.integers(-3, 4, (2, 400, 3)).astype(np.float32)
    b = rng.integers(-3, 4, (2, 700, 3)).astype(np.float32)
    b[:, 300:500] = b[:, :200]
    save("nn_ties", **nn_case(a, b))                                         # integer grid + duplicates
    for m in (1, 3, 4, 5, 6, 7, 511, 512, 513, 516, 2466):
        rng = np.random.default_rng(100 + m)
        save("nn_ragged_m%d" % m, **nn_case(rng.standard_normal((1, 97, 3)).astype(np.float32),
                                             rng.standard_normal((1, m, 3)).astype(np.float32)))
    # config 2 (3000 vs 3000): inputs are regenerated from seeds, only the reference outputs are stored
    gt, pr = meshgen.gt_cloud(2, 3000), meshgen.gt_cloud(2, 3000, first=100)
    c = nn_case(gt, pr)
    save("nn_config2_outputs", idx1=c["idx1"], idx2=c["idx2"], dist1=c["dist1"], dist2=c["dist2"],
         gt_first=np.int64(0), pred_first=np.int64(100), checksum=np.float64(gt.sum() + pr.sum()))


def make_nn_fma():
    """The second pinned arithmetic (GEOM_FLAG_NN_FMA): outputs of the SAME reference nnsearch source built with
    -mfma -ffp-contract=fast (oracle/_ref/libref_nnsearch_fma.so) on the inputs of every nn_* fixture."""
    assert oracle.have_ref_fma(), "needs oracle/_ref/libref_nnsearch_fma.so and an FMA-capable host"
    out = {}
    for name in sorted(n[:-4] for n in os.listdir(HERE) if n.startswith("nn_") and n.endswith(".npz")):
        g = dict(np.load(os.path.join(HERE, name + ".npz")))
        if name == "nn_config2_outputs":
            a, b = meshgen.gt_cloud(2, 3000, first=int(g["gt_first"])), meshgen.gt_cloud(2, 3000, first=int(g["pred_first"]))
        else:
            a, b = g["xyz1"], g["xyz2"]
        d1, i1, d2, i2 = oracle.chamfer_nn(a, b, oracle.FLAG_NN_FMA, use_ref=True)
        for k, v in (("idx1", i1), ("idx2", i2), ("dist1", d1), ("dist2", d2)):
            out[name + "." + k] = v
    save("nnfma_outputs", **out)


# -------------------------------------------------------- sampling + losses ----
def make_sampling_and_losses():
    V2, F2 = meshgen.icosphere(2)
    B, num = 2, 500
    verts = meshgen.jittered_batch(V2, B)
    faces = t(F2)
    gt = meshgen.gt_cloud(B, num)
    torch.manual_seed(41)

    pv = t(verts, grad=True)
    with CaptureDraws() as cap:
        pts = ref_utils.batch_sample(pv, faces, num=num)
    gp = torch.from_numpy(np.random.default_rng(3).standard_normal((B, num, 3)).astype(np.float32))
    pts.backward(gp)
    choices, u, v = cap.draws(B)
    save("sample_v162", verts=verts, faces=F2, choices=choices, u=u, v=v, points=pts.detach().numpy(),
         grad_points=gp.numpy(), grad_verts=pv.grad.numpy())

    info = {"faces": faces}
    for name, fn in (("p2p_v162", ref_utils.batch_point_to_point), ("p2s_v162", ref_utils.batch_point_to_surface)):
        pv = t(verts, grad=True)
        with CaptureDraws() as cap:
            loss, f1 = fn(pv, info, t(gt), num=num, f1=True)
        loss.backward()
        choices, u, v = cap.draws(B)
        save(name, verts=verts, faces=F2, gt=gt, choices=choices, u=u, v=v, loss=np.float32(loss.item()),
             f1=np.float64(f1), grad_verts=pv.grad.numpy())

    # calc_point_to_line on every option code, incl. ones the scan would not pick
    rng = np.random.default_rng(11)
    n = 7 * 40
    a, b, c = (rng.standard_normal((n, 3)).astype(np.float32) * 0.3 for _ in range(3))
    p = rng.standard_normal((n, 3)).astype(np.float32) * 0.5
    opt = np.repeat(np.arange(7, dtype=np.int32), 40)
    ta, tb, tc = t(a, True), t(b, True), t(c, True)
    loss = ref_utils.calc_point_to_line(t(p), [ta, tb, tc], t(opt))
    loss.backward()
    save("p2line_options", p=p, a=a, b=b, c=c, option=opt, loss=np.float32(loss.item()),
         grad_a=ta.grad.numpy(), grad_b=tb.grad.numpy(), grad_c=tc.grad.numpy())

    # the restated tri scan vs the reference's own closest-point formulas (pins oracle_tri_scan)
    V3_, F3 = meshgen.icosphere(3)
    verts3 = meshgen.jittered_batch(V3_, 1, first=5)
    pts3 = meshgen.gt_cloud(1, 600, first=5)
    tri = [np.ascontiguousarray(verts3[:, F3[:, k]]) for k in range(3)]
    d, code, idx = oracle.tri_scan(pts3, *tri)
    per_point = []
    for j in range(pts3.shape[1]):
        corners = [t(tri[k][0, idx[0, j]][None]) for k in range(3)]
        per_point.append(ref_utils.calc_point_to_line(t(pts3[0, j][None]), corners, t(code[0, j:j + 1])).item())
    save("tri_vs_ref_formulas", verts=verts3, faces=F3, points=pts3, dist=d, option=code, index=idx,
         ref_sqdist=np.asarray(per_point, np.float32))


# --------------------------------- independent pin of the tri scan (legacy Eberly regions) ----
def legacy_point_to_line():
    """The reference's OTHER point-to-triangle implementation: old_GEOMetrics/utils.py:734-1026 `point_to_line`
    (Eberly's region formulation, vectorised over all (point, face) pairs; returns the MEAN over points of the
    per-point minimum squared distance).  The legacy module as a whole does not import under python 3, so the
    function is compiled from the reference file where it lies, by line range (same recipe as
    oracle/build_ref.sh); nothing of it is written to the repo."""
    path = os.path.join(REF, "old_GEOMetrics", "utils.py")
    with open(path) as f:
        lines = f.readlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("def point_to_line("))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("def "))
    assert (start + 1, end) == (734, 1027), (start + 1, end)
    scope = {"torch": torch, "np": np}
    exec(compile("".join(lines[start:end]), path, "exec"), scope)
    return scope["point_to_line"]


def true_min_sqdist(fn, verts, faces, points):
    """Per-point minimum squared distance to the mesh from the legacy function, in float64, one point per call (a
    call returns the mean over its points, so one point per call isolates that point's minimum)."""
    v, f = torch.from_numpy(verts.astype(np.float64)), torch.from_numpy(faces.astype(np.int64))
    pts = torch.from_numpy(points.astype(np.float64))
    return np.array([float(fn(v, f, pts[j:j + 1])) for j in range(pts.shape[0])])


def make_tri_true():
    """tri_true_*.npz: what the arg-min of tri_distance.cu has to agree with -- the exact point-to-mesh squared
    distance of every query point, from an implementation that shares no code or structure with the kernel."""
    fn = legacy_point_to_line()
    V2, F2 = meshgen.icosphere(2)
    V4, F4 = meshgen.icosphere(4)
    cases = {
        # BASELINE config 1 (2 meshes), config 3 (mesh 0 of the bench workload), uniform-cube stress points
        "tri_true_config1": (meshgen.jittered_batch(V2, 2), F2, meshgen.gt_cloud(2, 500), dict(level=2, batch=2, num=500, cube=0)),
        "tri_true_config3": (meshgen.jittered_batch(V4, 1), F4, meshgen.gt_cloud(1, 3000), dict(level=4, batch=1, num=3000, cube=0)),
        "tri_true_cube": (meshgen.jittered_batch(V2, 2, first=3), F2, meshgen.gt_cloud(2, 400, first=3, cube=True),
                          dict(level=2, batch=2, num=400, cube=1, first=3)),
    }
    for name, (verts, faces, pts, meta) in cases.items():
        true = np.stack([true_min_sqdist(fn, verts[b], faces, pts[b]) for b in range(verts.shape[0])])
        save(name, true_sqdist=true, checksum=np.float64(verts.astype(np.float64).sum() + pts.astype(np.float64).sum()),
             **{k: np.int64(v) for k, v in meta.items()})


# ---------------------------------------------------------------- adjacency ----
def make_adjacency():
    info, feats = ref_utils.load_initial(os.path.join(REF, "482.obj"))
    faces = info["faces"].numpy()
    adj = info["adj"].numpy()
    r, c = np.nonzero(adj)
    save("adj_482", faces=faces, nnz_rows=r.astype(np.int32), nnz_cols=c.astype(np.int32), nnz_vals=adj[r, c],
         orig_rowsum=info["adj_orig"].numpy().sum(1).astype(np.float32), verts=feats.numpy())
    V2, F2 = meshgen.icosphere(2)
    info = ref_utils.adj_init(t(F2))
    save("adj_ico162", faces=F2, adj=info["adj"].numpy(), adj_orig=info["adj_orig"].numpy())


# ------------------------------------------------------------------- layers ----
def make_layers():
    import torch.nn.functional as F
    V2, F2 = meshgen.icosphere(2)
    adj = ref_utils.adj_init(t(F2))["adj"]
    V = V2.shape[0]
    torch.manual_seed(40)
    cases = {
        "ZERON_GCN": (ref_layers.ZERON_GCN(24, 60), (V, 24), F.elu),
        "BatchZERON_GCN": (ref_layers.BatchZERON_GCN(24, 60), (3, V, 24), F.elu),
        "Batch_Image_ZERON_GCNGCN": (ref_layers.Batch_Image_ZERON_GCNGCN(33, 48), (2, V, 33), F.relu),
        "Batch_Image_ZERON_GCNGCN_out3": (ref_layers.Batch_Image_ZERON_GCNGCN(48, 3), (2, V, 48), lambda x: x),
        "GCNMax": (ref_layers.GCNMax(30, 50), (V, 30), F.elu),
        "BatchGCNMax": (ref_layers.BatchGCNMax(30, 50), (2, V, 30), F.elu),
    }
    for name, (layer, shape, act) in cases.items():
        x = torch.randn(*shape, requires_grad=True)
        out = layer(x, adj, act)
        gout = torch.randn_like(out)
        out.backward(gout)
        arrays = dict(x=x.detach().numpy(), out=out.detach().numpy(), grad_out=gout.numpy(), grad_x=x.grad.numpy())
        for pn, p in layer.named_parameters():
            arrays["param." + pn] = p.detach().numpy()
            arrays["grad." + pn] = p.grad.numpy()
        save("layer_" + name, **arrays)
    save("layer_adj", adj=adj.numpy())


# ------------------------------------------------------------- regularisers ----
def make_regularisers():
    V2, F2 = meshgen.icosphere(2)
    B = 3
    verts = meshgen.jittered_batch(V2, B, first=9)
    info = ref_utils.adj_init(t(F2))
    pv = t(verts, grad=True)
    lap = ref_utils.batch_get_lap_info(pv, info)
    g = torch.from_numpy(np.random.default_rng(4).standard_normal(lap.shape).astype(np.float32))
    lap.backward(g)
    pe = t(verts, grad=True)
    edge = ref_utils.batch_calc_edge(pe, info)
    edge.backward()
    save("regularisers_v162", verts=verts, faces=F2, lap=lap.detach().numpy(), grad_lap=g.numpy(),
         grad_verts_lap=pv.grad.numpy(), edge=np.float32(edge.item()), grad_verts_edge=pe.grad.numpy())


# ------------------------------------------------------------ deformation block ----
def make_block():
    import models as ref_models                      # the reference's models.py
    assert ref_models.__file__.startswith(REF)
    V2, F2 = meshgen.icosphere(2)
    adj = ref_utils.adj_init(t(F2))["adj"]
    torch.manual_seed(44)
    block = ref_models.BatchMeshDeformationBlock(32, V2.shape[0], hidden=24, output_features=3)
    block.train()
    with torch.no_grad():                             # non-trivial BN affine parameters
        for i in range(1, 14):
            getattr(block, "bn%d" % i).weight.uniform_(0.5, 1.5)
            getattr(block, "bn%d" % i).bias.uniform_(-0.2, 0.2)
    state0 = {k: v.clone() for k, v in block.state_dict().items()}
    feats = torch.randn(3, V2.shape[0], 3, requires_grad=True)
    pooled = torch.randn(3, V2.shape[0], 29, requires_grad=True)
    out_f, coords = block(feats, pooled, adj)
    gf, gc = torch.randn_like(out_f), torch.randn_like(coords)
    (out_f * gf).sum().add((coords * gc).sum()).backward()
    arrays = dict(adj=adj.numpy(), features=feats.detach().numpy(), pooled=pooled.detach().numpy(),
                  out_features=out_f.detach().numpy(), coords=coords.detach().numpy(), g_features=gf.numpy(),
                  g_coords=gc.numpy(), grad_features=feats.grad.numpy(), grad_pooled=pooled.grad.numpy())
    for k, v in state0.items():
        arrays["state." + k] = v.numpy()
    for k in ("gc1.weight1", "gc1.bias", "gc7.weight1", "gc15.weight1", "gc15.bias", "bn1.weight", "bn1.bias",
              "bn13.weight", "bn6.bias"):
        arrays["grad." + k] = dict(block.named_parameters())[k].grad.numpy()
    for k in ("bn1.running_mean", "bn1.running_var", "bn13.running_mean", "bn13.running_var"):
        arrays["after." + k] = block.state_dict()[k].numpy()
    save("deformation_block_v162", **arrays)


# --------------------------------------------------------------- image pooling ----
def make_pooling():
    torch.cuda.LongTensor = torch.LongTensor          # the reference casts indices with .type(torch.cuda.LongTensor)
    V2, _ = meshgen.icosphere(2)
    B = 2
    rng = np.random.default_rng(17)
    verts = meshgen.jittered_batch(V2, B, first=3)
    verts[:, :12] *= 3.0                               # some vertices project outside the image -> clamped
    info = np.stack([rng.uniform(0, 360, B), rng.uniform(10, 40, B), rng.uniform(0.9, 1.3, B)], 1).astype(np.float32)
    chans, dims = (4, 8, 8, 16), (56, 28, 14, 7)
    blocks = [torch.from_numpy(rng.standard_normal((B, c, d, d)).astype(np.float32)).requires_grad_(True)
              for c, d in zip(chans, dims)]
    pv = t(verts, grad=True)
    feats = ref_utils.batched_pooling(blocks, pv, t(info))
    g = torch.from_numpy(rng.standard_normal(tuple(feats.shape)).astype(np.float32))
    feats.backward(g)
    cam_mat, cam_pos = ref_utils.batch_camera_info(t(info))
    arrays = dict(verts=verts, img_info=info, features=feats.detach().numpy(), grad_out=g.numpy(),
                  grad_verts=pv.grad.numpy(), cam_mat=cam_mat.numpy(), cam_pos=cam_pos.numpy())
    for i, blk in enumerate(blocks):
        arrays["block%d" % i] = blk.detach().numpy()
        arrays["grad_block%d" % i] = blk.grad.numpy()
    save("pooling_v162", **arrays)


# ---------------------------------------------------- ragged mesh encoder (8f row 4) ----
def ragged_meshes():
    """Three meshes of different sizes and degree patterns: icosphere(1), icosphere(2), and icosphere(1) with one
    face split at its centroid (a degree-3 vertex next to degree-6/7 ones)."""
    V1, F1 = meshgen.icosphere(1)
    V2, F2 = meshgen.icosphere(2)
    c = V1[F1[0]].mean(0, keepdims=True)
    V3 = np.concatenate([V1, c]).astype(np.float32)
    n = V1.shape[0]
    a, b, d = F1[0]
    F3 = np.concatenate([F1[1:], np.array([[a, b, n], [b, d, n], [d, a, n]], dtype=F1.dtype)])
    rng = np.random.default_rng(5)
    return [(V + 0.02 * rng.standard_normal(V.shape)).astype(np.float32) for V in (V1, V2, V3)], [F1, F2, F3]


def make_encoder():
    """auto_encoder.py:71-76 run literally: the reference MeshEncoder on one mesh at a time with its dense
    normalised adjacency (utils.py:256-257), latents stacked; gradients of sum(latent * g)."""
    import models as ref_models
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import fill_parameters
    assert ref_models.__file__.startswith(REF)
    verts, faces = ragged_meshes()
    enc = fill_parameters(ref_models.MeshEncoder(50), 77)
    pos = [t(v, grad=True) for v in verts]
    latents = torch.stack([enc(p, ref_utils.normalize_adj(ref_utils.calc_adj(t(f)))) for p, f in zip(pos, faces)])
    g = torch.from_numpy(np.random.default_rng(6).standard_normal(tuple(latents.shape)).astype(np.float32))
    (latents * g).sum().backward()
    arrays = dict(verts=np.concatenate(verts), faces=np.concatenate(faces), sizes=np.array([v.shape[0] for v in verts]),
                  face_counts=np.array([f.shape[0] for f in faces]), seed=np.array(77), latents=latents.detach().numpy(),
                  g=g.numpy(), grad_verts=np.concatenate([p.grad.numpy() for p in pos]))
    params = dict(enc.named_parameters())
    for k in ("h1.weight", "h1.bias", "h24.weight", "h11.bias", "reduce.weight_Ws.0", "reduce.weight_Bs.0"):
        arrays["grad." + k] = params[k].grad.numpy()
    # the same run in float64: how far fp32 summation order alone moves the result
    enc64 = fill_parameters(ref_models.MeshEncoder(50), 77).double()
    lat64 = torch.stack([enc64(t(v).double(), ref_utils.normalize_adj(ref_utils.calc_adj(t(f))).double())
                         for v, f in zip(verts, faces)])
    arrays["latents_f64"] = lat64.detach().numpy()
    save("mesh_encoder_ragged", **arrays)


if __name__ == "__main__":
    if "--encoder" in sys.argv:
        make_encoder()
        sys.exit(0)
    if "--pooling" in sys.argv:
        make_pooling()
        sys.exit(0)
    if "--block" in sys.argv:
        make_block()
        sys.exit(0)
    if "--nn-fma" in sys.argv:
        make_nn_fma()
        sys.exit(0)
    if "--tri-true" in sys.argv:
        make_tri_true()
        sys.exit(0)
    make_regularisers()
    if "--only-new" in sys.argv:
        sys.exit(0)
    make_nn()
    make_nn_fma()
    make_sampling_and_losses()
    make_adjacency()
    make_layers()
    make_block()
    make_pooling()
    make_tri_true()
