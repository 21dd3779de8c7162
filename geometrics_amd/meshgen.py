"""Synthetic inputs for the hot path: icospheres at the
BASELINE sizes, jittered per-mesh vertices, noisy-sphere GT clouds and the
pre-drawn sampling randoms.  numpy only; deterministic in the seeds."""
import numpy as np

RADIUS = 0.46  # mean radius of the reference's 482.obj template


def icosphere(level):
    """12-vertex icosahedron + `level` midpoint subdivisions, de-duplicated.
    level 2 -> 162 v / 320 f, level 4 -> 2562 v / 5120 f.  Outward winding.
    Returns (verts float32 [V,3] on the RADIUS sphere, faces int64 [F,3])."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache = {}

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                p = verts[a] + verts[b]
                verts.append(p / np.linalg.norm(p))
                cache[key] = len(verts) - 1
            return cache[key]

        nxt = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nxt += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nxt
    V = (np.stack(verts) * RADIUS).astype(np.float32)
    F = np.asarray(faces, np.int64)
    return V, F


def uv_sphere(rings=15, segments=32):
    """Latitude/longitude sphere with two poles: rings*segments + 2 vertices, 2*segments*rings faces, outward
    winding.  The default 15 x 32 has the SIZE and the degree extremes of the reference's training template
    `482.obj` (GEOMetrics.py:44: 482 vertices, 960 faces, two poles with 32 neighbours -> adjacency rows of 33
    entries beside rows of 5-9); its interior rows all have 7 entries where 482.obj mixes 6-10."""
    v = [(0.0, 1.0, 0.0)]
    for r in range(1, rings + 1):
        th = np.pi * r / (rings + 1)
        for s in range(segments):
            ph = 2.0 * np.pi * s / segments
            v.append((np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)))
    v.append((0.0, -1.0, 0.0))
    south = len(v) - 1
    ring = lambda r, s: 1 + r * segments + (s % segments)
    f = []
    for s in range(segments):
        f.append((0, ring(0, s + 1), ring(0, s)))
        f.append((south, ring(rings - 1, s), ring(rings - 1, s + 1)))
    for r in range(rings - 1):
        for s in range(segments):
            a, b, c, d = ring(r, s), ring(r, s + 1), ring(r + 1, s), ring(r + 1, s + 1)
            f += [(a, b, d), (a, d, c)]
    V = (np.asarray(v, np.float64) * RADIUS).astype(np.float32)
    F = np.asarray(f, np.int64)
    # outward winding check: normal . centroid > 0 for every face
    n = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
    flip = (n * V[F].mean(1)).sum(1) < 0
    F[flip] = F[flip][:, ::-1]
    return V, F


def jittered_batch(verts, batch, first=0, sigma=0.02, seed=41):
    """Per-mesh Gaussian vertex jitter, seed 41+b (41 = the reference's default --seed)."""
    out = np.empty((batch,) + verts.shape, np.float32)
    for i in range(batch):
        rng = np.random.default_rng(seed + first + i)
        out[i] = verts + (sigma * rng.standard_normal(verts.shape)).astype(np.float32)
    return out


def gt_cloud(batch, num, first=0, seed=1041, cube=False):
    """GT points: noisy sphere r = RADIUS*(1+0.05 N(0,1)) (or U(-.5,.5)^3 when cube)."""
    out = np.empty((batch, num, 3), np.float32)
    for i in range(batch):
        rng = np.random.default_rng(seed + first + i)
        if cube:
            out[i] = (rng.random((num, 3)) - 0.5).astype(np.float32)
        else:
            x = rng.standard_normal((num, 3))
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            r = RADIUS * (1.0 + 0.05 * rng.standard_normal((num, 1)))
            out[i] = (x * r).astype(np.float32)
    return out


def face_areas(verts, faces):
    a = verts[:, faces[:, 0]] - verts[:, faces[:, 1]]
    b = verts[:, faces[:, 1]] - verts[:, faces[:, 2]]
    return 0.5 * np.linalg.norm(np.cross(a, b), axis=-1)


def sampling_draws(verts, faces, num, first=0, seed=2041):
    """Pre-drawn (choices int64 [B,num], u f32 [B,num] (already sqrt'ed), v f32 [B,num])
    so the CPU oracle and the GPU path consume identical randoms."""
    batch = verts.shape[0]
    areas = face_areas(verts.astype(np.float64), faces)
    choices = np.empty((batch, num), np.int64)
    u = np.empty((batch, num), np.float32)
    v = np.empty((batch, num), np.float32)
    for i in range(batch):
        rng = np.random.default_rng(seed + first + i)
        p = areas[i] / areas[i].sum()
        choices[i] = rng.choice(faces.shape[0], size=num, replace=True, p=p)
        u[i] = np.sqrt(rng.random(num, dtype=np.float32))
        v[i] = rng.random(num, dtype=np.float32)
    return choices, u, v
