// Mesh regularisers of the training loss as sparse kernels.
//
//   batch_get_lap_info (reference utils.py:654-662): lap = p - (A_orig p - p) / (deg), where the
//     reference multiplies by the DENSE binary adjacency (26 MB read per call at V=2562, six calls
//     per training step, GEOMetrics.py:156-161).  Here: one thread per vertex walks its CSR row.
//   batch_calc_edge (reference utils.py:636-651): mean squared length of the three edges of every
//     face: three [B,F,3] gathers + six elementwise passes there, one kernel here.
#include "geom_common.h"
#include "tri_math.h"

namespace {

using geom::V3;
constexpr int RG_THREADS = 256;

__device__ __forceinline__ V3 ld3(const float *p) { return geom::mk(p[0], p[1], p[2]); }

// forward: out[v] = x[v] - (sum_{j in row v} x[j] - x[v]) * inv_deg[v]          (row includes the self loop)
// backward (transpose; A_orig is symmetric):
//          out[v] = g[v] - (sum_{j in row v} g[j] * inv_deg[j] - g[v] * inv_deg[v])
template <bool BACKWARD>
__global__ __launch_bounds__(RG_THREADS) void laplacian_kernel(int b, int nv, const int *rowptr, const int *col,
                                                                const float *inv_deg, const float *x, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x;
    if (i >= (int64_t)b * nv) return;
    const int mesh = (int)(i / nv);
    const int v = (int)(i - (int64_t)mesh * nv);
    const float *X = x + (size_t)mesh * nv * 3;
    V3 s = geom::mk(0.f, 0.f, 0.f);
    for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
        const int j = col[e];
        const V3 xj = ld3(X + 3 * j);
        s = s + (BACKWARD ? xj * inv_deg[j] : xj);
    }
    const V3 self = ld3(X + 3 * v);
    const V3 r = BACKWARD ? self - (s - self * inv_deg[v]) : self - (s - self) * inv_deg[v];
    out[3 * i + 0] = r.x;
    out[3 * i + 1] = r.y;
    out[3 * i + 2] = r.z;
}

// per face: |p2-p1|^2 + |p3-p1|^2 + |p2-p3|^2
__global__ __launch_bounds__(RG_THREADS) void edge_sqlen_fwd_kernel(int b, int nv, const float *verts, int nf,
                                                                     const int64_t *faces, float *per_face)
{
    const int64_t i = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x;
    if (i >= (int64_t)b * nf) return;
    const int mesh = (int)(i / nf);
    const int f = (int)(i - (int64_t)mesh * nf);
    const float *V = verts + (size_t)mesh * nv * 3;
    const V3 p1 = ld3(V + 3 * faces[3 * (size_t)f + 0]);
    const V3 p2 = ld3(V + 3 * faces[3 * (size_t)f + 1]);
    const V3 p3 = ld3(V + 3 * faces[3 * (size_t)f + 2]);
    const V3 e1 = p2 - p1, e2 = p3 - p1, e3 = p2 - p3;
    per_face[i] = (geom::dot3(e1, e1) + geom::dot3(e2, e2)) + geom::dot3(e3, e3);
}

// d/dverts of coef * sum_faces (|e1|^2 + |e2|^2 + |e3|^2)
__global__ __launch_bounds__(RG_THREADS) void edge_sqlen_bwd_kernel(int b, int nv, const float *verts, int nf,
                                                                     const int64_t *faces, const float *coef_dev,
                                                                     float coef_host, float *grad_verts)
{
    const int64_t i = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x;
    if (i >= (int64_t)b * nf) return;
    const int mesh = (int)(i / nf);
    const int f = (int)(i - (int64_t)mesh * nf);
    const float coef = 2.f * coef_host * (coef_dev ? coef_dev[0] : 1.f);
    const float *V = verts + (size_t)mesh * nv * 3;
    float *G = grad_verts + (size_t)mesh * nv * 3;
    const int64_t i1 = faces[3 * (size_t)f + 0], i2 = faces[3 * (size_t)f + 1], i3 = faces[3 * (size_t)f + 2];
    const V3 p1 = ld3(V + 3 * i1), p2 = ld3(V + 3 * i2), p3 = ld3(V + 3 * i3);
    const V3 e1 = (p2 - p1) * coef, e2 = (p3 - p1) * coef, e3 = (p2 - p3) * coef;
    const V3 g1 = (e1 + e2) * -1.f, g2 = e1 + e3, g3 = e2 - e3;
    atomicAdd(G + 3 * i1 + 0, g1.x), atomicAdd(G + 3 * i1 + 1, g1.y), atomicAdd(G + 3 * i1 + 2, g1.z);
    atomicAdd(G + 3 * i2 + 0, g2.x), atomicAdd(G + 3 * i2 + 1, g2.y), atomicAdd(G + 3 * i2 + 2, g2.z);
    atomicAdd(G + 3 * i3 + 0, g3.x), atomicAdd(G + 3 * i3 + 1, g3.y), atomicAdd(G + 3 * i3 + 2, g3.z);
}

inline dim3 rg_grid(int64_t count) { return dim3((unsigned)((count + RG_THREADS - 1) / RG_THREADS)); }

} // namespace

extern "C" int geom_laplacian_f32(int b, int nv, const int *rowptr, const int *col, const float *inv_deg,
                                  const float *x, int transpose, float *out, void *stream)
{
    if (b < 0 || nv < 0) return GEOM_EINVAL;
    if (b == 0 || nv == 0) return 0;
    if (!rowptr || !col || !inv_deg || !x || !out) return GEOM_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (transpose)
        hipLaunchKernelGGL(laplacian_kernel<true>, rg_grid((int64_t)b * nv), dim3(RG_THREADS), 0, s, b, nv, rowptr, col, inv_deg, x, out);
    else
        hipLaunchKernelGGL(laplacian_kernel<false>, rg_grid((int64_t)b * nv), dim3(RG_THREADS), 0, s, b, nv, rowptr, col, inv_deg, x, out);
    return geom::launch_status();
}

extern "C" int geom_edge_sqlen_fwd_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                                       float *per_face, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0) return GEOM_EINVAL;
    if (b == 0 || nf == 0) return 0;
    if (!verts || !faces || !per_face) return GEOM_EINVAL;
    hipLaunchKernelGGL(edge_sqlen_fwd_kernel, rg_grid((int64_t)b * nf), dim3(RG_THREADS), 0,
                       static_cast<hipStream_t>(stream), b, nv, verts, nf, faces, per_face);
    return geom::launch_status();
}

extern "C" int geom_edge_sqlen_bwd_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                                       const float *coef_dev, float coef_host, float *grad_verts, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0) return GEOM_EINVAL;
    if (b == 0 || nf == 0) return 0;
    if (!verts || !faces || !grad_verts) return GEOM_EINVAL;
    hipLaunchKernelGGL(edge_sqlen_bwd_kernel, rg_grid((int64_t)b * nf), dim3(RG_THREADS), 0,
                       static_cast<hipStream_t>(stream), b, nv, verts, nf, faces, coef_dev, coef_host, grad_verts);
    return geom::launch_status();
}
