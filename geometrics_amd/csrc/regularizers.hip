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

// ---- the regularisers of ONE deformation stage in one launch per direction ------------------------------------------------
// The reference's driver adds, per stage (GEOMetrics.py:147-161), the edge term of the new positions, the squared difference
// of the Laplacian coordinates of the previous and the new positions and (stages 2, 3) their squared displacement:
//     w_edge * mean_faces(|e1|^2 + |e2|^2 + |e3|^2) / 3 ... + w_lap * mean_v |lap(prev) - lap(cur)|^2 + w_move * mean_v |prev - cur|^2
// -- as torch expressions on top of the kernels above that is ~20 launches forward and ~25 backward per stage for 7 712 x 3
// floats.  lap is linear, so lap(prev) - lap(cur) = lap(prev - cur): the forward launch forms d = prev - cur, its Laplacian
// coordinates (saved for the backward) and all three sums' per-workgroup partials; the backward launch is a gather per vertex:
// the transposed Laplacian of the saved coordinates, the displacement, and the edge term's gradient over the vertex's incident
// (face, corner) list -- no float atomics, a fixed summation order.
struct StageRegArgs {
    const float *prev, *cur;   // prev [b,nv,3] or [nv,3] (prev_stride = 0); cur [b,nv,3]
    int64_t prev_stride;       // floats between two meshes of prev
    int b, nv, nf;
    const int64_t *faces;
    const int *rowptr, *col;
    const float *inv_deg;
    const int *vf_ptr, *vf_item; // vertex -> incident (face * 4 + corner), backward
    float c_lap, c_move, c_edge; // the weights divided by the means' counts
    float *lapd;                 // [b,nv,3]: lap(prev - cur)
    float *partial;              // forward: one partial sum per workgroup
    const float *gout;           // backward: the gradient of the scalar (device)
    float *grad_prev, *grad_cur; // backward ([b,nv,3]; grad_prev may be null)
};

__device__ __forceinline__ V3 rg_d(const StageRegArgs &a, int mesh, int j)
{
    return ld3(a.prev + (size_t)mesh * a.prev_stride + 3 * (size_t)j) - ld3(a.cur + ((size_t)mesh * a.nv + j) * 3);
}

// Eight lanes per vertex: a vertex's adjacency row (5-9 entries, 33 at the poles of 482.obj) and its incident-corner list are
// walked eight entries at a time and folded by a fixed xor tree -- one thread per vertex made both launches a chain of
// dependent round trips per entry (18.8 us backward, 9.7 forward for 7 712 vertices; the poles' threads set the pace).
constexpr int RG_SUB = 8;
__device__ __forceinline__ V3 rg_fold8(V3 s)
{
#pragma unroll
    for (int m = RG_SUB / 2; m > 0; m >>= 1) {
        s.x += __shfl_xor(s.x, m, GEOM_WAVE), s.y += __shfl_xor(s.y, m, GEOM_WAVE), s.z += __shfl_xor(s.z, m, GEOM_WAVE);
    }
    return s;
}

__global__ __launch_bounds__(RG_THREADS) void stage_reg_fwd_kernel(StageRegArgs a)
{
    __shared__ float red[RG_THREADS / GEOM_WAVE];
    const int64_t i = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x;
    const int64_t vi = i / RG_SUB;
    const int sub = (int)(i % RG_SUB);
    float t = 0.f;
    const bool vertex_on = vi < (int64_t)a.b * a.nv; // (uniform over the 8 lanes of a vertex)
    {
        const int mesh = vertex_on ? (int)(vi / a.nv) : 0, v = vertex_on ? (int)(vi - (int64_t)mesh * a.nv) : 0;
        V3 s = geom::mk(0.f, 0.f, 0.f);
        if (vertex_on)
            for (int e = a.rowptr[v] + sub; e < a.rowptr[v + 1]; e += RG_SUB) s = s + rg_d(a, mesh, a.col[e]);
        s = rg_fold8(s);
        if (vertex_on && sub == 0) {
            const V3 self = rg_d(a, mesh, v);
            const V3 lap = self - (s - self) * a.inv_deg[v];
            a.lapd[3 * vi + 0] = lap.x, a.lapd[3 * vi + 1] = lap.y, a.lapd[3 * vi + 2] = lap.z;
            t = a.c_lap * geom::dot3(lap, lap) + a.c_move * geom::dot3(self, self);
        }
    }
    if (a.c_edge != 0.f && i < (int64_t)a.b * a.nf) {
        const int mesh = (int)(i / a.nf), f = (int)(i - (int64_t)mesh * a.nf);
        const float *V = a.cur + (size_t)mesh * a.nv * 3;
        const V3 p1 = ld3(V + 3 * a.faces[3 * (size_t)f + 0]), p2 = ld3(V + 3 * a.faces[3 * (size_t)f + 1]), p3 = ld3(V + 3 * a.faces[3 * (size_t)f + 2]);
        const V3 e1 = p2 - p1, e2 = p3 - p1, e3 = p2 - p3;
        t += a.c_edge * ((geom::dot3(e1, e1) + geom::dot3(e2, e2)) + geom::dot3(e3, e3));
    }
    for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) t += __shfl_down(t, off, GEOM_WAVE);
    if ((threadIdx.x & (GEOM_WAVE - 1)) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) a.partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(RG_THREADS) void stage_reg_bwd_kernel(StageRegArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x;
    const int64_t vi = i / RG_SUB;
    const int sub = (int)(i % RG_SUB);
    const bool on = vi < (int64_t)a.b * a.nv;
    const int mesh = on ? (int)(vi / a.nv) : 0, v = on ? (int)(vi - (int64_t)mesh * a.nv) : 0;
    const float go = a.gout[0];
    const float *Y = a.lapd + (size_t)mesh * a.nv * 3;
    V3 s = geom::mk(0.f, 0.f, 0.f);
    if (on)
        for (int e = a.rowptr[v] + sub; e < a.rowptr[v + 1]; e += RG_SUB) {
            const int j = a.col[e];
            s = s + ld3(Y + 3 * j) * a.inv_deg[j];
        }
    s = rg_fold8(s);
    V3 ge = geom::mk(0.f, 0.f, 0.f);
    if (a.c_edge != 0.f) {
        const float *V = a.cur + (size_t)mesh * a.nv * 3;
        if (on)
            for (int e = a.vf_ptr[v] + sub; e < a.vf_ptr[v + 1]; e += RG_SUB) { // the vertex's (face, corner) list
                const int item = a.vf_item[e], f = item >> 2, corner = item & 3;
                const V3 p1 = ld3(V + 3 * a.faces[3 * (size_t)f + 0]), p2 = ld3(V + 3 * a.faces[3 * (size_t)f + 1]), p3 = ld3(V + 3 * a.faces[3 * (size_t)f + 2]);
                const V3 e1 = p2 - p1, e2 = p3 - p1, e3 = p2 - p3;
                ge = ge + (corner == 0 ? (e1 + e2) * -1.f : corner == 1 ? e1 + e3 : e2 - e3);
            }
        ge = rg_fold8(ge) * (2.f * a.c_edge * go);
    }
    if (!on || sub != 0) return;
    const V3 y = ld3(Y + 3 * v);
    const V3 lty = y - (s - y * a.inv_deg[v]);                                  // transposed Laplacian (laplacian_kernel<true>)
    const V3 g = (lty * (2.f * a.c_lap) + rg_d(a, mesh, v) * (2.f * a.c_move)) * go; // d total / d (prev - cur)[v]
    if (a.grad_prev) a.grad_prev[3 * vi + 0] = g.x, a.grad_prev[3 * vi + 1] = g.y, a.grad_prev[3 * vi + 2] = g.z;
    a.grad_cur[3 * vi + 0] = ge.x - g.x, a.grad_cur[3 * vi + 1] = ge.y - g.y, a.grad_cur[3 * vi + 2] = ge.z - g.z;
}

} // namespace

// partial [geom_stage_regularisers_blocks(b, nv, nf)] per-workgroup sums of
//   c_edge * sum_faces(|e1|^2+|e2|^2+|e3|^2)(cur) + c_lap * sum_v |lap(prev - cur)|^2 + c_move * sum_v |prev - cur|^2
// (the caller folds the means' 1 / counts into the c_*; finish with geom_sum_f32), lapd [b,nv,3] = lap(prev - cur) for the
// backward.  prev: [b,nv,3] (prev_batched != 0) or ONE [nv,3] mesh for the whole batch (the template, GEOMetrics.py:156).
extern "C" int64_t geom_stage_regularisers_blocks(int b, int nv, int nf)
{
    const int64_t nvt = (int64_t)nv * 8; // eight lanes per vertex (RG_SUB)
    const int64_t n = (int64_t)b * (nvt > nf ? nvt : nf);
    return (n + RG_THREADS - 1) / RG_THREADS;
}
extern "C" int geom_stage_regularisers_fwd_f32(int b, int nv, const float *prev, int prev_batched, const float *cur, int nf,
                                               const int64_t *faces, const int *rowptr, const int *col, const float *inv_deg,
                                               float c_lap, float c_move, float c_edge, float *lapd, float *partial, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0) return GEOM_EINVAL;
    if (b == 0 || nv == 0) return 0;
    if (!prev || !cur || !rowptr || !col || !inv_deg || !lapd || !partial || (nf > 0 && !faces)) return GEOM_EINVAL;
    StageRegArgs a{};
    a.prev = prev, a.cur = cur, a.prev_stride = prev_batched ? (int64_t)nv * 3 : 0, a.b = b, a.nv = nv, a.nf = nf, a.faces = faces;
    a.rowptr = rowptr, a.col = col, a.inv_deg = inv_deg, a.c_lap = c_lap, a.c_move = c_move, a.c_edge = nf > 0 ? c_edge : 0.f;
    a.lapd = lapd, a.partial = partial;
    hipLaunchKernelGGL(stage_reg_fwd_kernel, dim3((unsigned)geom_stage_regularisers_blocks(b, nv, nf)), dim3(RG_THREADS), 0,
                       static_cast<hipStream_t>(stream), a);
    return geom::launch_status();
}
// grad_cur [b,nv,3] (and grad_prev [b,nv,3] when given: batched prev only) of gout[0] * that sum; vf_ptr [nv+1] / vf_item: every
// vertex's incident corners as face * 4 + corner, ascending (the edge term's gradient as a gather: no atomics).
extern "C" int geom_stage_regularisers_bwd_f32(int b, int nv, const float *prev, int prev_batched, const float *cur, int nf,
                                               const int64_t *faces, const int *rowptr, const int *col, const float *inv_deg,
                                               const int *vf_ptr, const int *vf_item, float c_lap, float c_move, float c_edge,
                                               const float *lapd, const float *gout, float *grad_prev, float *grad_cur, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0) return GEOM_EINVAL;
    if (b == 0 || nv == 0) return 0;
    if (!prev || !cur || !rowptr || !col || !inv_deg || !lapd || !gout || !grad_cur) return GEOM_EINVAL;
    if (nf > 0 && c_edge != 0.f && (!faces || !vf_ptr || !vf_item)) return GEOM_EINVAL;
    if (grad_prev && !prev_batched) return GEOM_EUNSUPPORTED;
    StageRegArgs a{};
    a.prev = prev, a.cur = cur, a.prev_stride = prev_batched ? (int64_t)nv * 3 : 0, a.b = b, a.nv = nv, a.nf = nf, a.faces = faces;
    a.rowptr = rowptr, a.col = col, a.inv_deg = inv_deg, a.vf_ptr = vf_ptr, a.vf_item = vf_item;
    a.c_lap = c_lap, a.c_move = c_move, a.c_edge = nf > 0 ? c_edge : 0.f;
    a.lapd = const_cast<float *>(lapd), a.gout = gout, a.grad_prev = grad_prev, a.grad_cur = grad_cur;
    hipLaunchKernelGGL(stage_reg_bwd_kernel, rg_grid((int64_t)b * nv * RG_SUB), dim3(RG_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return geom::launch_status();
}

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

// out[i] = ((t0[i] + t1[i]) + t2[i]) + ...   for up to GEOM_SUM_MAX_TENSORS equally sized fp32 tensors: the gradients that reach one
// tensor through several consumers (a stage's positions feed the pooling, the next block, two regularisers, the surface loss
// and the next stage's sum: autograd adds them with one launch per consumer), in ONE launch and a fixed order.
namespace {
struct SumArgs {
    const float *t[GEOM_SUM_MAX_TENSORS];
    int count;
};
__global__ __launch_bounds__(RG_THREADS) void sum_tensors_kernel(SumArgs a, int64_t n, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x;
    if (i >= n) return;
    float s = a.t[0][i];
#pragma unroll
    for (int k = 1; k < GEOM_SUM_MAX_TENSORS; ++k)
        if (k < a.count) s += a.t[k][i];
    out[i] = s;
}
} // namespace

namespace {
struct SumRowsArgs {
    const float *t[GEOM_SUM_MAX_TENSORS];
    int64_t ld[GEOM_SUM_MAX_TENSORS];
    int count, width;
};
__global__ __launch_bounds__(RG_THREADS) void sum_rows_kernel(SumRowsArgs a, int64_t n, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / a.width;
    const int c = (int)(i - r * a.width);
    float s = a.t[0][r * a.ld[0] + c];
#pragma unroll
    for (int k = 1; k < GEOM_SUM_MAX_TENSORS; ++k)
        if (k < a.count) s += a.t[k][r * a.ld[k] + c];
    out[i] = s;
}
} // namespace

// the same sum for [rows, width] operands of which some are column slices of wider row-major buffers: lds[k] = floats between
// two rows of tensors[k] (>= width); out is contiguous
extern "C" int geom_sum_tensors_rows_f32(int count, const float *const *tensors, const int64_t *lds, int64_t rows, int width, float *out,
                                         void *stream)
{
    if (count <= 0 || count > GEOM_SUM_MAX_TENSORS || rows < 0 || width <= 0) return GEOM_EINVAL;
    if (rows == 0) return 0;
    if (!tensors || !lds || !out) return GEOM_EINVAL;
    SumRowsArgs a{};
    for (int k = 0; k < count; ++k) {
        if (!tensors[k] || lds[k] < width) return GEOM_EINVAL;
        a.t[k] = tensors[k], a.ld[k] = lds[k];
    }
    a.count = count, a.width = width;
    hipLaunchKernelGGL(sum_rows_kernel, rg_grid(rows * width), dim3(RG_THREADS), 0, static_cast<hipStream_t>(stream), a, rows * width, out);
    return geom::launch_status();
}

extern "C" int geom_sum_tensors_f32(int count, const float *const *tensors, int64_t n, float *out, void *stream)
{
    if (count <= 0 || count > GEOM_SUM_MAX_TENSORS || n < 0) return GEOM_EINVAL;
    if (n == 0) return 0;
    if (!tensors || !out) return GEOM_EINVAL;
    SumArgs a{};
    for (int k = 0; k < count; ++k) {
        if (!tensors[k]) return GEOM_EINVAL;
        a.t[k] = tensors[k];
    }
    a.count = count;
    hipLaunchKernelGGL(sum_tensors_kernel, rg_grid(n), dim3(RG_THREADS), 0, static_cast<hipStream_t>(stream), a, n, out);
    return geom::launch_status();
}
