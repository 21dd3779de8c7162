// HBM-bound stages around the two scans: face areas, fused gather + barycentric sampling
// (forward / backward), Chamfer gather-loss gradient, point-to-surface loss (forward /
// backward) and a deterministic sum.  One thread per point; every operand is read once
// and every result written once (algorithmic bytes in DESIGN.md section 4).
//
// Reference stages replaced (all eager PyTorch op chains there):
//   utils.py:596-602   face areas                     -> face_areas_kernel
//   utils.py:615-631   gather + barycentric point     -> sample_fwd_kernel / sample_bwd_kernel
//   utils.py:454-462   NN-pair gather + squared diff  -> chamfer_grad_kernel (forward = the
//                                                        distances the NN scan already wrote)
//   utils.py:472-481, 506-587  calc_point_to_line     -> p2tri_fwd_kernel / p2tri_bwd_kernel
#include "geom_common.h"
#include "tri_math.h"
#include "draw_body.h"

namespace {

using geom::V3;

constexpr int PT_THREADS = 256;

__device__ __forceinline__ V3 ld3(const float *p) { return geom::mk(p[0], p[1], p[2]); }

__device__ __forceinline__ void atomic_add3(float *dst, V3 v)
{
    atomicAdd(dst + 0, v.x);
    atomicAdd(dst + 1, v.y);
    atomicAdd(dst + 2, v.z);
}

// ---------------------------------------------------------------- face areas ----
// utils.py:596-602: x = v0 - v1, y = v1 - v2, area = sqrt(a + b + c) / 2 with the three
// squared cross-product components in the reference's order.
__global__ __launch_bounds__(PT_THREADS) void face_areas_kernel(int b, int nv, const float *verts, int nf,
                                                                 const int64_t *faces, float *areas)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= (int64_t)b * nf) return;
    const int mesh = (int)(i / nf);
    const int f = (int)(i - (int64_t)mesh * nf);
    const float *V = verts + (size_t)mesh * nv * 3;
    const V3 v0 = ld3(V + 3 * faces[3 * (size_t)f + 0]);
    const V3 v1 = ld3(V + 3 * faces[3 * (size_t)f + 1]);
    const V3 v2 = ld3(V + 3 * faces[3 * (size_t)f + 2]);
    const V3 x = v0 - v1, y = v1 - v2;
    const float ca = x.y * y.z - x.z * y.y;
    const float cb = x.z * y.x - x.x * y.z;
    const float cc = x.x * y.y - x.y * y.x;
    const float s = (ca * ca + cb * cb) + cc * cc;
    areas[i] = sqrtf(s) / 2.f;
}

// ------------------------------------------------------------- random draws ----
// Area-weighted face choice with replacement + the two barycentric uniforms, from pre-generated
// uniforms (the reference: a python loop of B torch.multinomial calls plus two sample_n calls,
// utils.py:604-612, 627-628).  One workgroup = one (mesh, 1024-sample chunk): it rebuilds the
// mesh's face-area CDF in LDS (areas as in face_areas_kernel, block-wide inclusive scan in a fixed
// order) and every thread inverts it for its sample with a binary search: face = first f with
// cdf[f] > U0 * total.  Same distribution as multinomial(areas, num, replacement=True); the
// stream of draws is our own (torch's generator stream is not reproducible across devices either).
__global__ __launch_bounds__(DRAW_THREADS) void draw_samples_kernel(int nv, const float *verts, int nf,
                                                                     const int64_t *faces, int num,
                                                                     const float *uniforms, int64_t plane,
                                                                     unsigned long long *rng_state,
                                                                     int64_t *choices, float *u, float *v, float *points)
{
    draw_samples_body<false>(blockIdx.x, blockIdx.y, (unsigned long long)gridDim.x * gridDim.y, nv, verts, nf, faces, num, uniforms,
                             plane, rng_state, choices, u, v, points, DrawSort{});
}

// -------------------------------------------------------------- face sampling ----
struct SampleArgs {
    const float *verts;
    const int64_t *faces, *choices;
    const float *u, *v;
    int b, nv, nf, num;
};

// points = ((1-u)*x + (u*(1-v))*y) + (u*v)*z   (utils.py:630, torch's evaluation order)
__global__ __launch_bounds__(PT_THREADS) void sample_fwd_kernel(SampleArgs a, float *points)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= (int64_t)a.b * a.num) return;
    const int mesh = (int)(i / a.num);
    const int64_t f = a.choices[i];
    const float *V = a.verts + (size_t)mesh * a.nv * 3;
    const V3 x = ld3(V + 3 * a.faces[3 * f + 0]);
    const V3 y = ld3(V + 3 * a.faces[3 * f + 1]);
    const V3 z = ld3(V + 3 * a.faces[3 * f + 2]);
    const float u = a.u[i], v = a.v[i];
    const float w0 = 1.f - u;
    const float w1 = u * (1.f - v);
    const float w2 = u * v;
    const V3 p = (x * w0 + y * w1) + z * w2;
    points[3 * i + 0] = p.x;
    points[3 * i + 1] = p.y;
    points[3 * i + 2] = p.z;
}

__global__ __launch_bounds__(PT_THREADS) void sample_bwd_kernel(SampleArgs a, const float *grad_points,
                                                                 float *grad_verts)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= (int64_t)a.b * a.num) return;
    const int mesh = (int)(i / a.num);
    const int64_t f = a.choices[i];
    float *G = grad_verts + (size_t)mesh * a.nv * 3;
    const V3 g = ld3(grad_points + 3 * i);
    const float u = a.u[i], v = a.v[i];
    atomic_add3(G + 3 * a.faces[3 * f + 0], g * (1.f - u));
    atomic_add3(G + 3 * a.faces[3 * f + 1], g * (u * (1.f - v)));
    atomic_add3(G + 3 * a.faces[3 * f + 2], g * (u * v));
}

// Fused backward of  coef * sum |dst[idx] - src|^2  THROUGH the sampling: the gradient of a sampled
// point never exists in memory, it is scattered straight into grad_verts with the point's
// barycentric weights (what autograd does in the reference through utils.py:615-631 + 454-462).
//   via_nn == 0: thread t is sampled point t;            diff = point[t]  - other[idx[t]]
//   via_nn == 1: thread t is a point of `other`;         diff = point[si] - other[t],  si = idx[t]
//                (the second Chamfer direction of batch_point_to_point, utils.py:417)
__device__ __forceinline__ void sample_chamfer_bwd_item(int64_t i, const SampleArgs &a, const float *points, int n_other,
                                                        const float *other, const int *idx, int via_nn,
                                                        const float *coef_dev, float coef_host, float *grad_verts)
{
    const int count = via_nn ? n_other : a.num;
    const int mesh = (int)(i / count);
    const float coef = 2.f * coef_host * (coef_dev ? coef_dev[0] : 1.f);
    int64_t si, oi; // sampled point / other point (flattened over the batch)
    if (via_nn) {
        si = (int64_t)mesh * a.num + idx[i];
        oi = i;
    } else {
        si = i;
        oi = (int64_t)mesh * n_other + idx[i];
    }
    const V3 g = (ld3(points + 3 * si) - ld3(other + 3 * oi)) * coef;
    const int64_t f = a.choices[si];
    const float u = a.u[si], v = a.v[si];
    float *G = grad_verts + (size_t)mesh * a.nv * 3;
    atomic_add3(G + 3 * a.faces[3 * f + 0], g * (1.f - u));
    atomic_add3(G + 3 * a.faces[3 * f + 1], g * (u * (1.f - v)));
    atomic_add3(G + 3 * a.faces[3 * f + 2], g * (u * v));
}

__global__ __launch_bounds__(PT_THREADS) void sample_chamfer_bwd_kernel(SampleArgs a, const float *points,
                                                                         int n_other, const float *other,
                                                                         const int *idx, int via_nn,
                                                                         const float *coef_dev, float coef_host,
                                                                         float *grad_verts)
{
    const int count = via_nn ? n_other : a.num;
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= (int64_t)a.b * count) return;
    sample_chamfer_bwd_item(i, a, points, n_other, other, idx, via_nn, coef_dev, coef_host, grad_verts);
}


// ------------------------------------------------------- Chamfer gather loss ----
// d/dsrc, d/ddst of  sum_j |dst[idx[j]] - src[j]|^2  scaled by coef (utils.py:416-417, 462).
__global__ __launch_bounds__(PT_THREADS) void chamfer_grad_kernel(int b, int n, const float *src, int m,
                                                                   const float *dst, const int *idx,
                                                                   const float *coef_dev, float coef_host,
                                                                   float *grad_src, int accumulate, float *grad_dst)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= (int64_t)b * n) return;
    const int mesh = (int)(i / n);
    const float coef = 2.f * coef_host * (coef_dev ? coef_dev[0] : 1.f);
    const size_t t = (size_t)mesh * m + idx[i];
    const V3 diff = ld3(src + 3 * i) - ld3(dst + 3 * t);
    const V3 g = diff * coef;
    if (grad_src) {
        if (accumulate) {
            grad_src[3 * i + 0] += g.x;
            grad_src[3 * i + 1] += g.y;
            grad_src[3 * i + 2] += g.z;
        } else {
            grad_src[3 * i + 0] = g.x;
            grad_src[3 * i + 1] = g.y;
            grad_src[3 * i + 2] = g.z;
        }
    }
    if (grad_dst) atomic_add3(grad_dst + 3 * t, g * -1.f);
}

// -------------------------------------------------- point-to-surface loss ----
struct P2TArgs {
    const float *xyz, *verts;
    const int64_t *faces;
    const int *option, *index;
    int b, n, nv, nf;
};

__global__ __launch_bounds__(PT_THREADS) void p2tri_fwd_kernel(P2TArgs a, float *sqdist, float *closest,
                                                                float *weights)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= (int64_t)a.b * a.n) return;
    const int mesh = (int)(i / a.n);
    const float *V = a.verts + (size_t)mesh * a.nv * 3;
    const int64_t f = a.index[i];
    const V3 A = ld3(V + 3 * a.faces[3 * f + 0]);
    const V3 B = ld3(V + 3 * a.faces[3 * f + 1]);
    const V3 C = ld3(V + 3 * a.faces[3 * f + 2]);
    const V3 p = ld3(a.xyz + 3 * i);
    V3 w;
    const V3 q = geom::closest_on_triangle(p, A, B, C, a.option[i], w);
    const V3 d = q - p;
    sqdist[i] = geom::dot3(d, d);
    if (closest) {
        closest[3 * i + 0] = q.x;
        closest[3 * i + 1] = q.y;
        closest[3 * i + 2] = q.z;
    }
    if (weights) {
        weights[3 * i + 0] = w.x;
        weights[3 * i + 1] = w.y;
        weights[3 * i + 2] = w.z;
    }
}

// d/dX_k sum |q - p|^2 = 2 w_k (q - p): q is an exact minimiser along every free parameter
// (edge parameter, plane foot), so the parameter derivatives vanish (envelope theorem); this
// equals the reference autograd through calc_point_to_line.
__device__ __forceinline__ void p2tri_bwd_item(int64_t i, int n, const float *xyz, int nv, const int64_t *faces,
                                               const int *index, const float *closest, const float *weights,
                                               const float *coef_dev, float coef_host, float *grad_verts)
{
    const int mesh = (int)(i / n);
    const float coef = 2.f * coef_host * (coef_dev ? coef_dev[0] : 1.f);
    const int64_t f = index[i];
    const V3 g = (ld3(closest + 3 * i) - ld3(xyz + 3 * i)) * coef;
    const V3 w = ld3(weights + 3 * i);
    float *G = grad_verts + (size_t)mesh * nv * 3;
    if (w.x != 0.f) atomic_add3(G + 3 * faces[3 * f + 0], g * w.x);
    if (w.y != 0.f) atomic_add3(G + 3 * faces[3 * f + 1], g * w.y);
    if (w.z != 0.f) atomic_add3(G + 3 * faces[3 * f + 2], g * w.z);
}

__global__ __launch_bounds__(PT_THREADS) void p2tri_bwd_kernel(int b, int n, const float *xyz, int nv,
                                                                const int64_t *faces, const int *index,
                                                                const float *closest, const float *weights,
                                                                const float *coef_dev, float coef_host,
                                                                float *grad_verts)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= (int64_t)b * n) return;
    p2tri_bwd_item(i, n, xyz, nv, faces, index, closest, weights, coef_dev, coef_host, grad_verts);
}

// Backward of batch_point_to_surface in ONE launch: workgroups [0, split) scatter the Chamfer term of the sampled
// points, the rest the point-to-triangle term -- both into the same zeroed grad_verts.
struct SurfaceBwdArgs {
    SampleArgs sample;
    const float *points, *gt;
    const int *idx_g, *index;
    const float *closest, *weights, *coef_dev;
    float coef_sample, coef_tri;
    float *grad_verts;
    int n_gt;
    unsigned split;
};
__global__ __launch_bounds__(PT_THREADS) void surface_bwd_kernel(SurfaceBwdArgs a)
{
    if (blockIdx.x < a.split) {
        const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
        if (i < (int64_t)a.sample.b * a.sample.num)
            sample_chamfer_bwd_item(i, a.sample, a.points, a.n_gt, a.gt, a.idx_g, 0, a.coef_dev, a.coef_sample, a.grad_verts);
    } else {
        const int64_t i = (int64_t)(blockIdx.x - a.split) * PT_THREADS + threadIdx.x;
        if (i < (int64_t)a.sample.b * a.n_gt)
            p2tri_bwd_item(i, a.n_gt, a.gt, a.sample.nv, a.sample.faces, a.index, a.closest, a.weights, a.coef_dev, a.coef_tri,
                           a.grad_verts);
    }
}

// ------------------------------------------------------------ deterministic sum ----
// One workgroup, fixed traversal and a fixed shuffle/LDS tree: the same input always
// produces the same bits (the reference's torch.mean has no such guarantee either way).
constexpr int SUM_THREADS = 1024;
__global__ __launch_bounds__(SUM_THREADS) void sum_kernel(int64_t n, const float *x, float scale, float *out)
{
    __shared__ float partial[SUM_THREADS / GEOM_WAVE];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += SUM_THREADS) acc += x[i];
    for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, GEOM_WAVE);
    if ((threadIdx.x & (GEOM_WAVE - 1)) == 0) partial[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < GEOM_WAVE) {
        float v = threadIdx.x < SUM_THREADS / GEOM_WAVE ? partial[threadIdx.x] : 0.f;
        for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, GEOM_WAVE);
        if (threadIdx.x == 0) out[0] = v * scale;
    }
}

constexpr int SUM_BATCH = 8;
__device__ __forceinline__ float thread_sum(int64_t n, const float *x)
{
    float acc = 0.f;
    const bool vec = (((uintptr_t)x) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (int64_t base = 0; base < n4; base += (int64_t)SUM_BATCH * SUM_THREADS) {
        float4 v[SUM_BATCH];
#pragma unroll
        for (int j = 0; j < SUM_BATCH; ++j) {
            const int64_t i = base + threadIdx.x + (int64_t)j * SUM_THREADS;
            v[j] = i < n4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < SUM_BATCH; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += SUM_THREADS) acc += x[i]; // tail / unaligned input
    return acc;
}

// out[0] = s1 * sum(x1) + s2 * sum(x2): the whole  (dist_1 + dist_2) * 3000  of utils.py:420/484
// in one launch, same fixed reduction tree per segment.
__global__ __launch_bounds__(SUM_THREADS) void sum2_kernel(int64_t n1, const float *x1, float s1, int64_t n2,
                                                            const float *x2, float s2, float *out, float *clear,
                                                            int64_t clear_count)
{
    __shared__ float partial[2][SUM_THREADS / GEOM_WAVE];
    // optional side job: zero the buffer the backward will accumulate into (saves the backward a fill launch --
    // about 5 us inside a captured graph, where this workgroup's stores are free)
    for (int64_t i = threadIdx.x; i < clear_count; i += SUM_THREADS) clear[i] = 0.f;
    // One workgroup: the sum is a latency chain unless many loads are in flight.  Each thread issues up to
    // SUM_BATCH float4 loads per segment back to back (48 000 floats = 12 per thread: one round trip), then
    // adds them in a fixed order -- the association is static, so the result stays bit-reproducible.
    float a1 = thread_sum(n1, x1), a2 = thread_sum(n2, x2);
    for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) {
        a1 += __shfl_down(a1, off, GEOM_WAVE);
        a2 += __shfl_down(a2, off, GEOM_WAVE);
    }
    if ((threadIdx.x & (GEOM_WAVE - 1)) == 0) {
        partial[0][threadIdx.x >> 6] = a1;
        partial[1][threadIdx.x >> 6] = a2;
    }
    __syncthreads();
    if (threadIdx.x < GEOM_WAVE) {
        float v1 = threadIdx.x < SUM_THREADS / GEOM_WAVE ? partial[0][threadIdx.x] : 0.f;
        float v2 = threadIdx.x < SUM_THREADS / GEOM_WAVE ? partial[1][threadIdx.x] : 0.f;
        for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) {
            v1 += __shfl_down(v1, off, GEOM_WAVE);
            v2 += __shfl_down(v2, off, GEOM_WAVE);
        }
        if (threadIdx.x == 0) out[0] = v1 * s1 + v2 * s2;
    }
}

inline dim3 pt_grid(int64_t count) { return dim3((unsigned)((count + PT_THREADS - 1) / PT_THREADS)); }

} // namespace

extern "C" int geom_face_areas_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                                   float *areas, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0) return GEOM_EINVAL;
    if (b == 0 || nf == 0) return 0;
    if (!verts || !faces || !areas) return GEOM_EINVAL;
    hipLaunchKernelGGL(face_areas_kernel, pt_grid((int64_t)b * nf), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), b, nv, verts, nf, faces, areas);
    return geom::launch_status();
}

extern "C" int geom_sample_faces_fwd_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                                         int num, const int64_t *choices, const float *u, const float *v,
                                         float *points, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0) return GEOM_EINVAL;
    if (b == 0 || num == 0) return 0;
    if (!verts || !faces || !choices || !u || !v || !points) return GEOM_EINVAL;
    SampleArgs a{verts, faces, choices, u, v, b, nv, nf, num};
    hipLaunchKernelGGL(sample_fwd_kernel, pt_grid((int64_t)b * num), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), a, points);
    return geom::launch_status();
}

extern "C" int geom_sample_faces_bwd_f32(int b, int nv, int nf, const int64_t *faces,
                                         int num, const int64_t *choices, const float *u, const float *v,
                                         const float *grad_points, float *grad_verts, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0) return GEOM_EINVAL;
    if (b == 0 || num == 0) return 0;
    if (!faces || !choices || !u || !v || !grad_points || !grad_verts) return GEOM_EINVAL;
    SampleArgs a{nullptr, faces, choices, u, v, b, nv, nf, num};
    hipLaunchKernelGGL(sample_bwd_kernel, pt_grid((int64_t)b * num), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), a, grad_points, grad_verts);
    return geom::launch_status();
}

extern "C" int geom_chamfer_grad_f32(int b, int n, const float *src, int m, const float *dst,
                                     const int *idx, const float *coef_dev, float coef_host,
                                     float *grad_src, int accumulate, float *grad_dst, void *stream)
{
    if (b < 0 || n < 0 || m < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (!src || !dst || !idx || (!grad_src && !grad_dst)) return GEOM_EINVAL;
    hipLaunchKernelGGL(chamfer_grad_kernel, pt_grid((int64_t)b * n), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), b, n, src, m, dst, idx, coef_dev, coef_host,
                       grad_src, accumulate, grad_dst);
    return geom::launch_status();
}

extern "C" int geom_p2tri_loss_fwd_f32(int b, int n, const float *xyz, int nv, const float *verts,
                                       int nf, const int64_t *faces, const int *option, const int *index,
                                       float *sqdist, float *closest, float *weights, void *stream)
{
    if (b < 0 || n < 0 || nv < 0 || nf < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (!xyz || !verts || !faces || !option || !index || !sqdist) return GEOM_EINVAL;
    P2TArgs a{xyz, verts, faces, option, index, b, n, nv, nf};
    hipLaunchKernelGGL(p2tri_fwd_kernel, pt_grid((int64_t)b * n), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), a, sqdist, closest, weights);
    return geom::launch_status();
}

extern "C" int geom_p2tri_loss_bwd_f32(int b, int n, const float *xyz, int nv, int nf, const int64_t *faces,
                                       const int *index, const float *closest, const float *weights,
                                       const float *coef_dev, float coef_host, float *grad_verts, void *stream)
{
    if (b < 0 || n < 0 || nv < 0 || nf < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (!xyz || !faces || !index || !closest || !weights || !grad_verts) return GEOM_EINVAL;
    hipLaunchKernelGGL(p2tri_bwd_kernel, pt_grid((int64_t)b * n), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), b, n, xyz, nv, faces, index, closest, weights,
                       coef_dev, coef_host, grad_verts);
    return geom::launch_status();
}

extern "C" int geom_sum_f32(int64_t n, const float *x, float scale, float *out, void *stream)
{
    if (n < 0 || !out || (n > 0 && !x)) return GEOM_EINVAL;
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(SUM_THREADS), 0, static_cast<hipStream_t>(stream), n, x, scale, out);
    return geom::launch_status();
}

extern "C" int geom_sample_chamfer_bwd_f32(int b, int nv, int nf, const int64_t *faces, int num,
                                           const int64_t *choices, const float *u, const float *v,
                                           const float *points, int n_other, const float *other,
                                           const int *idx, int via_nn, const float *coef_dev, float coef_host,
                                           float *grad_verts, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0 || n_other < 0) return GEOM_EINVAL;
    const int count = via_nn ? n_other : num;
    if (b == 0 || count == 0) return 0;
    if (!faces || !choices || !u || !v || !points || !other || !idx || !grad_verts) return GEOM_EINVAL;
    SampleArgs a{nullptr, faces, choices, u, v, b, nv, nf, num};
    hipLaunchKernelGGL(sample_chamfer_bwd_kernel, pt_grid((int64_t)b * count), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), a, points, n_other, other, idx, via_nn, coef_dev, coef_host,
                       grad_verts);
    return geom::launch_status();
}

extern "C" int geom_surface_loss_bwd_f32(int b, int nv, int nf, const int64_t *faces, int num, const int64_t *choices,
                                         const float *u, const float *v, const float *points, int n_gt, const float *gt,
                                         const int *idx_g, const int *index, const float *closest, const float *weights,
                                         const float *coef_dev, float coef_sample, float coef_tri, float *grad_verts,
                                         void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0 || n_gt < 0) return GEOM_EINVAL;
    if (b == 0 || (num == 0 && n_gt == 0)) return 0;
    if (!faces || !grad_verts || (num > 0 && (!choices || !u || !v || !points || !gt || !idx_g || n_gt == 0)) ||
        (n_gt > 0 && (!gt || !index || !closest || !weights)))
        return GEOM_EINVAL;
    const int64_t blocks_s = ((int64_t)b * num + PT_THREADS - 1) / PT_THREADS;
    const int64_t blocks_t = ((int64_t)b * n_gt + PT_THREADS - 1) / PT_THREADS;
    if (blocks_s + blocks_t > 0x7fffffffLL) return GEOM_ETOOBIG;
    SurfaceBwdArgs a{{nullptr, faces, choices, u, v, b, nv, nf, num}, points, gt, idx_g, index, closest, weights, coef_dev,
                     coef_sample, coef_tri, grad_verts, n_gt, (unsigned)blocks_s};
    hipLaunchKernelGGL(surface_bwd_kernel, dim3((unsigned)(blocks_s + blocks_t)), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), a);
    return geom::launch_status();
}

extern "C" int geom_sum2_f32(int64_t n1, const float *x1, float scale1, int64_t n2, const float *x2, float scale2,
                             float *out, float *clear, int64_t clear_count, void *stream)
{
    if (n1 < 0 || n2 < 0 || !out || (n1 > 0 && !x1) || (n2 > 0 && !x2)) return GEOM_EINVAL;
    if (clear_count < 0 || (clear_count > 0 && !clear)) return GEOM_EINVAL;
    hipLaunchKernelGGL(sum2_kernel, dim3(1), dim3(SUM_THREADS), 0, static_cast<hipStream_t>(stream), n1, x1, scale1, n2,
                       x2, scale2, out, clear, clear_count);
    return geom::launch_status();
}

extern "C" int geom_draw_samples_f32(int b, int nv, const float *verts, int nf, const int64_t *faces, int num,
                                     const float *uniforms, int64_t *choices, float *u, float *v, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0) return GEOM_EINVAL;
    if (nf > DRAW_MAX_FACES) return GEOM_EUNSUPPORTED;
    if (b == 0 || num == 0) return 0;
    if (nf == 0 || !verts || !faces || !uniforms || !choices || !u || !v) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    hipLaunchKernelGGL(draw_samples_kernel, dim3((num + DRAW_THREADS - 1) / DRAW_THREADS, b), dim3(DRAW_THREADS), 0,
                       static_cast<hipStream_t>(stream), nv, verts, nf, faces, num, uniforms, (int64_t)b * num,
                       static_cast<unsigned long long *>(nullptr), choices, u, v, static_cast<float *>(nullptr));
    return geom::launch_status();
}

extern "C" int geom_draw_samples_rng_f32(int b, int nv, const float *verts, int nf, const int64_t *faces, int num,
                                         uint64_t *rng_state, int64_t *choices, float *u, float *v, float *points,
                                         void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0) return GEOM_EINVAL;
    if (nf > DRAW_MAX_FACES) return GEOM_EUNSUPPORTED;
    if (b == 0 || num == 0) return 0;
    if (nf == 0 || !verts || !faces || !rng_state || !choices || !u || !v) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(draw_samples_kernel, dim3((num + DRAW_THREADS - 1) / DRAW_THREADS, b), dim3(DRAW_THREADS), 0, s,
                       nv, verts, nf, faces, num, static_cast<const float *>(nullptr), (int64_t)b * num,
                       reinterpret_cast<unsigned long long *>(rng_state), choices, u, v, points);
    return geom::launch_status();
}

// ------------------------------------------------------------ vertex head ----
// pos[b,v,:] = base[b,v,:] + scale * feat[b,v,:3]   and its adjoint  grad_feat = [scale*grad_pos | 0]:
// the coordinate update of a deformation stage (GEOMetrics.py:121, 126, 131: positions + block output)
// when the coordinates are the leading channels of a wider feature tensor.
namespace {
__global__ __launch_bounds__(PT_THREADS) void vertex_head_fwd_kernel(int64_t rows, int c, const float *base,
                                                                      const float *feat, float scale, float *pos)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x;
    if (i >= rows * 3) return;
    const int64_t r = i / 3;
    pos[i] = base[i] + scale * feat[r * c + (i - 3 * r)];
}
__global__ __launch_bounds__(PT_THREADS) void vertex_head_bwd_kernel(int64_t rows, int c, const float *grad_pos,
                                                                      float scale, float *grad_feat)
{
    const int64_t i = (int64_t)blockIdx.x * PT_THREADS + threadIdx.x; // one thread per float4 of grad_feat
    const int c4 = c >> 2;
    if (i >= rows * c4) return;
    const int64_t r = i / c4;
    const int q = (int)(i - r * c4);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q == 0) g = make_float4(scale * grad_pos[3 * r], scale * grad_pos[3 * r + 1], scale * grad_pos[3 * r + 2], 0.f);
    reinterpret_cast<float4 *>(grad_feat)[i] = g;
}
} // namespace

extern "C" int geom_vertex_head_fwd_f32(int64_t rows, int c, const float *base, const float *feat, float scale,
                                        float *pos, void *stream)
{
    if (rows < 0 || c < 3) return GEOM_EINVAL;
    if (rows == 0) return 0;
    if (!base || !feat || !pos) return GEOM_EINVAL;
    hipLaunchKernelGGL(vertex_head_fwd_kernel, pt_grid(rows * 3), dim3(PT_THREADS), 0, static_cast<hipStream_t>(stream),
                       rows, c, base, feat, scale, pos);
    return geom::launch_status();
}

extern "C" int geom_vertex_head_bwd_f32(int64_t rows, int c, const float *grad_pos, float scale, float *grad_feat,
                                        void *stream)
{
    if (rows < 0 || c < 4 || (c & 3)) return GEOM_EINVAL;
    if (rows == 0) return 0;
    if (!grad_pos || !grad_feat || ((uintptr_t)grad_feat & 15)) return GEOM_EINVAL;
    hipLaunchKernelGGL(vertex_head_bwd_kernel, pt_grid(rows * (c >> 2)), dim3(PT_THREADS), 0,
                       static_cast<hipStream_t>(stream), rows, c, grad_pos, scale, grad_feat);
    return geom::launch_status();
}
