// 0N-GCN aggregation for gfx950: out = [ A . S[:, :k] | S[:, k:] ] + bias, activation fused.
//
// The reference multiplies by the DENSE row-normalised adjacency (layers.py:37, 111, 146:
// torch.mm/matmul(adj, support[..., :k]) -- V^2*k MACs and a V^2*4-byte read per layer with
// 99.7 % zeros at V=2562) and then torch.cat's the pass-through columns and adds the bias in
// two more passes.  Here one kernel reads `support` once and writes `out` once: a CSR row
// gather (6-7 neighbours on an icosphere, up to 33 on the reference's 482.obj poles) for
// the first k columns, a straight copy for the rest, bias and activation in the epilogue.
// The backward is the same kernel on CSR^T (the normalised adjacency is not symmetric) with
// the activation derivative folded into the read.
//
// One thread owns VEC consecutive columns of one row, so a wave reads/writes contiguous
// 256 B - 1 KiB runs; neighbour rows of the k-slice are re-read through L2 (the slice is
// b*V*k*4 bytes, 5.2 MB at the BASELINE shard).  HBM-bound: algorithmic bytes = read
// support + write out (+ CSR), see DESIGN.md.
#include "geom_common.h"

namespace {

constexpr int GCN_THREADS = 256;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2 };

struct GcnArgs {
    const int *rowptr, *col;
    const float *val;
    const float *x;     // support (forward) or grad_out (backward)
    const float *bias;  // forward only, may be null
    const float *saved; // backward only: forward output, for the activation derivative
    float *y;
    int64_t rows;       // b * nv
    int nv, c, k;
};

template <int ACT>
__device__ __forceinline__ float act_fwd(float v)
{
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_ELU) return v > 0.f ? v : expm1f(v);
    return v;
}

// derivative expressed through the saved OUTPUT (what torch's relu/elu backward use)
template <int ACT>
__device__ __forceinline__ float act_bwd(float g, float out)
{
    if (ACT == ACT_RELU) return out > 0.f ? g : 0.f;
    if (ACT == ACT_ELU) return out > 0.f ? g : g * (out + 1.f);
    return g;
}

template <int VEC>
struct Pack;
template <>
struct Pack<4> {
    float4 v;
    __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const float4 *>(p); }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = v; }
    __device__ __forceinline__ float &at(int i) { return (&v.x)[i]; }
};
template <>
struct Pack<1> {
    float v;
    __device__ __forceinline__ void load(const float *p) { v = *p; }
    __device__ __forceinline__ void store(float *p) const { *p = v; }
    __device__ __forceinline__ float &at(int) { return v; }
};

template <int VEC, int ACT, bool BACKWARD>
__global__ __launch_bounds__(GCN_THREADS) void zn_aggregate_kernel(GcnArgs a)
{
    const int groups = a.c / VEC;
    const int64_t tid = (int64_t)blockIdx.x * GCN_THREADS + threadIdx.x;
    if (tid >= a.rows * groups) return;
    const int64_t row = tid / groups;
    const int g = (int)(tid - row * groups);
    const int c0 = g * VEC;
    const int64_t mesh_row0 = (row / a.nv) * a.nv; // first row of this mesh
    const int r = (int)(row - mesh_row0);

    Pack<VEC> acc;
    if (c0 < a.k) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.at(i) = 0.f;
        const int e0 = a.rowptr[r], e1 = a.rowptr[r + 1];
        for (int e = e0; e < e1; ++e) {
            const int64_t nb = mesh_row0 + a.col[e];
            const float w = a.val[e];
            Pack<VEC> s;
            s.load(a.x + nb * a.c + c0);
            if (BACKWARD && ACT != ACT_NONE) {
                Pack<VEC> o;
                o.load(a.saved + nb * a.c + c0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) s.at(i) = act_bwd<ACT>(s.at(i), o.at(i));
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc.at(i) += w * s.at(i);
        }
    } else {
        acc.load(a.x + row * a.c + c0);
        if (BACKWARD && ACT != ACT_NONE) {
            Pack<VEC> o;
            o.load(a.saved + row * a.c + c0);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc.at(i) = act_bwd<ACT>(acc.at(i), o.at(i));
        }
    }
    if (!BACKWARD) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float v = acc.at(i);
            if (a.bias) v += a.bias[c0 + i];
            acc.at(i) = act_fwd<ACT>(v);
        }
    }
    acc.store(a.y + row * a.c + c0);
}

template <int VEC, bool BACKWARD>
int launch(const GcnArgs &a, int act, void *stream)
{
    const int64_t threads = a.rows * (a.c / VEC);
    dim3 grid((unsigned)((threads + GCN_THREADS - 1) / GCN_THREADS)), block(GCN_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (act) {
    case ACT_NONE: hipLaunchKernelGGL((zn_aggregate_kernel<VEC, ACT_NONE, BACKWARD>), grid, block, 0, s, a); break;
    case ACT_RELU: hipLaunchKernelGGL((zn_aggregate_kernel<VEC, ACT_RELU, BACKWARD>), grid, block, 0, s, a); break;
    case ACT_ELU: hipLaunchKernelGGL((zn_aggregate_kernel<VEC, ACT_ELU, BACKWARD>), grid, block, 0, s, a); break;
    default: return GEOM_EINVAL;
    }
    return geom::launch_status();
}

template <bool BACKWARD>
int dispatch(GcnArgs a, int b, int act, void *stream)
{
    if (b < 0 || a.nv < 0 || a.c < 0 || a.k < 0 || a.k > a.c) return GEOM_EINVAL;
    if (b == 0 || a.nv == 0 || a.c == 0) return 0;
    if (!a.rowptr || !a.x || !a.y || (a.k > 0 && (!a.col || !a.val))) return GEOM_EINVAL;
    if (BACKWARD && act != ACT_NONE && !a.saved) return GEOM_EINVAL;
    a.rows = (int64_t)b * a.nv;
    const bool vec4 = (a.c % 4 == 0) && (a.k % 4 == 0) && (((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.saved) % 16 == 0);
    if ((a.rows * (vec4 ? a.c / 4 : a.c) + GCN_THREADS - 1) / GCN_THREADS > 0x7fffffffLL) return GEOM_ETOOBIG;
    return vec4 ? launch<4, BACKWARD>(a, act, stream) : launch<1, BACKWARD>(a, act, stream);
}

} // namespace

extern "C" int geom_zn_gcn_aggregate_fwd_f32(int b, int nv, int c, int k, const int *rowptr, const int *col,
                                             const float *val, const float *support, const float *bias,
                                             int act, float *out, void *stream)
{
    GcnArgs a{rowptr, col, val, support, bias, nullptr, out, 0, nv, c, k};
    return dispatch<false>(a, b, act, stream);
}

extern "C" int geom_zn_gcn_aggregate_bwd_f32(int b, int nv, int c, int k, const int *rowptrT, const int *colT,
                                             const float *valT, const float *grad_out, const float *out,
                                             int act, float *grad_support, void *stream)
{
    GcnArgs a{rowptrT, colT, valT, grad_out, nullptr, out, grad_support, 0, nv, c, k};
    return dispatch<true>(a, b, act, stream);
}
