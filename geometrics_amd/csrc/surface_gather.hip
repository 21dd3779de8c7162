// Backward of the sampled-surface losses (batch_point_to_point / batch_point_to_surface, reference utils.py:393-502
// under autograd) as a GATHER: every vertex sums the contributions of the sampled points and gt points that landed on
// its incident faces, in a fixed order.
//
// The scatter formulation (sample_loss.hip: one thread per point, nine fp32 atomics into a zeroed grad_verts) costs
// 26 us for 432 000 atomics at the BASELINE shard, needs a zero-fill, and adds in arrival order, so the last bits of
// the gradient change from run to run.  Here the points are counting-sorted by face:
//   bin     one thread per point: its gradient vector (point - partner) * coefficient and its three corner weights are
//           computed here, fully parallel and coalesced, and stored as two float4 records; slot =
//           atomicAdd(count[mesh][face], 1) (integer, almost no contention) is remembered per point;
//   order   one workgroup per mesh, everything in LDS: exclusive scan of the face counts -> offsets; every point id is
//           dropped at offset[face] + slot; every face's segment is then put in ASCENDING ID ORDER (arrival order is
//           not reproducible, id order is) by ranking every id against its segment.  Any distribution is handled
//           exactly -- 1.2 points per face on the 5120-face
//           BASELINE mesh, 6 on average and dozens on the large faces of the reference's 960-face training template
//           (the first version kept 16 slots per face plus an overflow list and fell back to scanning all of a mesh's
//           points per (face, corner): 691 us per call at that shape, 20 us here);
//   gather  eight lanes per (mesh, vertex), one incident (face, corner) each from a static CSR built once per face
//           list: walk the face's segment, eight records in flight, accumulate with this corner's barycentric weight;
//           the lanes' partial sums are folded in lane order.
// grad_verts is written once per element (no zero-fill, no float atomics) and is bit-reproducible.
// The per-face counters must be zero on entry; the order pass leaves them zero again.
#include "geom_common.h"
#include "tri_math.h"
#include "surface_layout.h"
#include "finalize_body.h"

namespace {

using geom::V3;

constexpr int SGA_THREADS = 256;
constexpr int ORD_THREADS = 1024;
constexpr int ORD_WAVES = ORD_THREADS / GEOM_WAVE;
constexpr int VTX_LANES = 8;    // lanes that share a vertex in the gather: one incident face each, then an ordered fold
// dynamic LDS one workgroup may take for the ordering pass: what the device grants (160 KiB on gfx950) minus 10 KiB for the
// kernels' static arrays; read from the device once, so a part with less LDS answers GEOM_EUNSUPPORTED instead of failing
// the launch
inline size_t ord_lds_limit()
{
    static size_t limit = 0;
    if (!limit) {
        int dev = 0;
        hipDeviceProp_t prop;
        size_t have = 64 * 1024;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            have = prop.sharedMemPerBlock;
            if (prop.sharedMemPerBlockOptin > have) have = prop.sharedMemPerBlockOptin;
            if (prop.maxSharedMemoryPerMultiProcessor > have) have = prop.maxSharedMemoryPerMultiProcessor;
        }
        limit = have > 16 * 1024 ? have - 10 * 1024 : have;
        if (limit > 150 * 1024) limit = 150 * 1024;
    }
    return limit;
}
#define ORD_LDS_LIMIT (ord_lds_limit())

using geom_finalize::OTHER_NONE;
using geom_finalize::OTHER_NN;
using geom_finalize::OTHER_TRI;
using geom_finalize::FinalizeArgs;
using geom_finalize::ld3;

struct GatherArgs {
    const int *vf_ptr;  // [nv+1]
    const int *vf_item; // [3*nf]  (face << 2) | corner, ascending per vertex
    const int64_t *choices;
    const float *u, *v, *points, *gt;
    const int *idx_g;   // [b,num]   nearest gt point of each sampled point
    const int *idx_p;   // [b,n_gt]  nearest sampled point of each gt point   (OTHER_NN)
    const int *index;   // [b,n_gt]  winning triangle of each gt point        (OTHER_TRI)
    const float *closest, *weights; // [b,n_gt,3] (OTHER_TRI)
    const float *coef_dev;
    float coef_sample, coef_other;
    int b, nv, nf, num, n_gt, other, per; // per = points per mesh that take part (num [+ n_gt])
    int *counts; // [b,nf]     points per face (zero on entry, zero again after the order pass)
    int *off;    // [b,nf+1]   exclusive offsets of the face segments
    int *slot;   // [b,per]    arrival rank of the point inside its face
    int *pface;  // [b,per]    face of the point, -1 = none
    int *seg;    // [b,per]    point ids, face by face, ascending inside a face
    float4 *rec; // [b,per,2]  {gradient vector, skip-zero-weights flag}, {w0, w1, w2, -}
    float *grad_verts;
};


__global__ __launch_bounds__(SGA_THREADS) void surface_bin_kernel(GatherArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * SGA_THREADS + threadIdx.x;
    if (i >= (int64_t)a.b * a.per) return;
    const int mesh = (int)(i / a.per), id = (int)(i - (int64_t)mesh * a.per);
    const float scale = 2.f * (a.coef_dev ? a.coef_dev[0] : 1.f);
    // the point's gradient vector and corner weights, exactly as the scatter kernels form them
    int64_t f, sp = -1;
    V3 g;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    float skip_zero = 0.f;
    if (id < a.num) {
        sp = (int64_t)mesh * a.num + id;
        f = a.choices[sp];
        g = (ld3(a.points + 3 * sp) - ld3(a.gt + 3 * ((int64_t)mesh * a.n_gt + a.idx_g[sp]))) * (scale * a.coef_sample);
    } else {
        const int64_t o = (int64_t)mesh * a.n_gt + (id - a.num);
        if (a.other == OTHER_TRI) {
            f = a.index[o];
            g = (ld3(a.closest + 3 * o) - ld3(a.gt + 3 * o)) * (scale * a.coef_other);
            w = make_float4(a.weights[3 * o], a.weights[3 * o + 1], a.weights[3 * o + 2], 0.f);
            skip_zero = 1.f; // the scatter does not touch a corner whose weight is exactly zero
        } else {
            sp = (int64_t)mesh * a.num + a.idx_p[o];
            f = a.choices[sp];
            g = (ld3(a.points + 3 * sp) - ld3(a.gt + 3 * o)) * (scale * a.coef_other);
        }
    }
    if (sp >= 0) {
        const float u = a.u[sp], v = a.v[sp];
        w = make_float4(1.f - u, u * (1.f - v), u * v, 0.f);
    }
    a.rec[2 * i + 0] = make_float4(g.x, g.y, g.z, skip_zero);
    a.rec[2 * i + 1] = w;
    const bool on_mesh = f >= 0 && f < a.nf; // else: contributes nowhere (the scatter would have faulted)
    a.pface[i] = on_mesh ? (int)f : -1;
    if (on_mesh) a.slot[i] = atomicAdd(&a.counts[(int64_t)mesh * a.nf + f], 1);
}

// One workgroup per mesh.  LDS: off[nf+1] | seg[per] | wave totals.
__global__ __launch_bounds__(ORD_THREADS) void surface_order_kernel(GatherArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int ord_lds[];
    int *off = ord_lds;
    int *seg = off + (a.nf + 1);
    int *wave_total = seg + a.per;   // [ORD_WAVES]
    const int mesh = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & (GEOM_WAVE - 1), wave = tid >> 6;
    int *counts = a.counts + (int64_t)mesh * a.nf;

    // ---- exclusive scan of the face counts (consecutive faces per thread), counters re-armed on the way ----
    const int chunk = (a.nf + ORD_THREADS - 1) / ORD_THREADS;
    const int f0 = min(a.nf, tid * chunk), f1 = min(a.nf, f0 + chunk);
    int run = 0;
    for (int f = f0; f < f1; ++f) {
        const int c = counts[f];
        counts[f] = 0;
        off[f] = run; // local exclusive prefix, completed below
        run += c;
    }
    int incl = run;
    for (int d = 1; d < GEOM_WAVE; d <<= 1) {
        const int t = __shfl_up(incl, d, GEOM_WAVE);
        if (lane >= d) incl += t;
    }
    if (lane == GEOM_WAVE - 1) wave_total[wave] = incl;
    __syncthreads();
    int base = incl - run;
    for (int w = 0; w < wave; ++w) base += wave_total[w];
    for (int f = f0; f < f1; ++f) off[f] += base;
    if (tid == ORD_THREADS - 1) off[a.nf] = base + run;
    __syncthreads();
    int *g_off = a.off + (int64_t)mesh * (a.nf + 1);
    for (int f = tid; f <= a.nf; f += ORD_THREADS) g_off[f] = off[f];

    // ---- every point id at offset[face] + arrival slot ----
    const int64_t p0 = (int64_t)mesh * a.per;
    for (int i = tid; i < a.per; i += ORD_THREADS) {
        const int f = a.pface[p0 + i];
        if (f >= 0) seg[off[f] + a.slot[p0 + i]] = i;
    }
    __syncthreads();

    // ---- ascending ids inside every face.  One thread per POINT: its rank among the ids of its face's segment (ids
    //      are distinct, so the ranks are a permutation) is where it goes.  The segment is read from LDS with
    //      independent loads -- no dependent chain, whatever the segment length (an insertion sort per face was
    //      measured first: 42 us for the order pass, all of it LDS latency in the sort's inner loop). ----
    int *g_seg = a.seg + p0;
    for (int i = tid; i < a.per; i += ORD_THREADS) {
        const int f = a.pface[p0 + i];
        if (f < 0) continue;
        const int s0 = off[f], n = off[f + 1] - s0;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += seg[s0 + j] < i ? 1 : 0;
        g_seg[s0 + rank] = i;
    }
}

// gradient contribution of a point to corner c of its face, from the records of the bin pass
__device__ __forceinline__ V3 apply_record(float4 g, float4 w, int c)
{
    const float wc = c == 0 ? w.x : (c == 1 ? w.y : w.z);
    if (g.w != 0.f && wc == 0.f) return geom::mk(0.f, 0.f, 0.f);
    return geom::mk(g.x, g.y, g.z) * wc;
}

// sum of the contributions of the points on face f to its corner c, in ascending point-id order
__device__ __forceinline__ V3 face_sum(const GatherArgs &a, int mesh, int f, int c)
{
    V3 acc = geom::mk(0.f, 0.f, 0.f);
    const int *off = a.off + (int64_t)mesh * (a.nf + 1);
    const int s0 = off[f], n = off[f + 1] - s0;
    const int *ids = a.seg + (int64_t)mesh * a.per + s0;
    const float4 *rec = a.rec + 2 * (int64_t)mesh * a.per;
    for (int h = 0; h < n; h += 8) { // eight records in flight per round trip
        int id[8];
        float4 g[8], w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) id[k] = h + k < n ? ids[h + k] : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (h + k < n) g[k] = rec[2 * id[k]], w[k] = rec[2 * id[k] + 1];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (h + k < n) acc = acc + apply_record(g[k], w[k], c);
    }
    return acc;
}

// VTX_LANES lanes per vertex: lane j sums the faces j, j + 8, ... of the vertex's incident list (one face each on a
// triangle mesh of valence <= 8), then the partial sums are folded in lane order -- a fixed association, so the
// result does not depend on timing.
__global__ __launch_bounds__(SGA_THREADS) void surface_gather_kernel(GatherArgs a)
{
    const int t = blockIdx.x * SGA_THREADS + threadIdx.x;
    const int vtx = t / VTX_LANES, j = t % VTX_LANES;
    const int mesh = blockIdx.y;
    const bool live = vtx < a.nv;
    V3 acc = geom::mk(0.f, 0.f, 0.f);
    if (live) {
        const int e1 = a.vf_ptr[vtx + 1];
        for (int e = a.vf_ptr[vtx] + j; e < e1; e += VTX_LANES) {
            const int item = a.vf_item[e];
            acc = acc + face_sum(a, mesh, item >> 2, item & 3);
        }
    }
    // ordered fold: lane 0 <- ((((l0 + l1) + l2) + ...) + l7)
    V3 total = acc;
#pragma unroll
    for (int k = 1; k < VTX_LANES; ++k) {
        const V3 other = geom::mk(__shfl_down(acc.x, k, VTX_LANES), __shfl_down(acc.y, k, VTX_LANES),
                                  __shfl_down(acc.z, k, VTX_LANES));
        total = total + other;
    }
    if (live && j == 0) {
        float *G = a.grad_verts + ((int64_t)mesh * a.nv + vtx) * 3;
        G[0] = total.x;
        G[1] = total.y;
        G[2] = total.z;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward-side FINALIZE: one launch after the two scans that (a) reduces the loss, (b) prepares the backward.
//
// One workgroup per mesh (1024 threads), everything that is shared in LDS:
//   * every thread takes its points (ids tid, tid + 1024, ...), forms their gradient records -- (point - partner) *
//     coefficient and the three corner weights, two float4 per point -- and counts them into their face with an LDS
//     atomic, remembering the arrival slot;
//   * exclusive scan of the face counts -> offsets; ids dropped at offset + slot; every id is then ranked against its
//     face's segment (LDS, independent loads) and written to its ASCENDING place: the order the gather adds in is a
//     function of the data only, never of timing;
//   * one more workgroup reduces the two loss sums (fixed tree) meanwhile.
// The backward is then ONE launch (surface_vertex_gather_kernel).  Compared with binning in the backward (bin ->
// order -> gather, plus a one-workgroup loss reduction in the forward) this removes two launches and every global atomic.
template <bool REGS>
__global__ __launch_bounds__(ORD_THREADS) void surface_finalize_kernel(FinalizeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int ord_lds[];
    geom_finalize::surface_finalize_body<REGS, ORD_THREADS>(a, ord_lds, blockIdx.x);
}

struct VGatherArgs {
    const int *vf_ptr, *vf_item;
    const int *off, *seg;
    const float4 *rec;
    const float *grad; // upstream gradient of the loss (device scalar), may be null (= 1)
    float *grad_verts;
    int nv, nf, per;
    const int *status; // [b + 1] (surface_layout.h)
    int b;
    const float *mesh_weight; // [b] or null: mesh m's gradient times w[m] (the loss was formed with the same weights)
};

// The backward: eight lanes per (mesh, vertex), one incident (face, corner) each; a face's segment is walked in its
// stored (ascending id) order, eight records in flight; lanes folded in lane order; result scaled by 2 * upstream.
__global__ __launch_bounds__(SGA_THREADS) void surface_vertex_gather_kernel(VGatherArgs a)
{
    const int t = blockIdx.x * SGA_THREADS + threadIdx.x;
    const int vtx = t / VTX_LANES, j = t % VTX_LANES;
    const int mesh = blockIdx.y;
    const bool live = vtx < a.nv;
    V3 acc = geom::mk(0.f, 0.f, 0.f);
    // a mesh whose ordering (or the loss it belongs to) was given up by a finalize role of the scan launch: NaN, loudly --
    // never a walk through a half-built order
    const bool broken = a.status[mesh] != 0 || a.status[a.b] != 0;
    if (live && !broken) {
        const int *off = a.off + (int64_t)mesh * (a.nf + 1);
        const int *seg = a.seg + (int64_t)mesh * a.per;
        const float4 *rec = a.rec + 2 * (int64_t)mesh * a.per;
        const int e1 = a.vf_ptr[vtx + 1];
        for (int e = a.vf_ptr[vtx] + j; e < e1; e += VTX_LANES) {
            const int item = a.vf_item[e];
            const int f = item >> 2, c = item & 3;
            const int s0 = off[f], n = off[f + 1] - s0;
            for (int h = 0; h < n; h += 8) {
                int id[8];
                float4 g[8], w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) id[k] = h + k < n ? seg[s0 + h + k] : 0;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (h + k < n) g[k] = rec[2 * id[k]], w[k] = rec[2 * id[k] + 1];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (h + k < n) acc = acc + apply_record(g[k], w[k], c);
            }
        }
    }
    V3 total = acc;
#pragma unroll
    for (int k = 1; k < VTX_LANES; ++k) {
        const V3 other = geom::mk(__shfl_down(acc.x, k, VTX_LANES), __shfl_down(acc.y, k, VTX_LANES),
                                  __shfl_down(acc.z, k, VTX_LANES));
        total = total + other;
    }
    if (live && j == 0) {
        float s = broken ? __builtin_nanf("") : 2.f * (a.grad ? a.grad[0] : 1.f);
        if (a.mesh_weight) s *= a.mesh_weight[mesh];
        float *G = a.grad_verts + ((int64_t)mesh * a.nv + vtx) * 3;
        G[0] = total.x * s;
        G[1] = total.y * s;
        G[2] = total.z * s;
    }
}

inline size_t order_lds_bytes(int nf, int per) { return ((size_t)nf + 1 + per + ORD_WAVES + 4) * sizeof(int); }

} // namespace

// int32 words behind `counts` (zero on entry) and `lists` (scratch) for a batch of b meshes of nf faces
extern "C" int64_t geom_surface_bin_count_words(int b, int nf) { return b <= 0 || nf < 0 ? 0 : (int64_t)b * nf; }
// off [b,nf+1] | slot, pface, seg [b,per] each (rounded up so that the float4 point records behind stay 16-byte aligned)
static inline int64_t list_words(int b, int nf, int64_t per) { return (((int64_t)b * (nf + 1) + 3 * (int64_t)b * per) + 3) & ~3ll; }
extern "C" int64_t geom_surface_bin_list_words(int b, int nf, int num, int n_gt)
{
    if (b <= 0 || nf < 0 || num < 0 || n_gt < 0) return 0;
    const int64_t per = (int64_t)num + n_gt;
    return list_words(b, nf, per) + (int64_t)b * per * 8;
}

extern "C" int geom_surface_loss_bwd_gather_f32(int b, int nv, int nf, const int *vf_ptr, const int *vf_item, int num,
                                                const int64_t *choices, const float *u, const float *v,
                                                const float *points, int n_gt, const float *gt, const int *idx_g,
                                                const int *idx_p, const int *index, const float *closest,
                                                const float *weights, const float *coef_dev, float coef_sample,
                                                float coef_other, int *counts, int *lists, float *grad_verts, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0 || n_gt < 0) return GEOM_EINVAL;
    if (b == 0 || nv == 0) return 0;
    if (!vf_ptr || !vf_item || !grad_verts || !counts || !lists) return GEOM_EINVAL;
    if (num > 0 && (!choices || !u || !v || !points || !gt || !idx_g || n_gt == 0)) return GEOM_EINVAL;
    if (idx_p && index) return GEOM_EINVAL; // one kind of gt-side term at a time
    if (index && (!closest || !weights || !gt)) return GEOM_EINVAL;
    if (idx_p && (!gt || num == 0)) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    if ((uintptr_t)lists & 15) return GEOM_EINVAL;
    const int other = idx_p ? OTHER_NN : (index ? OTHER_TRI : OTHER_NONE);
    const int64_t per64 = (int64_t)num + (other != OTHER_NONE ? n_gt : 0);
    if (per64 > 0x3fffffff) return GEOM_ETOOBIG;
    const int per = (int)per64;
    // the order pass keeps a mesh's offsets and ids in LDS: beyond that the caller uses the scatter formulation
    if (order_lds_bytes(nf, per) > ORD_LDS_LIMIT) return GEOM_EUNSUPPORTED;
    // scratch layout sized for (num + n_gt) points per mesh whatever `other` is
    const int64_t cap = (int64_t)num + n_gt;
    int *off = lists;
    int *slot = off + (int64_t)b * (nf + 1);
    int *pface = slot + (int64_t)b * cap;
    int *seg = pface + (int64_t)b * cap;
    GatherArgs a{vf_ptr, vf_item, choices, u, v, points, gt, idx_g, idx_p, index, closest, weights, coef_dev, coef_sample,
                 coef_other, b, nv, nf, num, n_gt, other, per, counts, off, slot, pface, seg,
                 reinterpret_cast<float4 *>(lists + list_words(b, nf, cap)), grad_verts};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t items = (int64_t)b * per;
    if ((items + SGA_THREADS - 1) / SGA_THREADS > 0x7fffffffLL) return GEOM_ETOOBIG;
    if (items > 0)
        hipLaunchKernelGGL(surface_bin_kernel, dim3((unsigned)((items + SGA_THREADS - 1) / SGA_THREADS)), dim3(SGA_THREADS), 0, s, a);
    static const hipError_t lds_opt_in = hipFuncSetAttribute(reinterpret_cast<const void *>(surface_order_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ORD_LDS_LIMIT);
    (void)lds_opt_in; // more than 64 KiB of dynamic LDS needs the opt-in; a refusal shows up as a launch error below
    hipLaunchKernelGGL(surface_order_kernel, dim3(b), dim3(ORD_THREADS), order_lds_bytes(nf, per), s, a);
    hipLaunchKernelGGL(surface_gather_kernel, dim3(((int64_t)nv * VTX_LANES + SGA_THREADS - 1) / SGA_THREADS, b),
                       dim3(SGA_THREADS), 0, s, a);
    return geom::launch_status();
}

// ---- forward-side finalize + single-launch backward --------------------------------------------------------------
// scratch layout (int32 words): off[b,nf+1] | seg[b,cap] | pface[b,cap] | slot[b,cap] | pad to 4 | rec[b,cap,2] float4 |
// with cap = num + n_gt
extern "C" int64_t geom_surface_order_words(int b, int nf, int num, int n_gt)
{
    if (b <= 0 || nf < 0 || num < 0 || n_gt < 0) return 0;
    const int64_t cap = (int64_t)num + n_gt;
    return geom_surface_status_offset(b, nf, cap) + geom_surface_status_ints(b);
}

extern "C" int geom_surface_finalize_w_f32(int b, int nf, int num, const int64_t *choices, const float *u, const float *v,
                                           const float *points, int n_gt, const float *gt, const int *idx_g,
                                           const int *idx_p, const int *index, const float *closest, const float *weights,
                                           const float *sq_sample, const float *sq_other, float scale_sample,
                                           float scale_other, float coef_sample, float coef_other, int want_order,
                                           int records_ready, int *order, float *loss, const float *mesh_weight, void *stream);

extern "C" int geom_surface_finalize_f32(int b, int nf, int num, const int64_t *choices, const float *u, const float *v,
                                         const float *points, int n_gt, const float *gt, const int *idx_g,
                                         const int *idx_p, const int *index, const float *closest, const float *weights,
                                         const float *sq_sample, const float *sq_other, float scale_sample,
                                         float scale_other, float coef_sample, float coef_other, int want_order,
                                         int records_ready, int *order, float *loss, void *stream)
{
    return geom_surface_finalize_w_f32(b, nf, num, choices, u, v, points, n_gt, gt, idx_g, idx_p, index, closest, weights, sq_sample,
                                       sq_other, scale_sample, scale_other, coef_sample, coef_other, want_order, records_ready, order,
                                       loss, nullptr, stream);
}

// mesh_weight (may be NULL = all ones): [b] device floats; the loss becomes sum_m w[m] * (mesh m's two sums) -- several
// equal-size batches (the stages of a cascade) stacked into ONE call, each with its own factor.  The gradient records do not
// carry the weights: geom_surface_gather_w_f32 applies the same array.
extern "C" int geom_surface_finalize_w_f32(int b, int nf, int num, const int64_t *choices, const float *u, const float *v,
                                           const float *points, int n_gt, const float *gt, const int *idx_g,
                                           const int *idx_p, const int *index, const float *closest, const float *weights,
                                           const float *sq_sample, const float *sq_other, float scale_sample,
                                           float scale_other, float coef_sample, float coef_other, int want_order,
                                           int records_ready, int *order, float *loss, const float *mesh_weight, void *stream)
{
    if (b < 0 || nf < 0 || num < 0 || n_gt < 0) return GEOM_EINVAL;
    if (!loss || !order || ((uintptr_t)order & 15)) return GEOM_EINVAL;
    if (b == 0) return 0;
    if ((num > 0 && !sq_sample) || (n_gt > 0 && !sq_other)) return GEOM_EINVAL;
    if (idx_p && index) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    const int other = idx_p ? OTHER_NN : (index ? OTHER_TRI : OTHER_NONE);
    const int64_t per64 = (int64_t)num + (other != OTHER_NONE ? n_gt : 0);
    if (per64 > 0x3fffffff) return GEOM_ETOOBIG;
    const int per = (int)per64;
    if (want_order) {
        if (num > 0 && (!choices || !u || !v || !points || !gt || !idx_g || n_gt == 0)) return GEOM_EINVAL;
        if (index && (!closest || !weights || !gt)) return GEOM_EINVAL;
        if (order_lds_bytes(nf, per) > ORD_LDS_LIMIT) return GEOM_EUNSUPPORTED;
    }
    const int64_t cap = (int64_t)num + n_gt;
    int *off = order;
    int *seg = off + (int64_t)b * (nf + 1);
    int *pface = seg + (int64_t)b * cap;
    int *slot = pface + (int64_t)b * cap;
    float4 *rec = reinterpret_cast<float4 *>(order + geom_surface_order_ints(b, nf, cap));
    FinalizeArgs a{choices, u, v, points, gt, idx_g, idx_p, index, closest, weights, sq_sample, sq_other, scale_sample,
                   scale_other, coef_sample, coef_other, b, nf, num, n_gt, other, per, want_order ? 1 : 0, records_ready ? 1 : 0, off, seg, pface,
                   slot, rec, loss, order + geom_surface_status_offset(b, nf, cap), mesh_weight};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // without ordering the single (loss) workgroup touches only its reduction scratch: independent of nf, so the
    // documented fallback beyond the ordering limit (want_order = 0 + scatter backward) really launches
    const size_t lds = geom_finalize::finalize_lds_ints(nf, per, ORD_THREADS, want_order != 0) * sizeof(int);
    if (per <= geom_finalize::FIN_REG_POINTS) {
        static const hipError_t opt_in = hipFuncSetAttribute(reinterpret_cast<const void *>(surface_finalize_kernel<true>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ORD_LDS_LIMIT + 1024);
        (void)opt_in;
        hipLaunchKernelGGL(surface_finalize_kernel<true>, dim3(want_order ? b + 1 : 1), dim3(ORD_THREADS), lds, s, a);
    } else {
        static const hipError_t opt_in = hipFuncSetAttribute(reinterpret_cast<const void *>(surface_finalize_kernel<false>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ORD_LDS_LIMIT + 1024);
        (void)opt_in;
        hipLaunchKernelGGL(surface_finalize_kernel<false>, dim3(want_order ? b + 1 : 1), dim3(ORD_THREADS), lds, s, a);
    }
    return geom::launch_status();
}

extern "C" int geom_surface_gather_w_f32(int b, int nv, int nf, const int *vf_ptr, const int *vf_item, int num, int n_gt,
                                         int has_other, const int *order, const float *grad, const float *mesh_weight,
                                         float *grad_verts, void *stream);

extern "C" int geom_surface_gather_f32(int b, int nv, int nf, const int *vf_ptr, const int *vf_item, int num, int n_gt,
                                       int has_other, const int *order, const float *grad, float *grad_verts, void *stream)
{
    return geom_surface_gather_w_f32(b, nv, nf, vf_ptr, vf_item, num, n_gt, has_other, order, grad, nullptr, grad_verts, stream);
}

// mesh_weight (may be NULL): the per-mesh factors of geom_surface_finalize_w_f32; mesh m's gradient = w[m] * 2 * grad[0] * (...)
extern "C" int geom_surface_gather_w_f32(int b, int nv, int nf, const int *vf_ptr, const int *vf_item, int num, int n_gt,
                                         int has_other, const int *order, const float *grad, const float *mesh_weight,
                                         float *grad_verts, void *stream)
{
    if (b < 0 || nv < 0 || nf < 0 || num < 0 || n_gt < 0) return GEOM_EINVAL;
    if (b == 0 || nv == 0) return 0;
    if (!vf_ptr || !vf_item || !order || !grad_verts || ((uintptr_t)order & 15)) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    const int64_t cap = (int64_t)num + n_gt;
    const int per = num + (has_other ? n_gt : 0);
    const int *off = order;
    const int *seg = off + (int64_t)b * (nf + 1);
    const float4 *rec = reinterpret_cast<const float4 *>(order + geom_surface_order_ints(b, nf, cap));
    VGatherArgs a{vf_ptr, vf_item, off, seg, rec, grad, grad_verts, nv, nf, per, order + geom_surface_status_offset(b, nf, cap), b,
                  mesh_weight};
    hipLaunchKernelGGL(surface_vertex_gather_kernel, dim3(((int64_t)nv * VTX_LANES + SGA_THREADS - 1) / SGA_THREADS, b),
                       dim3(SGA_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return geom::launch_status();
}
