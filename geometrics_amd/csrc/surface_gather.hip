// Backward of the sampled-surface losses (batch_point_to_point / batch_point_to_surface, reference utils.py:393-502
// under autograd) as a GATHER: every vertex sums the contributions of the sampled points and gt points that landed on
// its incident faces, in a fixed order.
//
// The scatter formulation (sample_loss.hip: one thread per point, nine fp32 atomics into a zeroed grad_verts) costs
// 26 us for 432 000 atomics at the BASELINE shard, needs a zero-fill, and adds in arrival order, so the last bits of
// the gradient change from run to run.  Here:
//   bin     one thread per point: its gradient vector (point - partner) * coefficient and its three corner weights are
//           computed here, fully parallel and coalesced, and stored as two float4 records; slot =
//           atomicAdd(count[mesh][face], 1) (integer, 48 000 of them, almost no contention) and the point's id goes
//           into a BIN_CAP-slot list of that face;
//   gather  eight lanes per (mesh, vertex), one incident (face, corner) each -- a static CSR built once per face
//           list: read the face's list, take its ids in ascending order and accumulate their gradients with this
//           corner's barycentric weight; the lanes' partial sums are folded in lane order.  Points beyond a face's slots go to a per-mesh overflow list; a face that has some is
//           summed by repeatedly extracting the next-larger id from (its list + the overflow list), or -- when that
//           would cost more -- by an ordered scan of all the mesh's points.  Either way: exact, and in id order.
// grad_verts is written once per element (no zero-fill, no float atomics) and is bit-reproducible.
// The per-face counters must be zero on entry (the forward's reduction launch clears them, see geom_sum2_f32).
#include "geom_common.h"
#include "tri_math.h"

namespace {

using geom::V3;

constexpr int SGA_THREADS = 256;
constexpr int BIN_CAP = 16;
constexpr int VTX_LANES = 8;  // lanes that share a vertex in the gather: one incident face each, then an ordered fold
constexpr int OVERFLOW_CAP = 2048; // per mesh

enum { OTHER_NONE = 0, OTHER_NN = 1, OTHER_TRI = 2 };

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
    int b, nv, nf, num, n_gt, other;
    int *counts; // [b,nf] points per face, then [b] overflow entries per mesh
    int *lists;  // [b,nf,BIN_CAP] point ids, then [b,OVERFLOW_CAP,2] (face, id) pairs, then the point records
    float4 *rec; // [b, num + n_gt, 2]: {gradient vector, skip-zero-weights flag}, {w0, w1, w2, -}
    float *grad_verts;
};

__device__ __forceinline__ V3 ld3(const float *p) { return geom::mk(p[0], p[1], p[2]); }

// face a gt point contributes to
__device__ __forceinline__ int64_t other_face(const GatherArgs &a, int mesh, int g)
{
    const int64_t o = (int64_t)mesh * a.n_gt + g;
    return a.other == OTHER_NN ? a.choices[(int64_t)mesh * a.num + a.idx_p[o]] : (int64_t)a.index[o];
}

__global__ __launch_bounds__(SGA_THREADS) void surface_bin_kernel(GatherArgs a)
{
    const int per = a.num + (a.other != OTHER_NONE ? a.n_gt : 0);
    const int64_t i = (int64_t)blockIdx.x * SGA_THREADS + threadIdx.x;
    if (i >= (int64_t)a.b * per) return;
    const int mesh = (int)(i / per), id = (int)(i - (int64_t)mesh * per);
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
    if (f < 0 || f >= a.nf) return; // not a face of this mesh: contributes nowhere (the scatter would have faulted)
    const int64_t bin = (int64_t)mesh * a.nf + f;
    const int slot = atomicAdd(&a.counts[bin], 1);
    if (slot < BIN_CAP) {
        a.lists[bin * BIN_CAP + slot] = id;
    } else {
        const int k = atomicAdd(&a.counts[(int64_t)a.b * a.nf + mesh], 1);
        if (k < OVERFLOW_CAP) {
            int *ov = a.lists + (int64_t)a.b * a.nf * BIN_CAP + ((int64_t)mesh * OVERFLOW_CAP + k) * 2;
            ov[0] = (int)f;
            ov[1] = id;
        }
    }
}

// gradient contribution of point `id` of `mesh` to corner c of its face, from the records of the bin pass
__device__ __forceinline__ V3 apply_record(float4 g, float4 w, int c)
{
    const float wc = c == 0 ? w.x : (c == 1 ? w.y : w.z);
    if (g.w != 0.f && wc == 0.f) return geom::mk(0.f, 0.f, 0.f);
    return geom::mk(g.x, g.y, g.z) * wc;
}
__device__ __forceinline__ V3 contribution(const GatherArgs &a, int mesh, int id, int c)
{
    const int64_t i = (int64_t)mesh * (a.num + (a.other != OTHER_NONE ? a.n_gt : 0)) + id;
    return apply_record(a.rec[2 * i], a.rec[2 * i + 1], c);
}

__device__ __forceinline__ void order2(int &x, int &y)
{
    const int lo = min(x, y), hi = max(x, y);
    x = lo;
    y = hi;
}

// sum of the contributions of the points on face f to its corner c, in ascending point-id order
__device__ __forceinline__ V3 face_sum(const GatherArgs &a, int mesh, int f, int c)
{
    V3 acc = geom::mk(0.f, 0.f, 0.f);
    const int64_t bin = (int64_t)mesh * a.nf + f;
    const int n = a.counts[bin];
    if (n == 0) return acc;
    const int *list = a.lists + bin * BIN_CAP;
    const int64_t per = a.num + (a.other != OTHER_NONE ? a.n_gt : 0);
    const float4 *rec = a.rec + 2 * (int64_t)mesh * per;
    if (n <= 4) { // the usual case (1.2 points per face on average): ids, 4-element network, records in one round trip
        int id[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) id[k] = k < n ? list[k] : INT_MAX;
        order2(id[0], id[1]), order2(id[2], id[3]), order2(id[0], id[2]), order2(id[1], id[3]), order2(id[1], id[2]);
        float4 g[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) g[k] = rec[2 * id[k]], w[k] = rec[2 * id[k] + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) acc = acc + apply_record(g[k], w[k], c);
        return acc;
    }
    if (n <= BIN_CAP) { // up to 16 points: Batcher's odd-even merge network on registers, records eight at a time
        int id[BIN_CAP];
#pragma unroll
        for (int k = 0; k < BIN_CAP; ++k) id[k] = k < n ? list[k] : INT_MAX;
#pragma unroll
        for (int p = 1; p < BIN_CAP; p <<= 1)
#pragma unroll
            for (int k = p; k >= 1; k >>= 1)
#pragma unroll
                for (int j = k % p; j <= BIN_CAP - 1 - k; j += 2 * k)
#pragma unroll
                    for (int i = 0; i < (k < BIN_CAP - j - k ? k : BIN_CAP - j - k); ++i)
                        if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) order2(id[i + j], id[i + j + k]);
#pragma unroll
        for (int h = 0; h < BIN_CAP; h += 8) {
            if (h >= n) break;
            float4 g[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (h + k < n) g[k] = rec[2 * id[h + k]], w[k] = rec[2 * id[h + k] + 1];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (h + k < n) acc = acc + apply_record(g[k], w[k], c);
        }
        return acc;
    }
    const int n_ov = a.counts[(int64_t)a.b * a.nf + mesh];
    const int *ov = a.lists + (int64_t)a.b * a.nf * BIN_CAP + (int64_t)mesh * OVERFLOW_CAP * 2;
    if (n_ov <= OVERFLOW_CAP && (int64_t)n * (BIN_CAP + n_ov) <= per) {
        // repeatedly take the next-larger id among the face's slots (+ the mesh's overflow entries when the face
        // has more points than slots): the arrival order in the lists is not reproducible, the id order is
        const int listed = BIN_CAP;
        int last = -1;
        for (int t = 0; t < n; ++t) {
            int best = INT_MAX;
            for (int k = 0; k < listed; ++k) {
                const int id = list[k];
                if (id > last && id < best) best = id;
            }
            for (int k = 0; k < n_ov; ++k)
                if (ov[2 * k] == f) {
                    const int id = ov[2 * k + 1];
                    if (id > last && id < best) best = id;
                }
            acc = acc + contribution(a, mesh, best, c);
            last = best;
        }
    } else { // crowded mesh: one ordered pass over all of its points is cheaper (and needs no lists)
        for (int s = 0; s < a.num; ++s)
            if (a.choices[(int64_t)mesh * a.num + s] == f) acc = acc + contribution(a, mesh, s, c);
        if (a.other != OTHER_NONE)
            for (int g = 0; g < a.n_gt; ++g)
                if (other_face(a, mesh, g) == f) acc = acc + contribution(a, mesh, a.num + g, c);
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

} // namespace

// ints needed behind `counts` / `lists` for a batch of b meshes of nf faces
extern "C" int64_t geom_surface_bin_count_words(int b, int nf) { return b <= 0 || nf < 0 ? 0 : (int64_t)b * nf + b; }
// (rounded up to a multiple of 4 so that the float4 point records behind the lists stay 16-byte aligned)
static inline int64_t list_words(int b, int nf) { return (((int64_t)b * nf * BIN_CAP + (int64_t)b * OVERFLOW_CAP * 2) + 3) & ~3ll; }
extern "C" int64_t geom_surface_bin_list_words(int b, int nf, int num, int n_gt)
{
    return b <= 0 || nf < 0 || num < 0 || n_gt < 0 ? 0 : list_words(b, nf) + (int64_t)b * (num + n_gt) * 8;
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
    GatherArgs a{vf_ptr, vf_item, choices, u, v, points, gt, idx_g, idx_p, index, closest, weights, coef_dev, coef_sample,
                 coef_other, b, nv, nf, num, n_gt, idx_p ? OTHER_NN : (index ? OTHER_TRI : OTHER_NONE), counts, lists,
                 reinterpret_cast<float4 *>(lists + list_words(b, nf)), grad_verts};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t items = (int64_t)b * (num + (a.other != OTHER_NONE ? n_gt : 0));
    if ((items + SGA_THREADS - 1) / SGA_THREADS > 0x7fffffffLL) return GEOM_ETOOBIG;
    if (items > 0)
        hipLaunchKernelGGL(surface_bin_kernel, dim3((unsigned)((items + SGA_THREADS - 1) / SGA_THREADS)), dim3(SGA_THREADS), 0, s, a);
    hipLaunchKernelGGL(surface_gather_kernel, dim3(((int64_t)nv * VTX_LANES + SGA_THREADS - 1) / SGA_THREADS, b),
                       dim3(SGA_THREADS), 0, s, a);
    return geom::launch_status();
}
