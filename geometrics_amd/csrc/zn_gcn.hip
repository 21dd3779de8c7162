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
struct Pack<2> {
    float2 v;
    __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const float2 *>(p); }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float2 *>(p) = v; }
    __device__ __forceinline__ float &at(int i) { return (&v.x)[i]; }
};
template <>
struct Pack<1> {
    float v;
    __device__ __forceinline__ void load(const float *p) { v = *p; }
    __device__ __forceinline__ void store(float *p) const { *p = v; }
    __device__ __forceinline__ float &at(int) { return v; }
};

// Thread layout: a block owns `rows_per_block` consecutive rows of ONE mesh; thread t owns column group
// (t % groups) for rows (t / groups), (t / groups) + rows_in_flight, ...  so that
//   * a wave reads/writes contiguous runs of a row (coalesced),
//   * a thread's column group is FIXED, which lets the backward kernel accumulate the bias
//     gradient (column sums of g) in registers and emit one partial per block -- reduced by a
//     second tiny kernel in a fixed order (deterministic, no atomics).
// Neighbour gathers are issued in batches of NB: first all (col, val) pairs of the batch, then
// all neighbour rows, so a row costs two memory round trips instead of two per neighbour.
constexpr int GCN_NB = 8;         // neighbours fetched per batch
constexpr int GCN_BWD_ITERS = 2;  // rows per thread in the backward (halves the bias-gradient partials)

template <int VEC, int ACT, bool BACKWARD>
__global__ __launch_bounds__(GCN_THREADS) void zn_aggregate_kernel(GcnArgs a, int groups, int rows_in_flight,
                                                                    int rows_per_block, float *colsum_partial)
{
    extern __shared__ float lds_colsum[]; // [rows_in_flight][c], backward with colsum only
    const int g = threadIdx.x % groups;
    const int rl = threadIdx.x / groups;
    const int c0 = g * VEC;
    const bool active = rl < rows_in_flight;
    // grid = (row chunks of one mesh, meshes): no integer division on the row index
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(r_begin + rows_per_block, a.nv);
    const int64_t mesh_row0 = (int64_t)blockIdx.y * a.nv; // first row of this mesh

    Pack<VEC> colsum;
#pragma unroll
    for (int i = 0; i < VEC; ++i) colsum.at(i) = 0.f;

    if (active) {
        for (int r = r_begin + rl; r < r_end; r += rows_in_flight) {
            const int64_t row = mesh_row0 + r;
            Pack<VEC> acc, own;
            const bool pass = c0 + VEC > a.k; // the group holds pass-through columns (all of them, or -- when k is not
                                              // a multiple of VEC -- the tail of the group that straddles column k)
            if (BACKWARD || pass) { // the thread's own element: pass-through value and/or bias-gradient term
                own.load(a.x + row * a.c + c0);
                if (BACKWARD && ACT != ACT_NONE) {
                    Pack<VEC> o;
                    o.load(a.saved + row * a.c + c0);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) own.at(i) = act_bwd<ACT>(own.at(i), o.at(i));
                }
            }
            if (c0 < a.k) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc.at(i) = 0.f;
                const int e1 = a.rowptr[r + 1];
                for (int e = a.rowptr[r]; e < e1; e += GCN_NB) {
                    int64_t nb[GCN_NB];
                    float w[GCN_NB];
#pragma unroll
                    for (int j = 0; j < GCN_NB; ++j) {
                        const bool in = e + j < e1;
                        nb[j] = in ? mesh_row0 + a.col[e + j] : row;
                        w[j] = in ? a.val[e + j] : 0.f;
                    }
                    Pack<VEC> sv[GCN_NB], ov[GCN_NB];
#pragma unroll
                    for (int j = 0; j < GCN_NB; ++j) {
                        sv[j].load(a.x + nb[j] * a.c + c0);
                        if (BACKWARD && ACT != ACT_NONE) ov[j].load(a.saved + nb[j] * a.c + c0);
                    }
#pragma unroll
                    for (int j = 0; j < GCN_NB; ++j) {
                        if (e + j < e1) { // keep the CSR order of the sum; padded slots contribute nothing
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                float v = sv[j].at(i);
                                if (BACKWARD && ACT != ACT_NONE) v = act_bwd<ACT>(v, ov[j].at(i));
                                acc.at(i) += w[j] * v;
                            }
                        }
                    }
                }
                if (pass) { // straddling group: the gathered values of columns >= k are discarded
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        if (c0 + i >= a.k) acc.at(i) = own.at(i);
                }
            } else {
                acc = own;
            }
            if (BACKWARD) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) colsum.at(i) += own.at(i);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float v = acc.at(i);
                    if (a.bias) v += a.bias[c0 + i];
                    acc.at(i) = act_fwd<ACT>(v);
                }
            }
            acc.store(a.y + row * a.c + c0);
        }
    }

    if (BACKWARD && colsum_partial) { // block partial of the bias gradient, fixed summation order
        if (active) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) lds_colsum[rl * a.c + c0 + i] = colsum.at(i);
        }
        __syncthreads();
        if (active && rl == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float t = 0.f;
                for (int l = 0; l < rows_in_flight; ++l) t += lds_colsum[l * a.c + c0 + i];
                colsum_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * a.c + c0 + i] = t;
            }
        }
    }
}

// grad_bias[c] = sum over blocks of partial[blk][c].  16 columns x 64 block-lanes per workgroup:
// lane l adds blocks l, l+64, ... (independent loads, pipelined), then the 64 lane sums are added
// in ascending lane order -- a fixed tree, so the result is bit-reproducible.
constexpr int CS_COLS = 16, CS_LANES = 64;
__global__ __launch_bounds__(CS_COLS *CS_LANES) void colsum_final_kernel(int blocks, int c, const float *partial,
                                                                         float *out)
{
    __shared__ float part[CS_LANES][CS_COLS];
    const int cl = threadIdx.x % CS_COLS, lane = threadIdx.x / CS_COLS;
    const int col = blockIdx.x * CS_COLS + cl;
    float t = 0.f;
    if (col < c) {
#pragma unroll 8
        for (int b = lane; b < blocks; b += CS_LANES) t += partial[(size_t)b * c + col];
    }
    part[lane][cl] = t;
    __syncthreads();
    if (lane == 0 && col < c) {
        float acc = 0.f;
        for (int l = 0; l < CS_LANES; ++l) acc += part[l][cl];
        out[col] = acc;
    }
}

// The same reduction for several pending bias gradients in ONE launch (blockIdx.y = job): a backward pass through N
// layers leaves N sets of partials, and N separate 4.7 us launches are 5 % of the reference's training-shape step.
struct ColsumJobs {
    const float *partial[GEOM_COLSUM_MAX_JOBS];
    float *out[GEOM_COLSUM_MAX_JOBS];
    int blocks[GEOM_COLSUM_MAX_JOBS], c[GEOM_COLSUM_MAX_JOBS];
};

__global__ __launch_bounds__(CS_COLS *CS_LANES) void colsum_batch_kernel(ColsumJobs jobs)
{
    __shared__ float part[CS_LANES][CS_COLS];
    const int job = blockIdx.y;
    const int blocks = jobs.blocks[job], c = jobs.c[job];
    if ((int)blockIdx.x * CS_COLS >= c) return; // a job narrower than the widest one
    const float *partial = jobs.partial[job];
    const int cl = threadIdx.x % CS_COLS, lane = threadIdx.x / CS_COLS;
    const int col = blockIdx.x * CS_COLS + cl;
    float t = 0.f;
    if (col < c) {
#pragma unroll 8
        for (int b = lane; b < blocks; b += CS_LANES) t += partial[(size_t)b * c + col];
    }
    part[lane][cl] = t;
    __syncthreads();
    if (lane == 0 && col < c) {
        float acc = 0.f;
        for (int l = 0; l < CS_LANES; ++l) acc += part[l][cl];
        jobs.out[job][col] = acc;
    }
}

struct GcnGeometry {
    int groups, rows_in_flight, rows_per_block;
    int chunks;     // blocks per mesh (grid.x)
    int64_t blocks; // chunks * meshes
};

inline GcnGeometry gcn_geometry(int b, int nv, int c, int vec, bool backward)
{
    GcnGeometry g;
    g.groups = c / vec;
    g.rows_in_flight = g.groups >= GCN_THREADS ? 1 : GCN_THREADS / g.groups;
    g.rows_per_block = g.rows_in_flight * (backward ? GCN_BWD_ITERS : 1);
    g.chunks = (nv + g.rows_per_block - 1) / g.rows_per_block;
    g.blocks = (int64_t)g.chunks * b;
    return g;
}

inline bool gcn_vec4(int c, int /*k*/) { return c % 4 == 0; } // any k: the group straddling column k is mixed

template <int VEC, bool BACKWARD>
int launch(const GcnArgs &a, int act, float *colsum_partial, float *grad_bias, void *stream)
{
    const int meshes = (int)(a.rows / a.nv);
    const GcnGeometry geo = gcn_geometry(meshes, a.nv, a.c, VEC, BACKWARD);
    if (geo.groups > GCN_THREADS) return GEOM_ETOOBIG; // > 1024 channels (VEC=4): not a 0N-GCN shape
    if (meshes > 65535 || geo.blocks > 0x7fffffffLL) return GEOM_ETOOBIG;
    dim3 grid((unsigned)geo.chunks, (unsigned)meshes), block(GCN_THREADS);
    const size_t lds = (BACKWARD && colsum_partial) ? (size_t)geo.rows_in_flight * a.c * sizeof(float) : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (act) {
    case ACT_NONE: hipLaunchKernelGGL((zn_aggregate_kernel<VEC, ACT_NONE, BACKWARD>), grid, block, lds, s, a, geo.groups, geo.rows_in_flight, geo.rows_per_block, colsum_partial); break;
    case ACT_RELU: hipLaunchKernelGGL((zn_aggregate_kernel<VEC, ACT_RELU, BACKWARD>), grid, block, lds, s, a, geo.groups, geo.rows_in_flight, geo.rows_per_block, colsum_partial); break;
    case ACT_ELU: hipLaunchKernelGGL((zn_aggregate_kernel<VEC, ACT_ELU, BACKWARD>), grid, block, lds, s, a, geo.groups, geo.rows_in_flight, geo.rows_per_block, colsum_partial); break;
    default: return GEOM_EINVAL;
    }
    if (BACKWARD && colsum_partial && grad_bias) // grad_bias == null: partials only (geom_colsum_batch_f32 finishes them)
        hipLaunchKernelGGL(colsum_final_kernel, dim3((a.c + CS_COLS - 1) / CS_COLS), dim3(CS_COLS * CS_LANES), 0, s,
                           (int)geo.blocks, a.c, colsum_partial, grad_bias);
    return geom::launch_status();
}

// ---------------------------------------------------------------------------------------
// ELL fast path (k % 4 == 0, C == k * (NC + 1) with NC = 2 (split 3) or 9 (split 10); the first W = 8 / 16
// entries of every row in a fixed-stride table, the few longer rows -- the two 33-entry poles of the reference's
// training template 482.obj -- continue in a short CSR tail: every hidden layer of the reference models on a
// triangle mesh).
//
// Thread (row, j) owns the aggregated float4 j of the row AND the NC pass-through float4s
// j + (k/4)*i of the same row.  Compared with the generic kernel above:
//   * every lane gathers (no wave where 2/3 of the lanes wait for the gathering third),
//     b*V*k/4 threads fit the chip in ONE resident round;
//   * neighbour indices/weights sit at a fixed stride (ELL, -1 padded): no rowptr round trip, the
//     index loads and the thread's own pass-through loads are issued together, then all W
//     neighbour rows -- two memory round trips per thread;
//   * the thread stores NC + 1 float4s; in the backward it also owns the bias-gradient terms of
//     those columns, reduced per workgroup through LDS (fixed order) into one partial per block.
struct EllArgs {
    const int *col;     // [nv][W], -1 = padding
    const float *val;   // [nv][W]
    const float *x, *bias, *saved;
    float *y;
    int nv, c, k;
    unsigned short *mask; // ReLU sign bits, one word per thread: bit 4*i + e <-> element e of the thread's float4 i
    // rows longer than W (the 33-entry poles of the reference's 482.obj): entries W.. of every row in CSR form,
    // summed after the table slots, i.e. still in CSR order.  over_ptr == null: no row is longer than W.
    const int *over_ptr, *over_col;
    const float *over_val;
    int b, chunks; // meshes, workgroups per mesh (filled in by the dispatcher)
    // HEAD (the layer whose three leading output channels are a coordinate update, GEOMetrics.py:121,126,131):
    //   forward : head_out[row] = head_in[row] + head_scale * out[row, :3]   (base positions in, new positions out)
    //   backward: grad_out is not read at all -- it is [head_scale * head_in[row, :3] | 0 ...] by construction
    //             (head_in = grad_pos [rows, 3]); x may be null
    const float *head_in;
    float *head_out;
    float head_scale;
};

// MASK (ReLU, NC == 2 only): the forward also stores the sign of every output element as one bit (12 bits per
// thread -> 0.66 MB per layer at the BASELINE shard), and the backward takes relu' from those bits instead of
// re-reading the 15.7 MB forward output: 47 -> 32 MB of traffic for the backward launch.
template <int ACT, bool BACKWARD, int W, int NC, bool MASK = false, bool HEAD = false>
__global__ __launch_bounds__(GCN_THREADS) void zn_aggregate_ell_kernel(EllArgs a, int rows_per_block, int iters,
                                                                        float *colsum_partial)
{
    static_assert(!MASK || (ACT == ACT_RELU && NC == 2), "the sign mask is the ReLU / split-3 fast path");
    extern __shared__ float lds_cs[]; // [rows_per_block][c]
    const int kg = a.k >> 2;                         // aggregated float4 groups per row
    const int j = threadIdx.x % kg;
    const int rl = threadIdx.x / kg;
    // a mesh's workgroups share one XCD (geom::xcd_assign): the neighbour rows every thread gathers -- each row of the
    // k-slice is read by ~7 workgroups -- then hit that XCD's L2 instead of being fetched into all eight
    int mesh_i, chunk_i;
    if (!geom::xcd_assign(blockIdx.x, a.b, a.chunks, mesh_i, chunk_i)) return;
    const int64_t mesh_row0 = (int64_t)mesh_i * a.nv;
    const int c0 = 4 * j;
    // `iters` consecutive row tiles per workgroup (the backward uses 2: half the bias-gradient partials to reduce)
    float4 csum[NC + 1]; // bias-gradient terms of this thread's columns over its rows (backward)
#pragma unroll
    for (int i = 0; i <= NC; ++i) csum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int it = 0; it < iters; ++it) {
    const int r = (chunk_i * iters + it) * rows_per_block + rl;
    const bool active = rl < rows_per_block && r < a.nv;
    float4 own[NC + 1]; // [0] = own aggregated-slot element (backward only), [1..NC] = pass-through
#pragma unroll
    for (int i = 0; i <= NC; ++i) own[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    if (active) {
        const int64_t row = mesh_row0 + r;
        const float *xrow = a.x + row * a.c;
        // round trip 1: indices, weights and the thread's own elements
        int nb[W];
        float w[W];
#pragma unroll
        for (int n = 0; n < W; n += 4) {
            const int4 ci = *reinterpret_cast<const int4 *>(a.col + (size_t)r * W + n);
            const float4 wi = *reinterpret_cast<const float4 *>(a.val + (size_t)r * W + n);
            nb[n] = ci.x, nb[n + 1] = ci.y, nb[n + 2] = ci.z, nb[n + 3] = ci.w;
            w[n] = wi.x, w[n + 1] = wi.y, w[n + 2] = wi.z, w[n + 3] = wi.w;
        }
        constexpr int TAIL = (ACT == ACT_NONE && NC == 2) ? 16 : 8; // entries of a long row per round (below)
        int e0 = 0, e1 = 0; // the row's entries beyond the table width, in the CSR tail
        if (a.over_ptr) e0 = a.over_ptr[r], e1 = a.over_ptr[r + 1];
        unsigned own_bits = 0u;
        if (BACKWARD && MASK) own_bits = a.mask[row * kg + j];
#pragma unroll
        for (int i = BACKWARD ? 0 : 1; i <= NC; ++i) {
            if (BACKWARD && HEAD) { // the upstream gradient is [scale * grad_pos | 0]: only the row's first float4 is non-zero
                if (i == 0 && j == 0) {
                    const float *gp = a.head_in + row * 3;
                    own[0] = make_float4(a.head_scale * gp[0], a.head_scale * gp[1], a.head_scale * gp[2], 0.f);
                }
            } else
            own[i] = *reinterpret_cast<const float4 *>(xrow + c0 + a.k * i);
            if (BACKWARD && MASK) {
                const unsigned m = own_bits >> (4 * i);
                own[i].x = (m & 1u) ? own[i].x : 0.f;
                own[i].y = (m & 2u) ? own[i].y : 0.f;
                own[i].z = (m & 4u) ? own[i].z : 0.f;
                own[i].w = (m & 8u) ? own[i].w : 0.f;
            } else if (BACKWARD && ACT != ACT_NONE) {
                const float4 o = *reinterpret_cast<const float4 *>(a.saved + row * a.c + c0 + a.k * i);
                own[i].x = act_bwd<ACT>(own[i].x, o.x);
                own[i].y = act_bwd<ACT>(own[i].y, o.y);
                own[i].z = act_bwd<ACT>(own[i].z, o.z);
                own[i].w = act_bwd<ACT>(own[i].w, o.w);
            }
        }
        // round trip 2: the neighbour rows of the aggregated slot
        float4 sv[W], ov[W];
        unsigned nbits[W];
#pragma unroll
        for (int n = 0; n < W; ++n) {
            const int64_t nrow = mesh_row0 + (nb[n] >= 0 ? nb[n] : r);
            if (BACKWARD && HEAD) {
                sv[n] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j == 0) {
                    const float *gp = a.head_in + nrow * 3;
                    sv[n] = make_float4(a.head_scale * gp[0], a.head_scale * gp[1], a.head_scale * gp[2], 0.f);
                }
            } else
            sv[n] = *reinterpret_cast<const float4 *>(a.x + nrow * a.c + c0);
            if (BACKWARD && MASK) nbits[n] = a.mask[nrow * kg + j];
            else if (BACKWARD && ACT != ACT_NONE) ov[n] = *reinterpret_cast<const float4 *>(a.saved + nrow * a.c + c0);
        }
        int tcol[TAIL];
        float tval[TAIL];
        if (e0 < e1) { // a long row: the indices of its first TAIL extra entries travel with the neighbour rows
#pragma unroll
            for (int t = 0; t < TAIL; ++t) {
                const int ee = e0 + t < e1 ? e0 + t : e1 - 1;
                tcol[t] = a.over_col[ee];
                tval[t] = a.over_val[ee];
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < W; ++n) {
            if (nb[n] >= 0) { // ELL order == CSR order of the row
                float4 v = sv[n];
                if (BACKWARD && MASK) {
                    v.x = (nbits[n] & 1u) ? v.x : 0.f;
                    v.y = (nbits[n] & 2u) ? v.y : 0.f;
                    v.z = (nbits[n] & 4u) ? v.z : 0.f;
                    v.w = (nbits[n] & 8u) ? v.w : 0.f;
                } else if (BACKWARD && ACT != ACT_NONE) {
                    v.x = act_bwd<ACT>(v.x, ov[n].x);
                    v.y = act_bwd<ACT>(v.y, ov[n].y);
                    v.z = act_bwd<ACT>(v.z, ov[n].z);
                    v.w = act_bwd<ACT>(v.w, ov[n].w);
                }
                acc.x += w[n] * v.x;
                acc.y += w[n] * v.y;
                acc.z += w[n] * v.z;
                acc.w += w[n] * v.w;
            }
        }
        // the row's entries beyond the table width (wave-divergent only at the few long rows).  TAIL entries per round:
        // their rows in one round trip, the NEXT round's indices requested meanwhile (one entry at a time the 25 extra
        // entries of a 482.obj pole cost 50 dependent round trips, and the two pole rows set the launch time)
        for (int e = e0; e < e1; e += TAIL) {
            int64_t nrow[TAIL];
            float wv[TAIL];
            float4 tv[TAIL], to[TAIL];
            unsigned tb[TAIL];
#pragma unroll
            for (int t = 0; t < TAIL; ++t) {
                nrow[t] = mesh_row0 + tcol[t];
                wv[t] = tval[t];
                if (BACKWARD && HEAD) {
                    tv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j == 0) {
                        const float *gp = a.head_in + nrow[t] * 3;
                        tv[t] = make_float4(a.head_scale * gp[0], a.head_scale * gp[1], a.head_scale * gp[2], 0.f);
                    }
                } else
                tv[t] = *reinterpret_cast<const float4 *>(a.x + nrow[t] * a.c + c0);
                if (BACKWARD && MASK) tb[t] = a.mask[nrow[t] * kg + j];
                else if (BACKWARD && ACT != ACT_NONE) to[t] = *reinterpret_cast<const float4 *>(a.saved + nrow[t] * a.c + c0);
            }
            if (e + TAIL < e1) {
#pragma unroll
                for (int t = 0; t < TAIL; ++t) {
                    const int ee = e + TAIL + t < e1 ? e + TAIL + t : e1 - 1;
                    tcol[t] = a.over_col[ee];
                    tval[t] = a.over_val[ee];
                }
            }
#pragma unroll
            for (int t = 0; t < TAIL; ++t) {
                if (e + t < e1) { // still in CSR order
                    float4 v = tv[t];
                    if (BACKWARD && MASK) {
                        v.x = (tb[t] & 1u) ? v.x : 0.f;
                        v.y = (tb[t] & 2u) ? v.y : 0.f;
                        v.z = (tb[t] & 4u) ? v.z : 0.f;
                        v.w = (tb[t] & 8u) ? v.w : 0.f;
                    } else if (BACKWARD && ACT != ACT_NONE) {
                        v.x = act_bwd<ACT>(v.x, to[t].x);
                        v.y = act_bwd<ACT>(v.y, to[t].y);
                        v.z = act_bwd<ACT>(v.z, to[t].z);
                        v.w = act_bwd<ACT>(v.w, to[t].w);
                    }
                    acc.x += wv[t] * v.x;
                    acc.y += wv[t] * v.y;
                    acc.z += wv[t] * v.z;
                    acc.w += wv[t] * v.w;
                }
            }
        }
        float *yrow = a.y + row * a.c;
        unsigned sign_bits = 0u;
#pragma unroll
        for (int i = 0; i <= NC; ++i) {
            float4 v = i == 0 ? acc : own[i];
            if (!BACKWARD) {
                if (a.bias) {
                    const float4 bb = *reinterpret_cast<const float4 *>(a.bias + c0 + a.k * i);
                    v.x += bb.x, v.y += bb.y, v.z += bb.z, v.w += bb.w;
                }
                v.x = act_fwd<ACT>(v.x), v.y = act_fwd<ACT>(v.y), v.z = act_fwd<ACT>(v.z), v.w = act_fwd<ACT>(v.w);
                if (MASK) // out > 0, the predicate relu' is defined by (act_bwd)
                    sign_bits |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (4 * i);
            }
            *reinterpret_cast<float4 *>(yrow + c0 + a.k * i) = v;
            if (!BACKWARD && HEAD && i == 0 && j == 0) { // new positions = base + scale * the three leading channels
                const float *bp = a.head_in + row * 3;
                float *pp = a.head_out + row * 3;
                pp[0] = bp[0] + a.head_scale * v.x, pp[1] = bp[1] + a.head_scale * v.y, pp[2] = bp[2] + a.head_scale * v.z;
            }
        }
        if (!BACKWARD && MASK) a.mask[row * kg + j] = (unsigned short)sign_bits;
    }
    if (BACKWARD) {
#pragma unroll
        for (int i = 0; i <= NC; ++i)
            csum[i].x += own[i].x, csum[i].y += own[i].y, csum[i].z += own[i].z, csum[i].w += own[i].w;
    }
    } // row tiles

    if (BACKWARD && colsum_partial) {
        if (rl < rows_per_block) {
#pragma unroll
            for (int i = 0; i <= NC; ++i)
                *reinterpret_cast<float4 *>(lds_cs + rl * a.c + c0 + a.k * i) = csum[i]; // zeros for inactive rows
        }
        __syncthreads();
        for (int c = threadIdx.x; c < a.c; c += GCN_THREADS) {
            float t = 0.f;
            for (int l = 0; l < rows_per_block; ++l) t += lds_cs[l * a.c + c];
            colsum_partial[((size_t)mesh_i * a.chunks + chunk_i) * a.c + c] = t;
        }
    }
}

inline int ell_rows_per_block(int k) { return GCN_THREADS / (k >> 2); }
inline int ell_nc(int c, int k) { return (k > 0 && k % 4 == 0 && c % k == 0) ? c / k - 1 : -1; }
inline bool ell_supported(int c, int k, int w)
{
    const int nc = ell_nc(c, k);
    return (nc == 2 || nc == 9) && (w == 8 || w == 16) && (k >> 2) <= GCN_THREADS;
}

template <int ACT, bool BACKWARD>
void launch_ell_shape(const EllArgs &a, int w, dim3 grid, size_t lds, hipStream_t s, int rpb, int iters, float *partial)
{
    const dim3 block(GCN_THREADS);
    const int nc = ell_nc(a.c, a.k);
    if (a.head_in) { // dispatch_ell admits it for split 3 with ReLU + mask or without activation only
        if constexpr (ACT == ACT_RELU) {
            if (w == 8) hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 8, 2, true, true>), grid, block, lds, s, a, rpb, iters, partial);
            else hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 16, 2, true, true>), grid, block, lds, s, a, rpb, iters, partial);
        } else if constexpr (ACT == ACT_NONE) {
            if (w == 8) hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 8, 2, false, true>), grid, block, lds, s, a, rpb, iters, partial);
            else hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 16, 2, false, true>), grid, block, lds, s, a, rpb, iters, partial);
        }
        return;
    }
    if constexpr (ACT == ACT_RELU) {
        if (a.mask && nc == 2) {
            if (w == 8) hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 8, 2, true>), grid, block, lds, s, a, rpb, iters, partial);
            else hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 16, 2, true>), grid, block, lds, s, a, rpb, iters, partial);
            return;
        }
    }
    if (w == 8 && nc == 2) hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 8, 2>), grid, block, lds, s, a, rpb, iters, partial);
    else if (w == 16 && nc == 2) hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 16, 2>), grid, block, lds, s, a, rpb, iters, partial);
    else if (w == 8 && nc == 9) hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 8, 9>), grid, block, lds, s, a, rpb, iters, partial);
    else hipLaunchKernelGGL((zn_aggregate_ell_kernel<ACT, BACKWARD, 16, 9>), grid, block, lds, s, a, rpb, iters, partial);
}

// ---------------------------------------------------------------------------------------
// ELL table, ANY width and split: the unbatched ZERON_GCN layers of the mesh encoder (reference layers.py:34-41 with
// models.py:299-348's widths: c = 60 ... 300, k = c / 10 = 6 ... 30 -- neither k % 4 == 0 nor c % k == 0 holds for most).
// The thread layout of the generic CSR kernel at the top of this file (one thread = VEC consecutive columns of a row, the
// group that straddles column k is mixed) with the neighbour entries from the fixed-stride table instead of the
// rowptr -> (col, val) chain: two dependent round trips per gathering thread instead of three, the table read issued with
// the thread's own elements.  VEC = 4 / 2 / 1 by the divisibility of c (150, 210, 250 are even, not multiples of 4).
// Workgroups run several row tiles (forward: keeps the grid at a few thousand workgroups; backward: at most ~1024
// bias-gradient partials to reduce -- 3072 partial rows of 300 columns cost an 8 us reduction launch per layer).
// Summation order = table order = CSR order: bit-identical to the generic kernel.
// ---------------------------------------------------------------------------------------
template <int VEC, int ACT, bool BACKWARD, int W>
__global__ __launch_bounds__(GCN_THREADS) void zn_aggregate_ell_any_kernel(EllArgs a, int groups, int rows_in_flight, int rows_per_block,
                                                                            float *colsum_partial)
{
    extern __shared__ float lds_colsum[]; // [rows_in_flight][c], backward with colsum only
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(r_begin + rows_per_block, a.nv);
    const int64_t mesh_row0 = (int64_t)blockIdx.y * a.nv;

    // ---- the aggregated columns: COMPACT gather threads.  Item (row, j): group j < kgs of the row (kgs = the groups that hold
    // an aggregated column, the straddling one included); every lane of a wave gathers -- with the gathers left to the few
    // lanes of a row-major layout that own aggregated columns (8 of 75 at c = 300) a wave issued its 12 memory instructions
    // for 128 useful bytes each, and the launch took 18.7 us against 10.9 for the copy alone (tools/probe/agg_any_probe.py)
    const int kgs = (a.k + VEC - 1) / VEC;
    for (int item = threadIdx.x; item < (r_end - r_begin) * kgs; item += GCN_THREADS) {
        const int r = r_begin + item / kgs, c0 = (item % kgs) * VEC;
        const int64_t row = mesh_row0 + r;
        int nb[W];
        float w[W];
#pragma unroll
        for (int n = 0; n < W; n += 4) { // round trip 1: the row's table entries
            const int4 ci = *reinterpret_cast<const int4 *>(a.col + (size_t)r * W + n);
            const float4 wi = *reinterpret_cast<const float4 *>(a.val + (size_t)r * W + n);
            nb[n] = ci.x, nb[n + 1] = ci.y, nb[n + 2] = ci.z, nb[n + 3] = ci.w;
            w[n] = wi.x, w[n + 1] = wi.y, w[n + 2] = wi.z, w[n + 3] = wi.w;
        }
        int e0 = 0, e1 = 0;
        if (a.over_ptr) e0 = a.over_ptr[r], e1 = a.over_ptr[r + 1];
        Pack<VEC> sv[W], ov[W];
#pragma unroll
        for (int n = 0; n < W; ++n) { // round trip 2: the neighbour rows
            const int64_t nrow = mesh_row0 + (nb[n] >= 0 ? nb[n] : r);
            sv[n].load(a.x + nrow * a.c + c0);
            if (BACKWARD && ACT != ACT_NONE) ov[n].load(a.saved + nrow * a.c + c0);
        }
        Pack<VEC> acc;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.at(i) = 0.f;
#pragma unroll
        for (int n = 0; n < W; ++n) {
            if (nb[n] >= 0) { // table order == CSR order of the row
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float v = sv[n].at(i);
                    if (BACKWARD && ACT != ACT_NONE) v = act_bwd<ACT>(v, ov[n].at(i));
                    acc.at(i) += w[n] * v;
                }
            }
        }
        for (int e = e0; e < e1; e += GCN_NB) { // a row longer than the table: its CSR tail, still in order
            int64_t tb[GCN_NB];
            float tw[GCN_NB];
#pragma unroll
            for (int j = 0; j < GCN_NB; ++j) {
                const bool in = e + j < e1;
                tb[j] = in ? mesh_row0 + a.over_col[e + j] : row;
                tw[j] = in ? a.over_val[e + j] : 0.f;
            }
            Pack<VEC> tv[GCN_NB], to[GCN_NB];
#pragma unroll
            for (int j = 0; j < GCN_NB; ++j) {
                tv[j].load(a.x + tb[j] * a.c + c0);
                if (BACKWARD && ACT != ACT_NONE) to[j].load(a.saved + tb[j] * a.c + c0);
            }
#pragma unroll
            for (int j = 0; j < GCN_NB; ++j) {
                if (e + j < e1) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        float v = tv[j].at(i);
                        if (BACKWARD && ACT != ACT_NONE) v = act_bwd<ACT>(v, to[j].at(i));
                        acc.at(i) += tw[j] * v;
                    }
                }
            }
        }
        if (!BACKWARD) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float v = acc.at(i);
                if (a.bias && c0 + i < a.k) v += a.bias[c0 + i];
                acc.at(i) = act_fwd<ACT>(v);
            }
        }
        if (c0 + VEC <= a.k) {
            acc.store(a.y + row * a.c + c0);
        } else { // the straddling group: its aggregated elements only (the others belong to the pass-through threads below)
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (c0 + i < a.k) a.y[row * a.c + c0 + i] = acc.at(i);
        }
    }

    // ---- everything else, row-major: thread (rl, g) owns VEC consecutive columns of rows rl, rl + rows_in_flight, ...: the
    // pass-through columns (copy + bias + activation / derivative) and, in the backward, the bias-gradient terms of ALL columns
    const int g = threadIdx.x % groups;
    const int rl = threadIdx.x / groups;
    const int c0 = g * VEC;
    const bool active = rl < rows_in_flight;
    const bool pass = c0 + VEC > a.k; // pass-through columns in the group (all, or the tail of the straddling group)

    Pack<VEC> colsum;
#pragma unroll
    for (int i = 0; i < VEC; ++i) colsum.at(i) = 0.f;

    if (active && (BACKWARD || pass)) {
        for (int r = r_begin + rl; r < r_end; r += rows_in_flight) {
            const int64_t row = mesh_row0 + r;
            Pack<VEC> own;
            own.load(a.x + row * a.c + c0);
            if (BACKWARD) {
                if (ACT != ACT_NONE) {
                    Pack<VEC> o;
                    o.load(a.saved + row * a.c + c0);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) own.at(i) = act_bwd<ACT>(own.at(i), o.at(i));
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) colsum.at(i) += own.at(i);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float v = own.at(i);
                    if (a.bias) v += a.bias[c0 + i];
                    own.at(i) = act_fwd<ACT>(v);
                }
            }
            if (c0 >= a.k) {
                own.store(a.y + row * a.c + c0);
            } else if (pass) {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (c0 + i >= a.k) a.y[row * a.c + c0 + i] = own.at(i);
            }
        }
    }

    if (BACKWARD && colsum_partial) { // block partial of the bias gradient, fixed summation order
        if (active) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) lds_colsum[rl * a.c + c0 + i] = colsum.at(i);
        }
        __syncthreads();
        if (active && rl == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float t = 0.f;
                for (int l = 0; l < rows_in_flight; ++l) t += lds_colsum[l * a.c + c0 + i];
                colsum_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * a.c + c0 + i] = t;
            }
        }
    }
}

inline int ell_any_vec(int c) { return c % 4 == 0 ? 4 : c % 2 == 0 ? 2 : 1; }
inline bool ell_any_supported(int c, int w) { return (w == 8 || w == 16) && c > 0 && c / ell_any_vec(c) <= GCN_THREADS; }

inline GcnGeometry ell_any_geometry(int b, int nv, int c, bool backward)
{
    GcnGeometry g;
    g.groups = c / ell_any_vec(c);
    g.rows_in_flight = GCN_THREADS / g.groups;
    // row tiles per workgroup: at most ~1024 workgroups in either direction (backward: = bias-gradient partials to reduce;
    // forward: a 300-column copy alone ran at 3.3 TB/s as 3072 two-tile workgroups and at 4.0 as 1024 six-tile ones)
    const int64_t tiles = ((int64_t)b * nv + g.rows_in_flight - 1) / g.rows_in_flight;
    int64_t iters = (tiles + 1023) / 1024;
    const int64_t lo = backward ? GCN_BWD_ITERS : 1, hi = 64;
    iters = iters < lo ? lo : iters > hi ? hi : iters;
    g.rows_per_block = g.rows_in_flight * (int)iters;
    g.chunks = (nv + g.rows_per_block - 1) / g.rows_per_block;
    g.blocks = (int64_t)g.chunks * b;
    return g;
}

template <int VEC, int ACT, bool BACKWARD>
void launch_ell_any_w(const EllArgs &a, int w, const GcnGeometry &geo, dim3 grid, size_t lds, hipStream_t s, float *partial)
{
    if (w == 8) hipLaunchKernelGGL((zn_aggregate_ell_any_kernel<VEC, ACT, BACKWARD, 8>), grid, dim3(GCN_THREADS), lds, s, a, geo.groups, geo.rows_in_flight, geo.rows_per_block, partial);
    else hipLaunchKernelGGL((zn_aggregate_ell_any_kernel<VEC, ACT, BACKWARD, 16>), grid, dim3(GCN_THREADS), lds, s, a, geo.groups, geo.rows_in_flight, geo.rows_per_block, partial);
}

template <int ACT, bool BACKWARD>
void launch_ell_any_vec(const EllArgs &a, int w, const GcnGeometry &geo, dim3 grid, size_t lds, hipStream_t s, float *partial)
{
    switch (ell_any_vec(a.c)) {
    case 4: launch_ell_any_w<4, ACT, BACKWARD>(a, w, geo, grid, lds, s, partial); break;
    case 2: launch_ell_any_w<2, ACT, BACKWARD>(a, w, geo, grid, lds, s, partial); break;
    default: launch_ell_any_w<1, ACT, BACKWARD>(a, w, geo, grid, lds, s, partial); break;
    }
}

template <bool BACKWARD>
int dispatch_ell_any(EllArgs a, int b, int w, int act, float *grad_bias, float *scratch, void *stream)
{
    if (!ell_any_supported(a.c, w) || a.head_in || a.mask) return GEOM_EUNSUPPORTED;
    if (b == 0 || a.nv == 0) return 0;
    if (!a.col || !a.val || !a.y || !a.x) return GEOM_EINVAL;
    if (BACKWARD && act != ACT_NONE && !a.saved) return GEOM_EINVAL;
    if (grad_bias && !scratch) return GEOM_EINVAL;
    if (a.over_ptr && (!a.over_col || !a.over_val)) return GEOM_EINVAL;
    // the table rows are read 16 bytes at a time; the operands VEC floats at a time (rows of c floats: VEC divides c)
    if ((((uintptr_t)a.col | (uintptr_t)a.val) % 16) != 0) return GEOM_EINVAL;
    if ((((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.saved) % (4 * ell_any_vec(a.c))) != 0) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    const GcnGeometry geo = ell_any_geometry(b, a.nv, a.c, BACKWARD);
    float *partial = scratch;
    const size_t lds = (BACKWARD && partial) ? (size_t)geo.rows_in_flight * a.c * sizeof(float) : 0;
    const dim3 grid((unsigned)geo.chunks, (unsigned)b);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (act) {
    case ACT_NONE: launch_ell_any_vec<ACT_NONE, BACKWARD>(a, w, geo, grid, lds, s, partial); break;
    case ACT_RELU: launch_ell_any_vec<ACT_RELU, BACKWARD>(a, w, geo, grid, lds, s, partial); break;
    case ACT_ELU: launch_ell_any_vec<ACT_ELU, BACKWARD>(a, w, geo, grid, lds, s, partial); break;
    default: return GEOM_EINVAL;
    }
    if (BACKWARD && partial && grad_bias)
        hipLaunchKernelGGL(colsum_final_kernel, dim3((a.c + CS_COLS - 1) / CS_COLS), dim3(CS_COLS * CS_LANES), 0, s, (int)geo.blocks, a.c,
                           partial, grad_bias);
    return geom::launch_status();
}

template <bool BACKWARD>
int dispatch_ell(EllArgs a, int b, int w, int act, float *grad_bias, float *scratch, void *stream)
{
    if (b < 0 || a.nv < 0 || a.c < 0 || a.k < 0 || a.k > a.c) return GEOM_EINVAL;
    if (!ell_supported(a.c, a.k, w)) return dispatch_ell_any<BACKWARD>(a, b, w, act, grad_bias, scratch, stream); // any width / split
    if (b == 0 || a.nv == 0) return 0;
    if (!a.col || !a.val || !a.y || (!a.x && !(BACKWARD && a.head_in))) return GEOM_EINVAL;
    if (a.head_in) { // head mode: split 3, ReLU with the sign mask or no activation; the forward also needs head_out
        if (ell_nc(a.c, a.k) != 2 || !((act == ACT_RELU && a.mask) || act == ACT_NONE) || (!BACKWARD && !a.head_out))
            return GEOM_EUNSUPPORTED;
    }
    if (a.mask && !(act == ACT_RELU && ell_nc(a.c, a.k) == 2)) return GEOM_EINVAL; // sign mask: ReLU, split 3 only
    if (BACKWARD && act != ACT_NONE && !a.saved && !a.mask) return GEOM_EINVAL;
    if (grad_bias && !scratch) return GEOM_EINVAL;
    if (a.over_ptr && (!a.over_col || !a.over_val)) return GEOM_EINVAL;
    if ((((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.saved | (uintptr_t)a.bias | (uintptr_t)a.col | (uintptr_t)a.val) % 16) != 0)
        return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    const int rpb = ell_rows_per_block(a.k);
    // row tiles per workgroup (measured at the BASELINE shard, us: forward 8.1 / 8.4 / 9.2 / 11.1 for 1-4 tiles --
    // one row per thread keeps the most loads in flight; backward incl. the bias reduction 14.6 / 12.8 / 13.7 / 14.8)
    const int iters = BACKWARD ? 2 : 1;
    const int chunks = (a.nv + rpb * iters - 1) / (rpb * iters);
    float *partial = scratch; // with grad_bias == null the partials stay un-reduced (geom_colsum_batch_f32 finishes them)
    const size_t lds = (BACKWARD && partial) ? (size_t)rpb * a.c * sizeof(float) : 0;
    a.b = b;
    a.chunks = chunks;
    dim3 grid(geom::xcd_grid(b, chunks));
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (act) {
    case ACT_NONE: launch_ell_shape<ACT_NONE, BACKWARD>(a, w, grid, lds, s, rpb, iters, partial); break;
    case ACT_RELU: launch_ell_shape<ACT_RELU, BACKWARD>(a, w, grid, lds, s, rpb, iters, partial); break;
    case ACT_ELU: launch_ell_shape<ACT_ELU, BACKWARD>(a, w, grid, lds, s, rpb, iters, partial); break;
    default: return GEOM_EINVAL;
    }
    if (BACKWARD && partial && grad_bias)
        hipLaunchKernelGGL(colsum_final_kernel, dim3((a.c + CS_COLS - 1) / CS_COLS), dim3(CS_COLS * CS_LANES), 0, s,
                           chunks * b, a.c, partial, grad_bias);
    return geom::launch_status();
}

template <bool BACKWARD>
int dispatch(GcnArgs a, int b, int act, float *grad_bias, float *scratch, void *stream)
{
    if (b < 0 || a.nv < 0 || a.c < 0 || a.k < 0 || a.k > a.c) return GEOM_EINVAL;
    if (b == 0 || a.nv == 0 || a.c == 0) return 0;
    if (!a.rowptr || !a.x || !a.y || (a.k > 0 && (!a.col || !a.val))) return GEOM_EINVAL;
    if (BACKWARD && act != ACT_NONE && !a.saved) return GEOM_EINVAL;
    if (grad_bias && !scratch) return GEOM_EINVAL;
    a.rows = (int64_t)b * a.nv;
    if (gcn_vec4(a.c, a.k) && (((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.saved) % 16 != 0))
        return GEOM_EINVAL; // row-major fp32 tensors from any allocator are 16-byte aligned; refuse odd views
    const bool vec4 = gcn_vec4(a.c, a.k);
    float *partial = scratch;
    if (!vec4 && a.c > GCN_THREADS) return GEOM_ETOOBIG;
    return vec4 ? launch<4, BACKWARD>(a, act, partial, grad_bias, stream) : launch<1, BACKWARD>(a, act, partial, grad_bias, stream);
}

} // namespace

extern "C" int geom_zn_gcn_aggregate_fwd_f32(int b, int nv, int c, int k, const int *rowptr, const int *col,
                                             const float *val, const float *support, const float *bias,
                                             int act, float *out, void *stream)
{
    GcnArgs a{rowptr, col, val, support, bias, nullptr, out, 0, nv, c, k};
    return dispatch<false>(a, b, act, nullptr, nullptr, stream);
}

extern "C" int64_t geom_zn_gcn_bwd_scratch_floats(int b, int nv, int c)
{
    if (b <= 0 || nv <= 0 || c <= 0) return 0;
    // upper bound over both vector widths (k is not known here): the scalar layout has fewer rows per block
    const int64_t b4 = (c % 4 == 0) ? gcn_geometry(b, nv, c, 4, true).blocks : 0;
    const int64_t b1 = gcn_geometry(b, nv, c, 1, true).blocks;
    return (b4 > b1 ? b4 : b1) * c;
}

// Rows of per-workgroup partial column sums the backward launches write into `scratch` for this shape: ell_w = the table
// width of the geom_zn_gcn_aggregate_ell_bwd_f32 call (8 / 16), 0 = geom_zn_gcn_aggregate_bwd_f32.  0: unsupported shape.
extern "C" int64_t geom_zn_gcn_bwd_partial_rows(int b, int nv, int c, int k, int ell_w)
{
    if (b <= 0 || nv <= 0 || c <= 0 || k < 0 || k > c) return 0;
    if (ell_w) {
        if (!ell_supported(c, k, ell_w)) return ell_any_supported(c, ell_w) ? ell_any_geometry(b, nv, c, true).blocks : 0;
        const int per = ell_rows_per_block(k) * 2; // two row tiles per workgroup in the backward
        return (int64_t)((nv + per - 1) / per) * b;
    }
    return gcn_geometry(b, nv, c, gcn_vec4(c, k) ? 4 : 1, true).blocks;
}

extern "C" int geom_colsum_batch_f32(int count, const float *const *partials, const int *rows, const int *cols,
                                     float *const *outs, void *stream)
{
    if (count < 0 || count > GEOM_COLSUM_MAX_JOBS) return GEOM_ETOOBIG;
    if (count == 0) return 0;
    if (!partials || !rows || !cols || !outs) return GEOM_EINVAL;
    ColsumJobs jobs;
    int widest = 0;
    for (int i = 0; i < count; ++i) {
        if (!partials[i] || !outs[i] || rows[i] < 0 || cols[i] <= 0) return GEOM_EINVAL;
        jobs.partial[i] = partials[i], jobs.out[i] = outs[i], jobs.blocks[i] = rows[i], jobs.c[i] = cols[i];
        widest = cols[i] > widest ? cols[i] : widest;
    }
    hipLaunchKernelGGL(colsum_batch_kernel, dim3((widest + CS_COLS - 1) / CS_COLS, count), dim3(CS_COLS * CS_LANES), 0,
                       static_cast<hipStream_t>(stream), jobs);
    return geom::launch_status();
}

extern "C" int geom_zn_gcn_aggregate_bwd_f32(int b, int nv, int c, int k, const int *rowptrT, const int *colT,
                                             const float *valT, const float *grad_out, const float *out,
                                             int act, float *grad_support, float *grad_bias, float *scratch,
                                             void *stream)
{
    GcnArgs a{rowptrT, colT, valT, grad_out, nullptr, out, grad_support, 0, nv, c, k};
    return dispatch<true>(a, b, act, grad_bias, scratch, stream);
}

extern "C" int geom_zn_gcn_aggregate_ell_fwd_f32(int b, int nv, int c, int k, int w, const int *ell_col,
                                                 const float *ell_val, const int *over_ptr, const int *over_col,
                                                 const float *over_val, const float *support, const float *bias,
                                                 int act, float *out, uint16_t *relu_mask, void *stream)
{
    EllArgs a{ell_col, ell_val, support, bias, nullptr, out, nv, c, k, relu_mask, over_ptr, over_col, over_val};
    return dispatch_ell<false>(a, b, w, act, nullptr, nullptr, stream);
}

extern "C" int geom_zn_gcn_aggregate_ell_head_fwd_f32(int b, int nv, int c, int k, int w, const int *ell_col,
                                                      const float *ell_val, const int *over_ptr, const int *over_col,
                                                      const float *over_val, const float *support, const float *bias,
                                                      int act, float *out, uint16_t *relu_mask, const float *base,
                                                      float scale, float *pos, void *stream)
{
    if (!base || !pos) return GEOM_EINVAL;
    EllArgs a{ell_col, ell_val, support, bias, nullptr, out, nv, c, k, relu_mask, over_ptr, over_col, over_val, 0, 0, base, pos, scale};
    return dispatch_ell<false>(a, b, w, act, nullptr, nullptr, stream);
}

extern "C" int64_t geom_zn_gcn_relu_mask_words(int b, int nv, int c, int k)
{
    if (b <= 0 || nv <= 0 || ell_nc(c, k) != 2) return 0; // 0: this shape has no mask path
    return (int64_t)b * nv * (k >> 2);
}

extern "C" int geom_zn_gcn_aggregate_ell_bwd_f32(int b, int nv, int c, int k, int w, const int *ell_colT,
                                                 const float *ell_valT, const int *over_ptrT, const int *over_colT,
                                                 const float *over_valT, const float *grad_out, const float *out,
                                                 const uint16_t *relu_mask, int act, float *grad_support,
                                                 float *grad_bias, float *scratch, void *stream)
{
    EllArgs a{ell_colT, ell_valT, grad_out, nullptr, out, grad_support, nv, c, k, const_cast<uint16_t *>(relu_mask),
              over_ptrT, over_colT, over_valT};
    return dispatch_ell<true>(a, b, w, act, grad_bias, scratch, stream);
}

extern "C" int geom_zn_gcn_aggregate_ell_head_bwd_f32(int b, int nv, int c, int k, int w, const int *ell_colT,
                                                      const float *ell_valT, const int *over_ptrT, const int *over_colT,
                                                      const float *over_valT, const float *grad_pos, float scale,
                                                      const uint16_t *relu_mask, int act, float *grad_support,
                                                      float *grad_bias, float *scratch, void *stream)
{
    if (!grad_pos) return GEOM_EINVAL;
    EllArgs a{ell_colT, ell_valT, nullptr, nullptr, nullptr, grad_support, nv, c, k, const_cast<uint16_t *>(relu_mask),
              over_ptrT, over_colT, over_valT, 0, 0, grad_pos, nullptr, scale};
    return dispatch_ell<true>(a, b, w, act, grad_bias, scratch, stream);
}
