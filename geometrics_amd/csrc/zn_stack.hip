// A 0N-GCN layer boundary in ONE launch on the gfx950 matrix cores (exact fp32, v_mfma_f32_16x16x4_f32):
//
//   forward :  X = act([A . S_prev[:, :k] | S_prev[:, k:]] + bias_prev)      (the aggregation of the PREVIOUS layer, zn_gcn.hip)
//              S = X . W                                                     (this layer's product)
//   backward:  G = [A^T . g'[:, :k] | g'[:, k:]],  g' = grad_out * act'(out) (this layer's aggregation backward)
//              grad_in = G . W^T                                             (this layer's input gradient)
//
// reference: layers.py:107-116 (`support = matmul(input, weight1)`, `matmul(adj, support[..., :k])`, cat, + bias) and the
// autograd graph of those lines.  As two launches per direction and layer these are a library product (16.6 us for the
// 20 496 x 192 x 192 hidden layers of the BASELINE shard = 0.58 of the matrix peak) and `zn_aggregate_ell_kernel`
// (9.2 us, two dependent gather round trips + launch + drain: 0.38 of HBM), with the 15.7 MB activation written by one
// and re-read by the other.  Here the aggregation IS the product's operand load: 23.7 us for the forward boundary against
// 25.8 (rocprofv3 kernel durations, profiles/r05_fused_boundary.txt) -- a launch saved, not the 10 us an overlap of the
// two roles would have been worth: see the second point.
//
//   * the weight operand never touches LDS: an MFMA wave owns 48 output columns and keeps its whole 192 x 48 slice of W in
//     144 registers for the life of the launch (12-byte loads, lane (x, g) holds W[k][48 wave + 3 x .. + 2] for its 48
//     values of k) -- no B panels, no fragment reads, no per-stage barrier, and one row-block (16 rows) at a time needs
//     only 3 accumulators, so rows finish progressively instead of all at the end;
//   * TWO KINDS OF WAVES per workgroup, one of each on every SIMD: waves 0-3 issue nothing but fragment reads and MFMAs
//     (33.2 cycles per MFMA: 0.965 of the pipe), waves 4-7 produce the activation operand and move every byte to and
//     from memory.  What this does and does not buy was measured (tools/probe/corun.cpp, profiles/r05_mfma_corun.txt): an
//     fp32 MFMA occupies its SIMD's issue port for all of its 32 cycles -- it IS the SIMD's fp32 vector unit -- so beside a
//     wave that streams them a second wave of the same SIMD issues NOTHING (VALU, buffer loads, LDS reads and stores
//     alike: its time is added, at any s_setprio), and inside one wave every other instruction is added to the MFMA time
//     as well (a memory instruction ~50 cycles, a VALU 2-4).  The two kinds of wave therefore ALTERNATE on a SIMD: the
//     gather waves run while the MFMA waves sit at the row-block's barrier.  What the split buys is that the gather
//     waves' memory round trips pass under the other kind's MFMAs (their loads are requested a whole turn ahead), that
//     neither instruction stream is bent around the other (the first version dealt the gather role out between the
//     MFMAs of ONE kind of wave with scheduling patterns: the same 24 us, and unreadable), and the launches it replaces:
//     one dispatch, one prologue, one drain instead of two, and 15.7 MB less traffic.
//   * gather thread (row, j) of a row-block owns the aggregated float4 j of its row and the two pass-through float4s
//     j + 16, j + 32 -- exactly the thread of zn_aggregate_ell_kernel, same neighbour order, same arithmetic, same bits --
//     finishes them (bias + activation + sign bits, or the masked transpose gather of the backward), writes them to
//     memory ONCE (the next launch's weight gradient needs them) and drops them into a 13 KB LDS panel; the output tile
//     leaves through LDS as well and is written by the gather waves in memory order (12 KB contiguous per row-block);
//   * per row-block the gather waves request the rows of row-block i + 2, store what block i - 1 produced and finish
//     row-block i + 1 (requested a turn ago) into the other panel: ~2 200 cycles of issue per SIMD beside the 4 608 of the
//     MFMAs, ONE workgroup barrier per row-block; the neighbour table of the workgroup's rows sits resolved in LDS;
//   * the panel is laid out [k-quarter g][row][52] so that the four lane groups of ds_read_b128 never share a bank
//     (row pitch 13 x 16 B, quarter pitch a multiple of 256 B), and the k index of MFMA step (j', c) of lane group g is
//     48 g + 4 j' + c: one 16-byte fragment read feeds four k-steps = 12 MFMAs.
//
// The backward consumes the weight TRANSPOSED ([c, cin] row-major) so that its register slice loads with the same
// contiguous 192-byte runs; the forward launch of the same layer emits that copy on the side (workgroup 0: 147 KB).
//
// Row-blocks are dealt to workgroups in contiguous runs per XCD (a mesh's rows stay in one L2: its k-slice is gathered
// ~7 times); the few row-blocks beyond an equal share (1281 = 5 * 256 + 1 at the BASELINE shard) are not given whole to
// single workgroups (+20 % for the launch) but cut three ways by column component, +48 MFMAs on 720 for three workgroups.
#include "geom_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
struct __attribute__((packed, aligned(4))) f3u { float x, y, z; };

constexpr int ZS_ROLE = 256;        // threads of one role: 4 waves, one per SIMD
constexpr int ZS_THREADS = 2 * ZS_ROLE;
constexpr int ZS_C = 192;           // inner dimension of the product = width of the aggregated operand
constexpr int ZS_K = 64;            // aggregated columns (split 3)
constexpr int ZS_KG = ZS_K / 4;     // float4 groups of the aggregated slice = gather threads per row
constexpr int ZS_W = 8;             // neighbour-table width
constexpr int ZS_LDR = 52;          // floats per (quarter, row) line of the panel: 48 used, pitch 13 x 16 B (odd)
constexpr int ZS_SUB = 16 * ZS_LDR; // one k-quarter of the panel: 832 floats, a multiple of 64 dwords
constexpr int ZS_PANEL = 4 * ZS_SUB;
constexpr int ZS_LDC = ZS_C + 4;    // row pitch of the output staging tile (196 % 32 == 4: the b128 writes of 8 rows hit 32 banks)
constexpr int ZS_CST = 16 * ZS_LDC; // one 16 x 192 output tile

enum { ZS_ACT_NONE = 0, ZS_ACT_RELU = 1, ZS_ACT_ELU = 2 };
enum { ZS_FWD = 0, ZS_BWD = 1 };

struct ZsArgs {
    // operand source
    const float *src;               // fwd: previous layer's raw support [rows, 192]; bwd: gradient of this layer's output
    const float *bias;              // fwd: previous layer's bias [192] or null
    const float *saved;             // bwd, ELU: this layer's activated output
    const unsigned short *mask_in;  // bwd, ReLU: sign words of this layer's output (zn_gcn.hip layout: [rows][16])
    const float *head_gp;           // bwd, head: gradient of the positions [rows, 3]; src is not read
    float head_scale;
    const int *ell_col;             // [nv][8] (-1 = padding): A for the forward, A^T for the backward
    const float *ell_val;
    // operand sinks
    float *a_out;                   // fwd: activated input X [rows, 192]; bwd: G [rows, 192]
    unsigned short *mask_out;       // fwd, ReLU: sign words of X
    float *colsum_partial;          // bwd: [workgroups][192] column sums of g' over the workgroup's rows, or null
    // product
    const float *bmat;              // [192][ldb]: W (forward) or W^T (backward)
    int ldb;
    float *c_out;                   // [rows][ldc]
    int ldc, n_out;                 // output columns (<= 192, % 12 == 0: whole lanes of the weight slice)
    float *bt_out;                  // fwd: transposed copy of bmat, [n_out][192], or null
    int rows, nv;
    int q, rem, split3;             // row-blocks per workgroup; leftover row-blocks; leftovers cut three ways
};

__device__ __forceinline__ f3u ldg3(const float *p) { return *reinterpret_cast<const f3u *>(p); }

#ifdef ZS_PROBE_STAMPS
// probe build (tools/probe/zs_variants.sh): shader-clock stamps of wave 0 of every workgroup at the phase boundaries
constexpr int ZS_STAMP_SLOTS = 64;
__device__ unsigned long long zs_stamps[1024 * ZS_STAMP_SLOTS];
#define ZS_STAMP(i)                                                                                                   \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        if ((threadIdx.x & 255) == 0 && (i) < ZS_STAMP_SLOTS / 2)                                                    \
            zs_stamps[blockIdx.x * ZS_STAMP_SLOTS + ZS_STAMP_ROLE * (ZS_STAMP_SLOTS / 2) + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
#else
#define ZS_STAMP(i) do { } while (0)
#endif

template <int ACT>
__device__ __forceinline__ float zs_act(float v)
{
    if (ACT == ZS_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ZS_ACT_ELU) return v > 0.f ? v : expm1f(v);
    return v;
}
template <int ACT>
__device__ __forceinline__ float zs_act_bwd(float g, float out)
{
    if (ACT == ZS_ACT_RELU) return out > 0.f ? g : 0.f;
    if (ACT == ZS_ACT_ELU) return out > 0.f ? g : g * (out + 1.f);
    return g;
}
__device__ __forceinline__ float4 zs_masked(float4 v, unsigned m)
{
    v.x = (m & 1u) ? v.x : 0.f, v.y = (m & 2u) ? v.y : 0.f, v.z = (m & 4u) ? v.z : 0.f, v.w = (m & 8u) ? v.w : 0.f;
    return v;
}

// the rows a row-block's table entries point at, in flight / arrived
template <int MODE, int ACT>
struct ZsGather {
    float4 sv[ZS_W], own[3];
    unsigned own_bits;
    unsigned nbits[MODE == ZS_BWD && ACT == ZS_ACT_RELU ? ZS_W : 1];
    float4 ov[MODE == ZS_BWD && ACT == ZS_ACT_ELU ? ZS_W : 1], oo[MODE == ZS_BWD && ACT == ZS_ACT_ELU ? 3 : 1];
};

// buffer addressing: a 32-bit byte offset per lane against a wave-uniform descriptor (half the address traffic of a flat
// access per instruction -- a VMEM instruction between two MFMAs costs its issue time), loads beyond the range return 0 and
// stores beyond it are dropped: predication without a branch in the MFMA stream
constexpr unsigned ZS_OOB = 0x80000000u; // beyond every range here (zs_check_common bounds the arrays to < 2 GB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t zs_rsrc(const void *p, int64_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ float4 zs_ld4(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void zs_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
                                           r, off, 0, 0);
}

// what both roles agree on: this workgroup's row-blocks
struct ZsPlan {
    int base, q, n_main, extra_rb, ue;
    __device__ __forceinline__ int rb_of(int it) const { return it < q ? base + it : (it == q ? extra_rb : -1); }
};
__device__ __forceinline__ ZsPlan zs_plan(const ZsArgs &a, int w, int G)
{
    ZsPlan p;
    p.q = a.q, p.n_main = a.q, p.extra_rb = -1, p.ue = -1;
    if ((G & 7) == 0) p.base = ((w & 7) * (G >> 3) + (w >> 3)) * a.q; // contiguous runs per XCD (workgroup w runs on XCD w % 8)
    else p.base = w * a.q;
    if (a.split3) {
        if (w < 3 * a.rem) p.extra_rb = a.q * G + w / 3, p.ue = w % 3; // a third of a leftover row-block (column component ue)
    } else if (w < a.rem) {
        p.extra_rb = a.q * G + w, ++p.n_main;                          // a whole extra row-block, processed like the others
    }
    return p;
}

// ---- MFMA role (waves 0-3) -------------------------------------------------------------------------------------------
// Per row-block: 12 fragment reads + 144 MFMAs out of panel it % 2, the output tile to staging tile it % 2, one barrier.
// Accumulator u of lane (x', g') holds C[row x'][48 wave + 12 g' + 3 r + u], r = 0..3 (the weight fragment is the
// instruction's first operand): twelve consecutive output columns per lane.
__device__ __forceinline__ void zs_mfma_role(const ZsArgs &a, float *lds, const ZsPlan &p, const int w)
{
    constexpr int ZS_STAMP_ROLE = 0;
    (void)ZS_STAMP_ROLE;
    ZS_STAMP(0);
    const int tid = threadIdx.x & (ZS_ROLE - 1), wave = tid >> 6, lane = tid & 63;
    const int x = lane & 15, g = lane >> 4;
    // the wave's slice of the weight: 192 x 48 in registers -- lane (x, g), step (jp, c): k = 48 g + 4 jp + c, columns
    // 48 wave + 3 x .. + 2 (clamped into the matrix)
    const __amdgpu_buffer_rsrc_t r_b = zs_rsrc(a.bmat, (int64_t)ZS_C * a.ldb * 4);
    const int jc = min(wave * 48 + 3 * x, a.n_out - 3);
    f3u b[12][4];
    {
        const unsigned b0 = ((unsigned)(48 * g) * (unsigned)a.ldb + (unsigned)jc) * 4u;
#pragma unroll
        for (int jp = 0; jp < 12; ++jp)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(r_b, b0 + (unsigned)(4 * jp + c) * (unsigned)a.ldb * 4u, 0, 0);
                b[jp][c] = f3u{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z)};
            }
    }
    if (a.bt_out && w == 0) { // the transposed copy the backward of this layer loads its slice from
#pragma unroll
        for (int jp = 0; jp < 12; ++jp)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = 48 * g + 4 * jp + c;
                if (wave * 48 + 3 * x + 2 < a.n_out) {
                    a.bt_out[(int64_t)(jc + 0) * ZS_C + k] = b[jp][c].x;
                    a.bt_out[(int64_t)(jc + 1) * ZS_C + k] = b[jp][c].y;
                    a.bt_out[(int64_t)(jc + 2) * ZS_C + k] = b[jp][c].z;
                }
            }
    }
    ZS_STAMP(1);
    __syncthreads(); // (the gather waves' table entries)
    __syncthreads(); // panel 0 is ready
    const float *pa = lds + g * ZS_SUB + x * ZS_LDR;
    for (int it = 0; it < p.n_main; ++it) {
        ZS_STAMP(4 + 4 * it);
        const float *panel = pa + (it & 1) * ZS_PANEL;
        f32x4 acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 af = *reinterpret_cast<const f32x4 *>(panel);
#pragma unroll
        for (int jp = 0; jp < 12; ++jp) {
            // the next group's fragment is requested in front of this group's 12 MFMAs (fenced: the compiler otherwise sinks
            // the read to its use and every group starts with an LDS round trip)
            f32x4 an = af;
            if (jp + 1 < 12) an = *reinterpret_cast<const f32x4 *>(panel + 4 * (jp + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#ifdef ZS_PROBE_NO_MFMA
                acc[0] = acc[0] + af * b[jp][c].x, acc[1] = acc[1] + af * b[jp][c].y, acc[2] = acc[2] + af * b[jp][c].z;
#else
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[jp][c].x, af[c], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[jp][c].y, af[c], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[jp][c].z, af[c], acc[2], 0, 0, 0);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            af = an;
        }
        ZS_STAMP(5 + 4 * it);
        float e[12];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 3; ++u) e[3 * r + u] = acc[u][r];
        float *dst = lds + 2 * ZS_PANEL + (it & 1) * ZS_CST + x * ZS_LDC + wave * 48 + 12 * g;
#pragma unroll
        for (int v = 0; v < 3; ++v) *reinterpret_cast<f32x4 *>(dst + 4 * v) = (f32x4){e[4 * v], e[4 * v + 1], e[4 * v + 2], e[4 * v + 3]};
        ZS_STAMP(6 + 4 * it);
        __syncthreads();
        ZS_STAMP(7 + 4 * it);
    }
    // one column component of a leftover row-block shared three ways (its panel was finished during the last block): two
    // accumulators alternate (even / odd k-steps) so that no MFMA waits for its predecessor; straight to memory
    if (p.ue >= 0) {
        const float *panel = pa + (p.n_main & 1) * ZS_PANEL;
        auto partial_block = [&](auto pick, int u) {
            f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
            f32x4 af = *reinterpret_cast<const f32x4 *>(panel);
#pragma unroll
            for (int jp = 0; jp < 12; ++jp) {
                f32x4 an = af;
                if (jp + 1 < 12) an = *reinterpret_cast<const f32x4 *>(panel + 4 * (jp + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c & 1) p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(b[jp][c]), af[c], p1, 0, 0, 0);
                    else p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(b[jp][c]), af[c], p0, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                af = an;
            }
            const int row = p.extra_rb * 16 + x;
            if (row < a.rows) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = wave * 48 + 12 * g + 3 * r + u;
#ifndef ZS_PROBE_NO_STORE
                    if (col < a.n_out) a.c_out[(int64_t)row * a.ldc + col] = p0[r] + p1[r];
#endif
                }
            }
        };
        if (p.ue == 0) partial_block([](const f3u &v) { return v.x; }, 0);
        else if (p.ue == 1) partial_block([](const f3u &v) { return v.y; }, 1);
        else partial_block([](const f3u &v) { return v.z; }, 2);
    }
}

// ---- gather role (waves 4-7) -----------------------------------------------------------------------------------------
// The neighbour table of the workgroup's rows lives RESOLVED in LDS (a ring of 16 row-blocks, refilled 8 at a time by a
// cooperative pass): per row the byte offsets of its 8 neighbour rows in the operand, their weights, the mask of real
// slots, the row's own offset -- 20 dwords -- so that a block costs five LDS reads instead of four memory instructions
// and the resolving arithmetic per thread (every instruction of this role is time the SIMD does not spend on MFMAs:
// the two kinds of wave do not overlap, they alternate -- tools/probe/corun.cpp).
constexpr int ZS_ENT = 20;                  // dwords per row entry
constexpr int ZS_RING = 16;                 // row-blocks in the ring
constexpr int ZS_RING_DW = ZS_RING * 16 * ZS_ENT;

template <int MODE, int ACT, bool HEAD>
__device__ __forceinline__ void zs_gather_role(const ZsArgs &a, float *lds, const ZsPlan &p, const int w)
{
    static_assert(!HEAD || MODE == ZS_BWD, "head mode: the backward of the layer whose leading channels are positions");
    constexpr int ZS_STAMP_ROLE = 1;
    (void)ZS_STAMP_ROLE;
    ZS_STAMP(0);
    const int tid = threadIdx.x & (ZS_ROLE - 1);
    const int rl = tid >> 4, j = tid & 15; // row of the row-block, float4 group
    const int c0 = 4 * j;
    const bool writer_extra = p.ue <= 0;   // of the three workgroups that share a leftover row-block, one writes its operand
    unsigned *ring = reinterpret_cast<unsigned *>(lds + 2 * ZS_PANEL + 2 * ZS_CST + 4 * ZS_C);

    const int64_t op_bytes = (int64_t)a.rows * ZS_C * 4;
    const __amdgpu_buffer_rsrc_t r_src = zs_rsrc(a.src, op_bytes), r_saved = zs_rsrc(a.saved, op_bytes);
    const __amdgpu_buffer_rsrc_t r_min = zs_rsrc(a.mask_in, (int64_t)a.rows * ZS_KG * 2), r_mout = zs_rsrc(a.mask_out, (int64_t)a.rows * ZS_KG * 2);
    const __amdgpu_buffer_rsrc_t r_gp = zs_rsrc(a.head_gp, (int64_t)a.rows * 12), r_aout = zs_rsrc(a.a_out, op_bytes);
    const __amdgpu_buffer_rsrc_t r_cout = zs_rsrc(a.c_out, (int64_t)a.rows * a.ldc * 4);
    const __amdgpu_buffer_rsrc_t r_col = zs_rsrc(a.ell_col, (int64_t)a.nv * ZS_W * 4), r_val = zs_rsrc(a.ell_val, (int64_t)a.nv * ZS_W * 4);

    // entries of row-blocks [first, first + 8) into their ring slots: thread s < 128 resolves row s % 16 of block first + s / 16
    auto fill = [&](int first) {
        if (tid < 128) {
            const int it = first + (tid >> 4), rr = tid & 15;
            const int rb = p.rb_of(it), row = rb * 16 + rr;
            const bool valid = rb >= 0 && row < a.rows;
            const int rowc = valid ? row : 0;
            const int mesh_row0 = rowc / a.nv * a.nv, r_in = rowc - mesh_row0;
            const unsigned off = (unsigned)r_in * (ZS_W * 4);
            const u32x4 c0v = __builtin_amdgcn_raw_buffer_load_b128(r_col, off, 0, 0), c1v = __builtin_amdgcn_raw_buffer_load_b128(r_col, off + 16, 0, 0);
            const u32x4 w0v = __builtin_amdgcn_raw_buffer_load_b128(r_val, off, 0, 0), w1v = __builtin_amdgcn_raw_buffer_load_b128(r_val, off + 16, 0, 0);
            const int nb[ZS_W] = {(int)c0v.x, (int)c0v.y, (int)c0v.z, (int)c0v.w, (int)c1v.x, (int)c1v.y, (int)c1v.z, (int)c1v.w};
            unsigned o[ZS_W], in_mask = 0u;
#pragma unroll
            for (int n = 0; n < ZS_W; ++n) {
                o[n] = (unsigned)(mesh_row0 + (nb[n] >= 0 ? nb[n] : r_in)) * (ZS_C * 4);
                in_mask |= (nb[n] >= 0 ? 1u : 0u) << n;
            }
            u32x4 *e = reinterpret_cast<u32x4 *>(ring + (((it & (ZS_RING - 1)) << 4) + rr) * ZS_ENT);
            e[0] = (u32x4){o[0], o[1], o[2], o[3]}, e[1] = (u32x4){o[4], o[5], o[6], o[7]};
            e[2] = w0v, e[3] = w1v;
            e[4] = (u32x4){in_mask, (unsigned)rowc * (ZS_C * 4), valid ? 1u : 0u, (unsigned)rowc};
        }
    };
    auto entry = [&](int it) -> const u32x4 * {
        return reinterpret_cast<const u32x4 *>(ring + (((it & (ZS_RING - 1)) << 4) + rl) * ZS_ENT);
    };

    // round trip: neighbour rows of the aggregated slot + the thread's own elements (+ sign words / saved outputs)
    auto head_at = [&](unsigned row_off) -> float4 { // [scale * grad_pos | 0] of a row: lanes j != 0 read beyond the range = zeros
        const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(r_gp, j == 0 ? row_off / (ZS_C * 4) * 12u : ZS_OOB, 0, 0);
        return make_float4(a.head_scale * __uint_as_float(t.x), a.head_scale * __uint_as_float(t.y), a.head_scale * __uint_as_float(t.z), 0.f);
    };
    auto gather = [&](ZsGather<MODE, ACT> &gv, int it) {
        const u32x4 *e = entry(it);
        const u32x4 o0 = e[0], o1 = e[1], misc = e[4];
        const unsigned o[ZS_W] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        const unsigned own = misc.y;
#pragma unroll
        for (int n = 0; n < ZS_W; ++n) {
#ifdef ZS_PROBE_NO_GATHER
            const unsigned off = own;
#else
            const unsigned off = o[n];
#endif
            if (HEAD) gv.sv[n] = head_at(off);
            else gv.sv[n] = zs_ld4(r_src, off + 4 * c0);
            if (MODE == ZS_BWD && ACT == ZS_ACT_RELU) gv.nbits[n] = __builtin_amdgcn_raw_buffer_load_b16(r_min, off / (ZS_C * 4 / (ZS_KG * 2)) + 2 * j, 0, 0);
            if (MODE == ZS_BWD && ACT == ZS_ACT_ELU) gv.ov[n] = zs_ld4(r_saved, off + 4 * c0);
        }
#pragma unroll
        for (int i = MODE == ZS_BWD ? 0 : 1; i < 3; ++i) {
            if (HEAD) gv.own[i] = i == 0 ? head_at(own) : make_float4(0.f, 0.f, 0.f, 0.f);
            else gv.own[i] = zs_ld4(r_src, own + 4 * c0 + 4 * ZS_K * i);
            if (MODE == ZS_BWD && ACT == ZS_ACT_ELU) gv.oo[i] = zs_ld4(r_saved, own + 4 * c0 + 4 * ZS_K * i);
        }
        if (MODE == ZS_BWD && ACT == ZS_ACT_RELU) gv.own_bits = __builtin_amdgcn_raw_buffer_load_b16(r_min, own / (ZS_C * 4 / (ZS_KG * 2)) + 2 * j, 0, 0);
    };

    // What a block produced leaves in the NEXT turn, behind that turn's loads (the memory counter is in order: a wait for a
    // load issued behind a store waits for the store as well): the three float4s of the operand row + its sign word out of
    // registers, and the output tile out of LDS in memory order (float4 number tid + 256 v: with ldc == 192 the tile is ONE
    // contiguous 12 KB run; a lane's own 12 columns straight from the accumulators are 16-byte pieces at a 48-byte pitch in
    // 16 rows = 32 partial cache lines per wave instruction, measured 2x the launch).  Rows that do not exist are stored
    // beyond the descriptor's range: dropped.
    float4 px[3];
    unsigned px_bits = 0u, px_off = ZS_OOB;
    auto flush_x = [&]() {
#ifdef ZS_PROBE_NO_STORE
        px_off = ZS_OOB;
#endif
#pragma unroll
        for (int i = 0; i < 3; ++i) zs_st4(r_aout, px_off + 4 * c0 + 4 * ZS_K * i, px[i]);
        if (MODE == ZS_FWD && ACT == ZS_ACT_RELU)
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)px_bits, r_mout, px_off == ZS_OOB ? ZS_OOB : px_off / (ZS_C * 4 / (ZS_KG * 2)) + 2 * j, 0, 0);
        px_off = ZS_OOB;
    };
    auto flush_c = [&](int it) { // the output tile of block `it` (staged by the MFMA waves, behind that block's barrier)
        const int rb = it >= 0 ? p.rb_of(it) : -1;
        const float *tile = lds + 2 * ZS_PANEL + (it & 1) * ZS_CST;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int idx = tid + ZS_ROLE * v, r = idx / 48, c4 = idx % 48, row = rb * 16 + r;
            const f32x4 val = *reinterpret_cast<const f32x4 *>(tile + r * ZS_LDC + 4 * c4);
            unsigned off = rb >= 0 && row < a.rows && 4 * c4 + 3 < a.n_out ? ((unsigned)row * (unsigned)a.ldc + 4u * c4) * 4u : ZS_OOB;
#ifdef ZS_PROBE_NO_STORE
            off = ZS_OOB;
#endif
            zs_st4(r_cout, off, make_float4(val[0], val[1], val[2], val[3]));
        }
    };

    float4 bias4[3], csum[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) bias4[i] = csum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // finish the thread's three float4s of row-block `it`, drop them into the panel and keep them for the flush
    auto finish = [&](ZsGather<MODE, ACT> &gv, int it, float *panel, bool writes) {
        const u32x4 *e = entry(it);
        const u32x4 w0 = e[2], w1 = e[3], misc = e[4];
        const unsigned wv[ZS_W] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const unsigned in_mask = misc.x;
        if (MODE == ZS_BWD) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (ACT == ZS_ACT_RELU) gv.own[i] = zs_masked(gv.own[i], gv.own_bits >> (4 * i));
                else if (ACT == ZS_ACT_ELU) {
                    gv.own[i].x = zs_act_bwd<ACT>(gv.own[i].x, gv.oo[i].x), gv.own[i].y = zs_act_bwd<ACT>(gv.own[i].y, gv.oo[i].y);
                    gv.own[i].z = zs_act_bwd<ACT>(gv.own[i].z, gv.oo[i].z), gv.own[i].w = zs_act_bwd<ACT>(gv.own[i].w, gv.oo[i].w);
                }
            }
        }
        float4 facc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < ZS_W; ++n) {
            // table order == CSR order of the row: the sum zn_aggregate_ell_kernel forms.  A padded slot adds -0.0, which leaves
            // EVERY value untouched, signed zeros included (x + -0.0 == x bit for bit) -- the skipped term of that kernel without
            // a branch (the product is formed and discarded; `v` of a padded slot is the row's own element)
            float4 v = gv.sv[n];
            if (MODE == ZS_BWD && ACT == ZS_ACT_RELU) v = zs_masked(v, gv.nbits[n]);
            else if (MODE == ZS_BWD && ACT == ZS_ACT_ELU) {
                v.x = zs_act_bwd<ACT>(v.x, gv.ov[n].x), v.y = zs_act_bwd<ACT>(v.y, gv.ov[n].y);
                v.z = zs_act_bwd<ACT>(v.z, gv.ov[n].z), v.w = zs_act_bwd<ACT>(v.w, gv.ov[n].w);
            }
            const bool in = (in_mask >> n) & 1u;
            const float wn = __uint_as_float(wv[n]);
            const float tx = wn * v.x, ty = wn * v.y, tz = wn * v.z, tw = wn * v.w;
            facc.x += in ? tx : -0.0f, facc.y += in ? ty : -0.0f, facc.z += in ? tz : -0.0f, facc.w += in ? tw : -0.0f;
        }
        unsigned sign_bits = 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float4 v = i == 0 ? facc : gv.own[i];
            if (MODE == ZS_FWD) {
                v.x += bias4[i].x, v.y += bias4[i].y, v.z += bias4[i].z, v.w += bias4[i].w;
                v.x = zs_act<ACT>(v.x), v.y = zs_act<ACT>(v.y), v.z = zs_act<ACT>(v.z), v.w = zs_act<ACT>(v.w);
                if (ACT == ZS_ACT_RELU) {
                    // out > 0 (the predicate relu' is defined by): a ReLU output is +0 or positive, so its bit pattern is zero
                    // or a positive integer -- min(bits, 1) instead of a compare + select per element
                    sign_bits |= (min(__float_as_uint(v.x), 1u) | min(__float_as_uint(v.y), 1u) << 1 | min(__float_as_uint(v.z), 1u) << 2 |
                                  min(__float_as_uint(v.w), 1u) << 3) << (4 * i);
                }
            }
            const int col = c0 + ZS_K * i; // -> quarter col / 48, position col % 48
            *reinterpret_cast<float4 *>(panel + (col / 48) * ZS_SUB + rl * ZS_LDR + col % 48) = v;
            px[i] = v;
        }
        const bool on = misc.z != 0u && writes;
        px_off = on ? misc.y : ZS_OOB, px_bits = sign_bits;
        if (MODE == ZS_BWD) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                csum[i].x += on ? gv.own[i].x : -0.0f, csum[i].y += on ? gv.own[i].y : -0.0f;
                csum[i].z += on ? gv.own[i].z : -0.0f, csum[i].w += on ? gv.own[i].w : -0.0f;
            }
        }
    };

    // ---- prologue: the first eight row-blocks' entries, then row-blocks 0 and 1 requested, row-block 0 finished -----------
    fill(0);
    if (MODE == ZS_FWD && a.bias) {
#pragma unroll
        for (int i = 0; i < 3; ++i) bias4[i] = *reinterpret_cast<const float4 *>(a.bias + c0 + ZS_K * i);
    }
    __syncthreads(); // the entries are in place (the MFMA waves load their weight slices meanwhile)
    ZsGather<MODE, ACT> gv0, gv1; // used alternately (the loop runs two row-blocks per trip so that they are compile-time names)
    gather(gv0, 0);
    gather(gv1, 1);
    finish(gv0, 0, lds, p.n_main > 0 || writer_extra);
    ZS_STAMP(2);
    __syncthreads();
    ZS_STAMP(3);
    // ---- turn `it` (between the barriers around the MFMA waves' row-block it): request the rows of row-block it + 2, store
    // what block it - 1 produced, finish row-block it + 1 (requested a whole turn ago) into the other panel
    auto turn = [&](int it, ZsGather<MODE, ACT> &fresh, ZsGather<MODE, ACT> &ready, bool writes) {
        ZS_STAMP(4 + 4 * it);
        gather(fresh, it + 2);
        flush_x();
        flush_c(it - 1);
        if (((it + 3) & 7) == 0) fill(it + 3); // visible behind this turn's barrier, first used in the next turn
        ZS_STAMP(5 + 4 * it);
        finish(ready, it + 1, lds + ((it + 1) & 1) * ZS_PANEL, writes);
        ZS_STAMP(6 + 4 * it);
        __syncthreads();
        ZS_STAMP(7 + 4 * it);
    };
    {
        int it = 0;
        for (; it + 1 < p.n_main; it += 2) {
            turn(it, gv0, gv1, true);
            turn(it + 1, gv1, gv0, it + 2 < p.n_main || writer_extra);
        }
        if (it < p.n_main) turn(it, gv0, gv1, writer_extra);
    }
    flush_x();
    flush_c(p.n_main - 1);

    // ---- bias gradient: the workgroup's column sums, rows folded in a fixed order: the four rows of a wave by shuffles
    // (lanes 16 apart), the four waves' sums through LDS by the closing code of zs_body
    if (MODE == ZS_BWD && a.colsum_partial) {
        float *cs = lds + 2 * ZS_PANEL + 2 * ZS_CST; // [4][192]
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float4 t = csum[i];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const int src = ((tid & 63) & 15) + 16 * k;
                t.x += __shfl(csum[i].x, src), t.y += __shfl(csum[i].y, src), t.z += __shfl(csum[i].z, src), t.w += __shfl(csum[i].w, src);
            }
            if (((tid & 63) >> 4) == 0) *reinterpret_cast<float4 *>(cs + (tid >> 6) * ZS_C + c0 + ZS_K * i) = t;
        }
    }
}

template <int MODE, int ACT, bool HEAD>
__device__ __forceinline__ void zs_body(const ZsArgs &a, float *lds, const int w, const int G)
{
    const ZsPlan p = zs_plan(a, w, G);
    if (threadIdx.x < ZS_ROLE) zs_mfma_role(a, lds, p, w);
    else zs_gather_role<MODE, ACT, HEAD>(a, lds, p, w);
    if (MODE == ZS_BWD && a.colsum_partial) {
        __syncthreads();
        const float *cs = lds + 2 * ZS_PANEL + 2 * ZS_CST;
        if (threadIdx.x < ZS_C) {
            const int c = threadIdx.x;
            a.colsum_partial[(size_t)w * ZS_C + c] = ((cs[c] + cs[ZS_C + c]) + cs[2 * ZS_C + c]) + cs[3 * ZS_C + c];
        }
    }
}

template <int MODE, int ACT, bool HEAD>
__global__ __launch_bounds__(ZS_THREADS, 2) void zs_layer_kernel(ZsArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * ZS_PANEL + 2 * ZS_CST + 4 * ZS_C + ZS_RING_DW];
    zs_body<MODE, ACT, HEAD>(a, lds, blockIdx.x, gridDim.x);
}

int zs_cus()
{
    static int cus[64]; // per device: partitions / devices of different sizes get a grid sized for themselves
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        cus[dev] = n > 0 ? n : 256;
    }
    return cus[dev];
}

struct ZsGeo {
    int grid, q, rem, split3;
};
ZsGeo zs_geometry(int rows)
{
    const int nrb = (rows + 15) / 16, cus = zs_cus();
    ZsGeo g;
    if (nrb >= cus) {
        g.grid = cus, g.q = nrb / cus, g.rem = nrb % cus;
    } else {
        g.grid = nrb, g.q = 1, g.rem = 0;
    }
    g.split3 = g.rem > 0 && 3 * g.rem <= g.grid ? 1 : 0;
    return g;
}

inline bool zs_aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int zs_check_common(int b, int nv, int c, int k, int ell_w, int n_out)
{
    if (b < 0 || nv < 0 || c <= 0 || k < 0 || n_out <= 0) return GEOM_EINVAL;
    if (c != ZS_C || k != ZS_K || ell_w != ZS_W || n_out > ZS_C || n_out % 12 != 0) return GEOM_EUNSUPPORTED;
    if ((int64_t)b * nv * ZS_C >= (1LL << 29)) return GEOM_EUNSUPPORTED; // every array < 2 GB: 32-bit byte offsets
    return 0;
}

} // namespace

#ifdef ZS_PROBE_STAMPS
extern "C" int geom_zs_probe_read(unsigned long long *dst, int n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(zs_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

// Workgroups of a geom_zn_layer_*_f32 launch over b * nv rows = rows of the backward's column-sum partials.
extern "C" int64_t geom_zn_layer_partial_rows(int b, int nv)
{
    if (b <= 0 || nv <= 0) return 0;
    return zs_geometry(b * nv).grid;
}

// x_out = act([A . s_prev[:, :k] | s_prev[:, k:]] + bias_prev)  and  s_out = x_out . w   in ONE launch.
//   s_prev [b*nv, 192] raw support of the previous layer, (ell_col, ell_val) [nv][8] its neighbour table (no row longer
//   than the table), bias_prev [192] or NULL, act 0 / 1 (ReLU) / 2 (ELU); w [192, n_out] row-major; x_out [b*nv, 192]
//   (same bits as geom_zn_gcn_aggregate_ell_fwd_f32), relu_mask (ReLU, optional) the sign words in that entry point's
//   layout, s_out [b*nv, n_out]; wt_out (optional) [n_out, 192] = w transposed, for geom_zn_layer_bwd_f32.
// GEOM_EUNSUPPORTED: any other shape (c != 192, k != 64, ell_w != 8, n_out > 192 or % 12) -- the caller then runs
// the aggregation and the product as the two separate entry points.
extern "C" int geom_zn_layer_fwd_f32(int b, int nv, int c, int k, int ell_w, const int *ell_col, const float *ell_val,
                                     const float *s_prev, const float *bias_prev, int act, const float *w, int n_out,
                                     float *x_out, uint16_t *relu_mask, float *s_out, float *wt_out, void *stream)
{
    const int code = zs_check_common(b, nv, c, k, ell_w, n_out);
    if (code) return code;
    if (b == 0 || nv == 0) return 0;
    if (!ell_col || !ell_val || !s_prev || !w || !x_out || !s_out) return GEOM_EINVAL;
    if (!zs_aligned16(ell_col) || !zs_aligned16(ell_val) || !zs_aligned16(s_prev) || !zs_aligned16(x_out) || !zs_aligned16(s_out) ||
        (bias_prev && !zs_aligned16(bias_prev)) || ((uintptr_t)w & 3) || ((uintptr_t)wt_out & 3))
        return GEOM_EINVAL;
    if (relu_mask && act != ZS_ACT_RELU) return GEOM_EINVAL;
    const int rows = b * nv;
    const ZsGeo geo = zs_geometry(rows);
    ZsArgs a{};
    a.src = s_prev, a.bias = bias_prev, a.ell_col = ell_col, a.ell_val = ell_val, a.a_out = x_out, a.mask_out = relu_mask;
    a.bmat = w, a.ldb = n_out, a.c_out = s_out, a.ldc = n_out, a.n_out = n_out, a.bt_out = wt_out;
    a.rows = rows, a.nv = nv, a.q = geo.q, a.rem = geo.rem, a.split3 = geo.split3;
    const dim3 grid(geo.grid), block(ZS_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (act) {
    case ZS_ACT_NONE: hipLaunchKernelGGL((zs_layer_kernel<ZS_FWD, ZS_ACT_NONE, false>), grid, block, 0, s, a); break;
    case ZS_ACT_RELU: hipLaunchKernelGGL((zs_layer_kernel<ZS_FWD, ZS_ACT_RELU, false>), grid, block, 0, s, a); break;
    case ZS_ACT_ELU: hipLaunchKernelGGL((zs_layer_kernel<ZS_FWD, ZS_ACT_ELU, false>), grid, block, 0, s, a); break;
    default: return GEOM_EINVAL;
    }
    return geom::launch_status();
}

// g_out = [A^T . g'[:, :k] | g'[:, k:]] with g' = grad_out * act'(out)  and  grad_in = g_out . w^T  in ONE launch.
//   (ell_col_t, ell_val_t) the table of A^T; grad_out [b*nv, 192] the gradient of the layer's activated output; act' from
//   relu_mask (act 1) / from `out` (act 2) / none (act 0); wt [192, n_in] = the layer's weight TRANSPOSED (what
//   geom_zn_layer_fwd_f32 emits as wt_out); g_out [b*nv, 192] (same bits as geom_zn_gcn_aggregate_ell_bwd_f32), grad_in
//   [b*nv, n_in]; colsum_partial (optional) [geom_zn_layer_partial_rows(b, nv)][192]: per-workgroup column sums of g' (the
//   bias gradient's partials, finished by geom_dense_reduce2_f32 / geom_colsum_batch_f32).
// Head mode (grad_pos != NULL): grad_out is [head_scale * grad_pos | 0 ...] by construction and is not read (see
// geom_zn_gcn_aggregate_ell_head_bwd_f32); act 0 or 1 only.
extern "C" int geom_zn_layer_bwd_f32(int b, int nv, int c, int k, int ell_w, const int *ell_col_t, const float *ell_val_t,
                                     const float *grad_out, const float *out, const uint16_t *relu_mask, int act,
                                     const float *grad_pos, float head_scale, const float *wt, int n_in, float *g_out,
                                     float *grad_in, float *colsum_partial, void *stream)
{
    const int code = zs_check_common(b, nv, c, k, ell_w, n_in);
    if (code) return code;
    if (b == 0 || nv == 0) return 0;
    if (!ell_col_t || !ell_val_t || !wt || !g_out || !grad_in || (!grad_out && !grad_pos)) return GEOM_EINVAL;
    if (act == ZS_ACT_RELU && !relu_mask) return GEOM_EINVAL;
    if (act == ZS_ACT_ELU && (!out || grad_pos)) return grad_pos ? GEOM_EUNSUPPORTED : GEOM_EINVAL;
    if (!zs_aligned16(ell_col_t) || !zs_aligned16(ell_val_t) || !zs_aligned16(grad_out) || !zs_aligned16(out) || !zs_aligned16(g_out) ||
        !zs_aligned16(grad_in) || ((uintptr_t)wt & 3) || ((uintptr_t)grad_pos & 3) || ((uintptr_t)colsum_partial & 3))
        return GEOM_EINVAL;
    const int rows = b * nv;
    const ZsGeo geo = zs_geometry(rows);
    ZsArgs a{};
    a.src = grad_out, a.saved = out, a.mask_in = relu_mask, a.head_gp = grad_pos, a.head_scale = head_scale;
    a.ell_col = ell_col_t, a.ell_val = ell_val_t, a.a_out = g_out, a.colsum_partial = colsum_partial;
    a.bmat = wt, a.ldb = n_in, a.c_out = grad_in, a.ldc = n_in, a.n_out = n_in;
    a.rows = rows, a.nv = nv, a.q = geo.q, a.rem = geo.rem, a.split3 = geo.split3;
    const dim3 grid(geo.grid), block(ZS_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool head = grad_pos != nullptr;
    switch (act) {
    case ZS_ACT_NONE:
        if (head) hipLaunchKernelGGL((zs_layer_kernel<ZS_BWD, ZS_ACT_NONE, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((zs_layer_kernel<ZS_BWD, ZS_ACT_NONE, false>), grid, block, 0, s, a);
        break;
    case ZS_ACT_RELU:
        if (head) hipLaunchKernelGGL((zs_layer_kernel<ZS_BWD, ZS_ACT_RELU, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((zs_layer_kernel<ZS_BWD, ZS_ACT_RELU, false>), grid, block, 0, s, a);
        break;
    case ZS_ACT_ELU: hipLaunchKernelGGL((zs_layer_kernel<ZS_BWD, ZS_ACT_ELU, false>), grid, block, 0, s, a); break;
    default: return GEOM_EINVAL;
    }
    return geom::launch_status();
}
