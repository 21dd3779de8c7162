// A hidden layer of the mesh deformation block in ONE launch per direction (reference models.py:237-297
// with layers.py:107-116): the block follows every 0N-GCN layer with BatchNorm1d(verts) + ReLU, and every second one with the
// residual average `(features + x) / 2`.  As separate operators a hidden layer is three launches each way at the reference's
// training shape (16 meshes x 482 vertices = 7 712 rows, 192 wide): product 9.3 us, aggregation 7.3 us, per-vertex
// BatchNorm 6-7 us (profiles/r05_driver_step_timeline.txt), every one of them a latency chain on a mostly idle chip, with the
// 5.9 MB activation written and re-read between them.
//
//   forward  (geom_deform_layer_fwd_f32), one workgroup per VERTEX v, its B <= 16 batch rows = one 16-row MFMA tile:
//       Z   = [A . S[:, :64] | S[:, 64:]] + bias          the layer's aggregation (zn_gcn.hip's order: same bits)
//       X'  = ReLU(BatchNorm_v(Z))  (+ residual, * scale)  statistics over the vertex's B * 192 values: tile-local
//       S'  = X' . W_next                                  the NEXT layer's product, exact fp32 on v_mfma_f32_16x16x4_f32
//     writes Z (the BatchNorm backward needs it), X' (the next layer's weight gradient and the residuals need it), S'.
//   backward (geom_deform_layer_bwd_f32), same tiling:
//       G   = [A^T . dZ_up[:, :64] | dZ_up[:, 64:]]        aggregation backward of the layer ABOVE (written: its dW needs it)
//       dX  = G . W_up^T  (+ a second upstream gradient)   its input-gradient product
//       dZ  = BatchNorm_v backward (ReLU mask, residual scale) of THIS layer: the two reductions are tile-local again
//     writes G, dZ, the residual's gradient, the BatchNorm parameter gradients and the vertex's bias-gradient column sums.
//
// Why this tiling: BatchNorm1d(verts) normalises a vertex over (batch, channel), so the natural tile is "all batch rows of one
// vertex" -- exactly M = 16 of the fp32 MFMA -- and the only cross-tile dependencies of a layer are the two gathers
// (neighbours' support rows forward, neighbours' dZ rows backward), which is where the launch boundaries sit.  The weight
// operand follows zn_stack.hip: a wave owns 48 output columns and holds its 192 x 48 slice in 144 registers, requested at
// kernel start so that it lands under the gather round trips; the activation tile goes through the same conflict-free
// [k-quarter][row][52] LDS panel, one ds_read_b128 per 12 MFMAs.  482 workgroups of 4 waves, two resident per CU: while one
// waits for its gathers the other runs its 144 MFMAs per wave.
#include "geom_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
struct __attribute__((packed, aligned(4))) f3u { float x, y, z; };

constexpr int DB_THREADS = 256;
constexpr int DB_C = 192;            // layer width
constexpr int DB_K = 64;             // aggregated columns (split 3)
constexpr int DB_W = 8;              // neighbour-table width
constexpr int DB_TAIL = GEOM_DEFORM_TAIL; // width of the tail table (entries of a row beyond the neighbour table)
constexpr int DB_LDR = 52;           // floats per (quarter, row) line of the operand panel: 48 used, pitch 13 x 16 B
constexpr int DB_SUB = 16 * DB_LDR;  // one k-quarter of the panel
constexpr int DB_PANEL = 4 * DB_SUB;
constexpr int DB_LDC = DB_C + 4;     // row pitch of the output staging tile
constexpr int DB_CST = 16 * DB_LDC;
constexpr int DB_RED = 16;           // floats of reduction scratch
constexpr unsigned DB_OOB = 0x80000000u;

#ifdef DB_PROBE_STAMPS
// probe build (tools/probe/db_stamps.sh): shader-clock stamps of wave 0 of every workgroup at the phase boundaries
constexpr int DB_STAMP_SLOTS = 16;
__device__ unsigned long long db_stamps[1024 * DB_STAMP_SLOTS];
#define DB_STAMP(i)                                                                                            \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (threadIdx.x == 0) db_stamps[blockIdx.x * DB_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memtime();     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#else
#define DB_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t db_rsrc(const void *p, int64_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
// AGENT: the access of a chain launch that another workgroup of the SAME launch produces / consumes (sc1: through the XCD's L2
// to the memory side, what an agent-scope atomic compiles to -- the eight L2s do not snoop each other)
constexpr int DB_SC1 = 16;
template <bool AGENT = false>
__device__ __forceinline__ float4 db_ld4(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AGENT ? DB_SC1 : 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <bool AGENT = false>
__device__ __forceinline__ void db_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
                                           r, off, 0, AGENT ? DB_SC1 : 0);
}

// ---- the layers of a block as ONE launch (db_fwd_chain_kernel): a vertex's workgroup runs layer after layer and waits, in front
// of a layer's gathers, until the workgroups of its NEIGHBOURS have published the previous layer's support rows -- the only
// cross-tile dependency of a layer (see the top of the file).  done[v * DB_CTR_STRIDE] = layers vertex v has published (a
// 128-byte line each: pollers and publishers of different vertices never share one); zero on entry (the weight-packing launch of
// the step zeroes it).  Every workgroup of the launch must be resident at once (the host checks; otherwise separate launches).
constexpr int DB_CTR_STRIDE = 32;
constexpr int DB_SPIN_LIMIT = 1 << 20; // polls before a wait gives up (seconds): the layer's outputs are then NaN, loudly

// wave-wide wait: lane's `target` vertex (< 0: none) has published `need` layers.  false: gave up.
__device__ __forceinline__ bool db_wait_published(const int *done, int target, int need)
{
    bool ok = target < 0;
    for (int polls = 0;; ++polls) {
        if (!ok) ok = __hip_atomic_load(done + (size_t)target * DB_CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
        if (__all(ok)) return true;
        if (polls > DB_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

// workgroup w -> vertex: contiguous vertex runs per XCD (workgroup w runs on XCD w % 8; a support row is gathered by its ~7
// neighbours, which are mostly near it in the numbering, so a run's rows stay in one L2)
__device__ __forceinline__ int db_vertex(int w, int vpx, int nv)
{
    const int v = (w & 7) * vpx + (w >> 3);
    return ((w >> 3) < vpx && v < nv) ? v : -1;
}

// fixed-order block reduction of two values: wave shuffles, then the four wave partials in order
__device__ __forceinline__ void db_sum2(float &a, float &b, float *red)
{
    for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) {
        a += __shfl_down(a, off, GEOM_WAVE);
        b += __shfl_down(b, off, GEOM_WAVE);
    }
    const int lane = threadIdx.x & (GEOM_WAVE - 1), wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[2 * wave] = a, red[2 * wave + 1] = b;
    __syncthreads();
    a = ((red[0] + red[2]) + red[4]) + red[6];
    b = ((red[1] + red[3]) + red[5]) + red[7];
}

// The wave's 192 x 48 slice of a weight, from the PACKED copy geom_deform_pack_weights_f32 makes once per step: element
// e = (4 jp + c) * 3 + u of lane (x, g) of wave w is W[k = 48 g + 4 jp + c][48 w + 3 x + u] (zn_stack.hip's operand layout),
// stored [wave][e / 4][lane][e % 4] -- every load instruction of a wave reads 1 KB of consecutive bytes (36 b128 loads per
// lane instead of 48 12-byte loads whose 192-byte runs straddle cache lines: 1.67 x fewer L2 bytes -- all 482 workgroups
// read the same 147 KB, the L2 of an XCD is what they queue at).
struct DbSlice {
    f32x4 q[36];
    __device__ __forceinline__ float at(int jp, int c, int u) const { const int e = (4 * jp + c) * 3 + u; return q[e >> 2][e & 3]; }
};
__device__ __forceinline__ void db_load_slice(DbSlice &bw, const float *packed, int wave, int lane)
{
    const __amdgpu_buffer_rsrc_t r_b = db_rsrc(packed, (int64_t)DB_C * DB_C * 4);
    const unsigned b0 = ((unsigned)(wave * 36) * 64u + (unsigned)lane) * 16u;
#pragma unroll
    for (int i = 0; i < 36; ++i) {
#ifdef DB_PROBE_HOT_SLICE
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r_b, b0 + (unsigned)(i & 1) * 1024u, 0, 0); // probe: 2 KB per wave (L1 hits)
#else
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r_b, b0 + (unsigned)i * 1024u, 0, 0);
#endif
        bw.q[i] = (f32x4){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
    }
}

// C tile [16 rows][192] = panel [16][192] . slice, into the staging tile (natural [row][col] layout, pitch DB_LDC)
__device__ __forceinline__ void db_product(const DbSlice &bw, const float *panel, float *stage, int wave, int x, int g)
{
    const float *pa = panel + g * DB_SUB + x * DB_LDR;
    f32x4 acc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 af = *reinterpret_cast<const f32x4 *>(pa);
#pragma unroll
    for (int jp = 0; jp < 12; ++jp) {
        f32x4 an = af;
        if (jp + 1 < 12) an = *reinterpret_cast<const f32x4 *>(pa + 4 * (jp + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#ifdef DB_PROBE_FEW_MFMA
            if (jp >= 2) continue; // probe: a sixth of the MFMAs (wrong results)
#endif
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.at(jp, c, 0), af[c], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.at(jp, c, 1), af[c], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.at(jp, c, 2), af[c], acc[2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        af = an;
    }
    // accumulator u of lane (x, g) holds C[row x][48 wave + 12 g + 3 r + u], r = 0..3: twelve consecutive columns
    float e[12];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 3; ++u) e[3 * r + u] = acc[u][r];
    float *dst = stage + x * DB_LDC + wave * 48 + 12 * g;
#pragma unroll
    for (int v = 0; v < 3; ++v) *reinterpret_cast<f32x4 *>(dst + 4 * v) = (f32x4){e[4 * v], e[4 * v + 1], e[4 * v + 2], e[4 * v + 3]};
}

// geom_deform_pack_weights_f32: thread = one element of one packed copy (2 * count copies of 36 864 floats)
struct DbPackArgs {
    const float *w[GEOM_DEFORM_MAX_PACK];
    float *fwd, *bwd;
    int count;
    int *zero;      // optional: words to clear (the chain launches' counters), by the workgroups behind the packing ones
    int zero_words, pack_blocks;
};
__global__ __launch_bounds__(256) void db_pack_kernel(DbPackArgs a)
{
    if ((int)blockIdx.x >= a.pack_blocks) {
        const int i = ((int)blockIdx.x - a.pack_blocks) * 256 + (int)threadIdx.x;
        if (i < a.zero_words) a.zero[i] = 0;
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int m = idx / (DB_C * DB_C), r = idx - m * (DB_C * DB_C);
    const int layer = m >> 1, dir = m & 1;
    if (layer >= a.count) return;
    const int wave = r / 9216, rem = r - wave * 9216;
    const int i = rem >> 8, lane = (rem & 255) >> 2, t = rem & 3;
    const int e = 4 * i + t, jp = e / 12, c = (e / 3) & 3, u = e % 3;
    const int g = lane >> 4, x = lane & 15;
    const int k = 48 * g + 4 * jp + c, col = 48 * wave + 3 * x + u;
    const float *w = a.w[layer];
    float *out = dir ? a.bwd : a.fwd;
    if (out) out[(size_t)layer * DB_C * DB_C + r] = dir ? w[col * DB_C + k] : w[k * DB_C + col]; // bwd: the slice of W^T
}

// The aggregated float4 of thread (row rl, group j) of vertex v: sum over the vertex's table entries, then its CSR tail, of
// val * src[mesh rl][neighbour][4 j ..] -- the order and arithmetic of zn_aggregate_ell_kernel (a padded slot adds -0.0: no
// value changes, signed zeros included).  `rowbase` = byte offset of mesh rl's first row; rows beyond the batch pass
// mesh_on = false and read zeros.
// The tail (a vertex with more entries than the table: the 33-entry poles of 482.obj) comes as a second, 32-wide table row
// [nv][DB_TAIL] (-1 = padding) that lanes 0..31 of every wave load with ONE vector load in the round trip of the table's
// scalar loads, so a pole's extra neighbour rows are requested TOGETHER with its table rows: the launch ends with its slowest
// workgroup, and a pole that walked its tail eight entries per dependent round trip took 3 x the time of every other vertex
// (26 000 cycles in the gather phase against 8 500: tools/probe/db_stamps.py).
// a vertex's row of the neighbour table + its tail row (round trip 1 of a layer: scalar loads -- the same for every thread of
// the workgroup -- and ONE vector load of the tail row); a chain launch fetches it once for all its layers
struct DbTable {
    int nb[DB_W];
    float wv[DB_W];
    int tcol;
    float tval;
    bool has_tail;
};
__device__ __forceinline__ DbTable db_table(int v, const int *ell_col, const float *ell_val, const int *tail_col, const float *tail_val,
                                            int lane)
{
    DbTable t;
    t.tcol = -1, t.tval = 0.f;
    if (tail_col) {
        t.tcol = tail_col[(size_t)v * DB_TAIL + (lane & (DB_TAIL - 1))];
        t.tval = tail_val[(size_t)v * DB_TAIL + (lane & (DB_TAIL - 1))];
    }
    const int4 ci0 = *reinterpret_cast<const int4 *>(ell_col + (size_t)v * DB_W), ci1 = *reinterpret_cast<const int4 *>(ell_col + (size_t)v * DB_W + 4);
    const float4 wi0 = *reinterpret_cast<const float4 *>(ell_val + (size_t)v * DB_W), wi1 = *reinterpret_cast<const float4 *>(ell_val + (size_t)v * DB_W + 4);
    t.nb[0] = ci0.x, t.nb[1] = ci0.y, t.nb[2] = ci0.z, t.nb[3] = ci0.w, t.nb[4] = ci1.x, t.nb[5] = ci1.y, t.nb[6] = ci1.z, t.nb[7] = ci1.w;
    t.wv[0] = wi0.x, t.wv[1] = wi0.y, t.wv[2] = wi0.z, t.wv[3] = wi0.w, t.wv[4] = wi1.x, t.wv[5] = wi1.y, t.wv[6] = wi1.z, t.wv[7] = wi1.w;
    t.has_tail = tail_col != nullptr;
    return t;
}

// CHAIN (need > 0): the rows are another workgroup's output of the same launch -- wait until the neighbours have published
// `need` layers (ok = false: the wait gave up), read with agent-scope loads.
template <bool SLICE, bool CHAIN = false>
__device__ __forceinline__ float4 db_aggregate(__amdgpu_buffer_rsrc_t r_src, bool mesh_on, unsigned rowbase, int v, int c0,
                                               const DbTable &tb, float4 *own, DbSlice &bw, const float *packed, int wave, int lane,
                                               const int *done = nullptr, int need = 0, bool *ok = nullptr)
{
    const int (&nb)[DB_W] = tb.nb;
    const float (&wv)[DB_W] = tb.wv;
    const int tcol = tb.tcol;
    const float tval = tb.tval;
    if (CHAIN && need > 0) { // lanes 0..7: the table's neighbours; lanes 8..39: the tail row's
        int target = -1;
#pragma unroll
        for (int n = 0; n < DB_W; ++n) target = lane == n ? nb[n] : target;
        const int from_tail = __shfl(tcol, (lane - DB_W) & (GEOM_WAVE - 1), GEOM_WAVE);
        if (lane >= DB_W && lane < DB_W + DB_TAIL) target = from_tail;
        const bool there = db_wait_published(done, target, need);
        if (!there) *ok = false;
    }
    // round trip 2: the neighbour rows + the thread's own pass-through elements
    float4 sv[DB_W];
#pragma unroll
    for (int n = 0; n < DB_W; ++n) {
        const unsigned off = rowbase + (unsigned)(nb[n] >= 0 ? nb[n] : v) * (DB_C * 4) + 4 * c0;
        sv[n] = db_ld4<CHAIN>(r_src, mesh_on ? off : DB_OOB);
    }
    const unsigned own_off = rowbase + (unsigned)v * (DB_C * 4) + 4 * c0;
#pragma unroll
    for (int i = 0; i < 2; ++i) own[i] = db_ld4<CHAIN>(r_src, mesh_on ? own_off + 4 * DB_K * (i + 1) : DB_OOB);
    float4 facc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto table_terms = [&]() {
#pragma unroll
        for (int n = 0; n < DB_W; ++n) {
            const bool in = nb[n] >= 0;
            const float tx = wv[n] * sv[n].x, ty = wv[n] * sv[n].y, tz = wv[n] * sv[n].z, tw = wv[n] * sv[n].w;
            facc.x += in ? tx : -0.0f, facc.y += in ? ty : -0.0f, facc.z += in ? tz : -0.0f, facc.w += in ? tw : -0.0f;
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    const bool has_tail = tb.has_tail && __builtin_amdgcn_readfirstlane(tcol) >= 0; // (uniform: entry 0 of the tail row)
    if (!has_tail) { // every vertex of an icosphere, all but the two poles of 482.obj
        if (SLICE) db_load_slice(bw, packed, wave, lane); // behind the gathers (in-order memory counter: see the callers)
        __builtin_amdgcn_sched_barrier(0);
        table_terms();
        return facc;
    }
    // the tail rows, all in flight at once; the weight slice behind them (its registers hold the tail's rows until then)
    float4 tv[DB_TAIL];
#pragma unroll
    for (int n = 0; n < DB_TAIL; ++n) {
        const int col = __builtin_amdgcn_readlane(tcol, n);
        tv[n] = db_ld4<CHAIN>(r_src, (mesh_on && col >= 0) ? rowbase + (unsigned)col * (DB_C * 4) + 4 * c0 : DB_OOB);
    }
    table_terms(); // summation order: the table's slots, then the tail, in CSR order
#pragma unroll
    for (int n = 0; n < DB_TAIL; ++n) {
        const bool in = __builtin_amdgcn_readlane(tcol, n) >= 0;
        const float wn = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tval), n));
        const float tx = wn * tv[n].x, ty = wn * tv[n].y, tz = wn * tv[n].z, tw = wn * tv[n].w;
        facc.x += in ? tx : -0.0f, facc.y += in ? ty : -0.0f, facc.z += in ? tz : -0.0f, facc.w += in ? tw : -0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (SLICE) db_load_slice(bw, packed, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    return facc;
}

__device__ __forceinline__ void db_to_panel(float *panel, int rl, int c0, const float4 (&x)[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int col = c0 + DB_K * i;
        *reinterpret_cast<float4 *>(panel + (col / 48) * DB_SUB + rl * DB_LDR + col % 48) = x[i];
    }
}

// One layer of vertex v.  CHAIN: layer `layer` (0-based) of a chain launch -- layer > 0 reads the previous layer's support
// rows from the neighbours' workgroups (wait + agent-scope loads), a layer with a product publishes its rows.
template <bool PRODUCT, bool CHAIN>
__device__ __forceinline__ void db_fwd_body(const geom_deform_fwd &a, const int v, float *lds, int *done, const int layer,
                                            const DbTable *table)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int x = lane & 15, g = lane >> 4; // matrix-core coordinates
    const int rl = tid >> 4, j = tid & 15;  // batch row (mesh) and float4 group of the gather / BatchNorm thread
    const int c0 = 4 * j;
    DB_STAMP(0);
    const bool mesh_on = rl < a.b;
    const int64_t op_bytes = (int64_t)a.b * a.nv * DB_C * 4;
    const __amdgpu_buffer_rsrc_t r_src = db_rsrc(a.s_in, op_bytes);
    const unsigned rowbase = (unsigned)rl * (unsigned)a.nv * (DB_C * 4);
    const unsigned own_off = mesh_on ? rowbase + (unsigned)v * (DB_C * 4) + 4 * c0 : DB_OOB;
    // the vertex's BatchNorm parameters, the bias and the residual travel with the gathers
    const float gamma = a.bn_w ? a.bn_w[v] : 1.f, beta = a.bn_b ? a.bn_b[v] : 0.f;
    const bool updates = a.training && tid == 0;
    const float old_mean = (updates && a.run_mean) ? a.run_mean[v] : 0.f, old_var = (updates && a.run_var) ? a.run_var[v] : 0.f;
    float4 bias4[3], rv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bias4[i] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + c0 + DB_K * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (a.res) {
        const __amdgpu_buffer_rsrc_t r_res = db_rsrc(a.res, ((int64_t)a.b * a.nv - 1) * a.res_ld * 4 + DB_C * 4);
        const unsigned roff = ((unsigned)rl * (unsigned)a.nv + (unsigned)v) * (unsigned)a.res_ld * 4u + 4 * c0;
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = db_ld4(r_res, mesh_on ? roff + 4 * DB_K * i : DB_OOB);
    }
    // The weight slice is requested BEHIND the gathers: the vector-memory counter retires in order, so a wave that asked for
    // its 36 KB of weights first would wait for them in front of every gather (the table entries come by scalar loads, which
    // have a counter of their own); in this order the statistics run while the slice is still on its way.
    DbSlice bw;
    float4 z[3];
    bool arrived = true; // (a chain launch: the neighbours' rows were published in time)
    if (CHAIN) {
        z[0] = db_aggregate<PRODUCT, true>(r_src, mesh_on, rowbase, v, c0, *table, &z[1], bw, a.w_next, wave, lane, done, layer, &arrived);
    } else {
        const DbTable tb = db_table(v, a.ell_col, a.ell_val, a.tail_col, a.tail_val, lane);
        z[0] = db_aggregate<PRODUCT, false>(r_src, mesh_on, rowbase, v, c0, tb, &z[1], bw, a.w_next, wave, lane);
    }
    if (CHAIN && !arrived) z[0].x = __builtin_nanf(""); // poisons the vertex's statistics: every output of the layer is NaN
#pragma unroll
    for (int i = 0; i < 3; ++i) z[i].x += bias4[i].x, z[i].y += bias4[i].y, z[i].z += bias4[i].z, z[i].w += bias4[i].w;

    DB_STAMP(1); // gathers arrived (z holds the aggregated row)
    // ---- BatchNorm1d(verts): one statistic per vertex over its b * 192 values (two-pass: mean, then the centred second moment)
    float *red = lds + DB_PANEL + DB_CST;
    const int n = a.b * DB_C;
    float mean, invstd;
    if (a.training) {
        float s = 0.f, dummy = 0.f;
        if (mesh_on) s = (((z[0].x + z[0].y) + (z[0].z + z[0].w)) + ((z[1].x + z[1].y) + (z[1].z + z[1].w))) + ((z[2].x + z[2].y) + (z[2].z + z[2].w));
        db_sum2(s, dummy, red);
        mean = s / n;
        float q = 0.f;
        if (mesh_on) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float d0 = z[i].x - mean, d1 = z[i].y - mean, d2 = z[i].z - mean, d3 = z[i].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        dummy = 0.f;
        db_sum2(q, dummy, red);
        const float var = q / n; // biased, as used for normalisation
        invstd = 1.f / sqrtf(var + a.eps);
        if (tid == 0) {
            a.save_mean[v] = mean, a.save_invstd[v] = invstd;
            if (a.run_mean) a.run_mean[v] = (1.f - a.momentum) * old_mean + a.momentum * mean;
            if (a.run_var) a.run_var[v] = (1.f - a.momentum) * old_var + a.momentum * (n > 1 ? q / (n - 1) : var);
        }
    } else {
        mean = a.run_mean[v];
        invstd = 1.f / sqrtf(a.run_var[v] + a.eps);
    }
    DB_STAMP(2); // statistics done
    auto finish = [&](float zz, float r) {
        float y = (zz - mean) * invstd * gamma + beta;
        if (a.relu) y = y > 0.f ? y : 0.f;
        if (a.res) y = (r + y) * a.scale;
        return mesh_on ? y : 0.f;
    };
    float4 xo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        xo[i] = make_float4(finish(z[i].x, rv[i].x), finish(z[i].y, rv[i].y), finish(z[i].z, rv[i].z), finish(z[i].w, rv[i].w));
    const __amdgpu_buffer_rsrc_t r_z = db_rsrc(a.z_out, op_bytes), r_x = db_rsrc(a.x_out, op_bytes);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (a.z_out) db_st4(r_z, own_off == DB_OOB ? DB_OOB : own_off + 4 * DB_K * i, z[i]);
        db_st4(r_x, own_off == DB_OOB ? DB_OOB : own_off + 4 * DB_K * i, xo[i]);
    }
    if (!PRODUCT) {
        // ---- the coordinate head's product (models.py:219,295: gc15 = 192 -> 3) inside the last hidden layer's launch:
        // s_head[row][o] = sum_c X[row][c] W_head[c][o]; a row's 192 columns sit in the 16 lanes of its group
        if (a.w_head && a.s_head) {
            float h[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float xv[4] = {xo[i].x, xo[i].y, xo[i].z, xo[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float *wr = a.w_head + (size_t)(c0 + DB_K * i + e) * 3;
                    h[0] += xv[e] * wr[0], h[1] += xv[e] * wr[1], h[2] += xv[e] * wr[2];
                }
            }
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) {
                h[0] += __shfl_xor(h[0], m, GEOM_WAVE), h[1] += __shfl_xor(h[1], m, GEOM_WAVE), h[2] += __shfl_xor(h[2], m, GEOM_WAVE);
            }
            if (j == 0 && mesh_on) {
                float *dst = a.s_head + ((size_t)rl * a.nv + v) * 3;
                dst[0] = h[0], dst[1] = h[1], dst[2] = h[2];
            }
        }
        return;
    }

    // ---- the next layer's product on the tile
    db_to_panel(lds, rl, c0, xo);
    DB_STAMP(3); // outputs requested, panel written
    __syncthreads();
    DB_STAMP(4);
    float *stage = lds + DB_PANEL;
#ifdef DB_PROBE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // probe: separate the wait for the weight slice from the MFMAs
    DB_STAMP(5);
#endif
    db_product(bw, lds, stage, wave, x, g);
    DB_STAMP(6); // MFMAs issued + staged
    __syncthreads();
    DB_STAMP(7);
    const __amdgpu_buffer_rsrc_t r_s = db_rsrc(a.s_out, op_bytes);
#pragma unroll
    for (int t = 0; t < 3; ++t) { // the tile leaves in memory order: 768 contiguous bytes per mesh row
        const int idx = tid + DB_THREADS * t, r = idx / 48, c4 = idx % 48;
        const f32x4 val = *reinterpret_cast<const f32x4 *>(stage + r * DB_LDC + 4 * c4);
        const unsigned off = r < a.b ? ((unsigned)r * (unsigned)a.nv + (unsigned)v) * (DB_C * 4) + 16u * c4 : DB_OOB;
        db_st4<CHAIN>(r_s, off, make_float4(val[0], val[1], val[2], val[3]));
    }
    DB_STAMP(8);
    if (CHAIN) { // publish: every wave's stores are acknowledged (written through), then the vertex's count moves
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done + (size_t)v * DB_CTR_STRIDE, layer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool PRODUCT>
__global__ __launch_bounds__(DB_THREADS, 2) void db_fwd_kernel(geom_deform_fwd a)
{
    __shared__ __attribute__((aligned(16))) float lds[DB_PANEL + DB_CST + DB_RED];
    const int v = db_vertex(blockIdx.x, a.vpx, a.nv);
    if (v < 0) return;
    db_fwd_body<PRODUCT, false>(a, v, lds, nullptr, 0, nullptr);
}

struct DbFwdChain {
    geom_deform_fwd layer[GEOM_DEFORM_CHAIN_MAX];
    int count;
    int *done;
};

__global__ __launch_bounds__(DB_THREADS, 2) void db_fwd_chain_kernel(DbFwdChain c)
{
    __shared__ __attribute__((aligned(16))) float lds[DB_PANEL + DB_CST + DB_RED];
    const int v = db_vertex(blockIdx.x, c.layer[0].vpx, c.layer[0].nv);
    if (v < 0) return;
    const DbTable tb = db_table(v, c.layer[0].ell_col, c.layer[0].ell_val, c.layer[0].tail_col, c.layer[0].tail_val, threadIdx.x & 63);
    for (int l = 0; l < c.count; ++l) {
        if (c.layer[l].w_next) db_fwd_body<true, true>(c.layer[l], v, lds, c.done, l, &tb);
        else db_fwd_body<false, true>(c.layer[l], v, lds, c.done, l, &tb);
        __syncthreads(); // (the layer's LDS is free again)
    }
}

// One backward layer of vertex v.  CHAIN: step `step` of a chain launch (0 = the top layer) -- a step > 0 gathers the dZ rows
// the neighbours' workgroups wrote in the step before (wait + agent-scope loads); every step publishes its dZ rows.
template <bool PRODUCT, bool CHAIN>
__device__ __forceinline__ void db_bwd_body(const geom_deform_bwd &a, const int v, float *lds, int *done, const int step,
                                            const DbTable *table)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int x = lane & 15, g = lane >> 4;
    const int rl = tid >> 4, j = tid & 15;
    const int c0 = 4 * j;
    DB_STAMP(0);
    const bool mesh_on = rl < a.b;
    const int64_t op_bytes = (int64_t)a.b * a.nv * DB_C * 4;
    const unsigned rowbase = (unsigned)rl * (unsigned)a.nv * (DB_C * 4);
    const unsigned own_off = mesh_on ? rowbase + (unsigned)v * (DB_C * 4) + 4 * c0 : DB_OOB;
    auto at = [&](int i) { return own_off == DB_OOB ? DB_OOB : own_off + 4 * DB_K * i; };
    // everything this layer's BatchNorm backward reads is requested with the gathers
    // g / g2 may be column slices of wider row-major buffers (row pitch g_ld / g2_ld floats; dword-aligned 16-byte buffer loads):
    // the next block's input gradient is read in place instead of through a slicing copy
    const int g_ld = a.g_ld ? a.g_ld : DB_C, g2_ld = a.g2_ld ? a.g2_ld : DB_C;
    const __amdgpu_buffer_rsrc_t r_z = db_rsrc(a.z, op_bytes);
    const __amdgpu_buffer_rsrc_t r_g2 = db_rsrc(a.g2, ((int64_t)a.b * a.nv - 1) * g2_ld * 4 + DB_C * 4);
    const __amdgpu_buffer_rsrc_t r_g = db_rsrc(a.g, ((int64_t)a.b * a.nv - 1) * g_ld * 4 + DB_C * 4);
    const unsigned g_off = mesh_on ? (((unsigned)rl * (unsigned)a.nv + (unsigned)v) * (unsigned)g_ld + (unsigned)c0) * 4u : DB_OOB;
    const unsigned g2_off = mesh_on ? (((unsigned)rl * (unsigned)a.nv + (unsigned)v) * (unsigned)g2_ld + (unsigned)c0) * 4u : DB_OOB;
    const float mean = a.save_mean[v], invstd = a.save_invstd[v];
    const float gamma = a.bn_w ? a.bn_w[v] : 1.f, beta = a.bn_b ? a.bn_b[v] : 0.f;
    float4 zv[3], g2v[3], go[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        zv[i] = db_ld4(r_z, at(i));
        g2v[i] = a.g2 ? db_ld4(r_g2, g2_off == DB_OOB ? DB_OOB : g2_off + 4 * DB_K * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!PRODUCT) go[i] = a.g ? db_ld4(r_g, g_off == DB_OOB ? DB_OOB : g_off + 4 * DB_K * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!PRODUCT && a.ds_head) {
        // ---- the coordinate head (gc15, 192 -> 3) inside the first backward launch: its input gradient dS_head . W_head^T is
        // added to g, and the vertex's partial of its weight gradient X^T . dS_head goes out with the column sums
        float dsh[3] = {0.f, 0.f, 0.f};
        if (mesh_on) {
            const float *src = a.ds_head + ((size_t)rl * a.nv + v) * 3;
            dsh[0] = src[0], dsh[1] = src[1], dsh[2] = src[2];
        }
        const __amdgpu_buffer_rsrc_t r_xt = db_rsrc(a.x_top, op_bytes);
        float *wsum = lds + DB_PANEL; // [4 waves][192 * 3]
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 xt = a.dw_head ? db_ld4(r_xt, at(i)) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float xv[4] = {xt.x, xt.y, xt.z, xt.w};
            float add[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = c0 + DB_K * i + e;
                const float *wr = a.w_head + (size_t)col * 3;
                add[e] = (dsh[0] * wr[0] + dsh[1] * wr[1]) + dsh[2] * wr[2];
                if (a.dw_head) { // rows of a wave: lanes 16 apart, in mesh order; the four waves through LDS
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float t = xv[e] * dsh[o];
                        float acc = t;
#pragma unroll
                        for (int k = 1; k < 4; ++k) acc += __shfl(t, (lane & 15) + 16 * k, GEOM_WAVE);
                        if ((lane >> 4) == 0) wsum[wave * (DB_C * 3) + col * 3 + o] = acc;
                    }
                }
            }
            go[i].x += add[0], go[i].y += add[1], go[i].z += add[2], go[i].w += add[3];
        }
        if (a.dw_head) {
            __syncthreads();
            for (int t = tid; t < DB_C * 3; t += DB_THREADS)
                a.dw_head[(size_t)v * (DB_C * 3) + t] = ((wsum[t] + wsum[DB_C * 3 + t]) + wsum[2 * DB_C * 3 + t]) + wsum[3 * DB_C * 3 + t];
            __syncthreads();
        }
    }
    float *stage = lds + DB_PANEL;
    if (PRODUCT) {
        // ---- aggregation backward of the layer above: G = [A^T . dZ_up[:, :64] | dZ_up[:, 64:]]
        const __amdgpu_buffer_rsrc_t r_src = db_rsrc(a.dz_up, op_bytes), r_ds = db_rsrc(a.ds_up, op_bytes);
        float4 gs[3];
        DbSlice bw;
        if (CHAIN) {
            bool arrived = true;
            gs[0] = db_aggregate<true, true>(r_src, mesh_on, rowbase, v, c0, *table, &gs[1], bw, a.wt_up, wave, lane, done, step, &arrived);
            if (!arrived) gs[0].x = __builtin_nanf(""); // (the wait gave up: the layer's gradients are NaN, loudly)
        } else {
            const DbTable tb = db_table(v, a.ell_col_t, a.ell_val_t, a.tail_col_t, a.tail_val_t, lane);
            gs[0] = db_aggregate<true, false>(r_src, mesh_on, rowbase, v, c0, tb, &gs[1], bw, a.wt_up, wave, lane);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) db_st4(r_ds, at(i), gs[i]); // the layer above's weight gradient reads it (X^T . G)
        db_to_panel(lds, rl, c0, gs);                           // (rows beyond the batch read zeros: zero rows of the tile)
        DB_STAMP(1); // gathers arrived, G stored + in the panel
        __syncthreads();
        DB_STAMP(2);
#ifdef DB_PROBE_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DB_STAMP(3);
#endif
        db_product(bw, lds, stage, wave, x, g);                 // dX = G . W_up^T
        DB_STAMP(4);
        __syncthreads();
        DB_STAMP(5);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(stage + rl * DB_LDC + c0 + DB_K * i);
            go[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
    // ---- this layer: residual scale, ReLU mask, BatchNorm backward
    float sum_g = 0.f, sum_gx = 0.f;
    float4 xh[3];
    const __amdgpu_buffer_rsrc_t r_gr = db_rsrc(a.grad_res, op_bytes), r_dz = db_rsrc(a.dz, op_bytes);
    auto one = [&](float zz, float &gg, float second, float &xhat) {
        xhat = (zz - mean) * invstd;
        gg += second;
        if (a.has_res) gg *= a.scale;
        const float pass = gg;
        if (a.relu && !(xhat * gamma + beta > 0.f)) gg = 0.f;
        if (!mesh_on) gg = 0.f;
        sum_g += gg;
        sum_gx += gg * xhat;
        return pass;
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 r = make_float4(one(zv[i].x, go[i].x, g2v[i].x, xh[i].x), one(zv[i].y, go[i].y, g2v[i].y, xh[i].y),
                                     one(zv[i].z, go[i].z, g2v[i].z, xh[i].z), one(zv[i].w, go[i].w, g2v[i].w, xh[i].w));
        if (a.has_res && a.grad_res) db_st4(r_gr, at(i), r);
    }
    float *red = lds + DB_PANEL + DB_CST;
    DB_STAMP(6);
    db_sum2(sum_g, sum_gx, red);
    DB_STAMP(7);
    if (tid == 0) {
        if (a.grad_bn_b) a.grad_bn_b[v] = sum_g;
        if (a.grad_bn_w) a.grad_bn_w[v] = sum_gx;
    }
    const int n = a.b * DB_C;
    const float kk = gamma * invstd, mg = sum_g / n, mgx = sum_gx / n;
    float4 dz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dz[i] = make_float4(kk * (go[i].x - mg - xh[i].x * mgx), kk * (go[i].y - mg - xh[i].y * mgx),
                            kk * (go[i].z - mg - xh[i].z * mgx), kk * (go[i].w - mg - xh[i].w * mgx));
        if (!mesh_on) dz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db_st4<CHAIN>(r_dz, at(i), dz[i]);
    }
    // ---- bias gradient of this layer: the vertex's column sums of dZ over its meshes, in mesh order (lanes 16 apart hold the
    // wave's four meshes, the four waves' sums through LDS); the host adds the vertices up
    if (a.colsum) {
        __syncthreads(); // (the staging tile may still be read above)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float4 t = dz[i];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const int src = (lane & 15) + 16 * k;
                t.x += __shfl(dz[i].x, src), t.y += __shfl(dz[i].y, src), t.z += __shfl(dz[i].z, src), t.w += __shfl(dz[i].w, src);
            }
            if ((lane >> 4) == 0) *reinterpret_cast<float4 *>(stage + wave * DB_C + c0 + DB_K * i) = t;
        }
        __syncthreads();
        if (tid < DB_C) a.colsum[(size_t)v * DB_C + tid] = ((stage[tid] + stage[DB_C + tid]) + stage[2 * DB_C + tid]) + stage[3 * DB_C + tid];
    }
    DB_STAMP(8);
    if (CHAIN) { // publish the vertex's dZ rows (every wave's stores acknowledged first)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done + (size_t)v * DB_CTR_STRIDE, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool PRODUCT>
__global__ __launch_bounds__(DB_THREADS, 2) void db_bwd_kernel(geom_deform_bwd a)
{
    __shared__ __attribute__((aligned(16))) float lds[DB_PANEL + DB_CST + DB_RED];
    const int v = db_vertex(blockIdx.x, a.vpx, a.nv);
    if (v < 0) return;
    db_bwd_body<PRODUCT, false>(a, v, lds, nullptr, 0, nullptr);
}

struct DbBwdChain {
    geom_deform_bwd layer[GEOM_DEFORM_CHAIN_MAX]; // in execution order: the top layer first
    int count;
    int *done;
    float *ds_first; // optional: [A^T . dZ[:, :64] | dZ[:, 64:]] of the LAST step's dZ (the first layer's support gradient)
};

__global__ __launch_bounds__(DB_THREADS, 2) void db_bwd_chain_kernel(DbBwdChain c)
{
    __shared__ __attribute__((aligned(16))) float lds[DB_PANEL + DB_CST + DB_RED];
    const int v = db_vertex(blockIdx.x, c.layer[0].vpx, c.layer[0].nv);
    if (v < 0) return;
    const geom_deform_bwd &t = c.layer[c.count > 1 ? 1 : 0]; // (the top layer may come without tables)
    const DbTable tb = db_table(v, t.ell_col_t, t.ell_val_t, t.tail_col_t, t.tail_val_t, threadIdx.x & 63);
    for (int l = 0; l < c.count; ++l) {
        if (c.layer[l].dz_up) db_bwd_body<true, true>(c.layer[l], v, lds, c.done, l, &tb);
        else db_bwd_body<false, true>(c.layer[l], v, lds, c.done, l, &tb);
        __syncthreads();
    }
    if (c.ds_first) {
        // one more hand-off instead of one more launch: the aggregation backward of the chain's first layer (no activation, no
        // product behind it: the first layer's products stay with the caller) -- the bits of geom_zn_gcn_aggregate_ell_bwd_f32
        const geom_deform_bwd &a = c.layer[c.count - 1];
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
        const int rl = tid >> 4, c0 = 4 * (tid & 15);
        const bool mesh_on = rl < a.b;
        const int64_t op_bytes = (int64_t)a.b * a.nv * DB_C * 4;
        const unsigned rowbase = (unsigned)rl * (unsigned)a.nv * (DB_C * 4);
        const __amdgpu_buffer_rsrc_t r_src = db_rsrc(a.dz, op_bytes), r_ds = db_rsrc(c.ds_first, op_bytes);
        float4 gs[3];
        DbSlice none;
        bool arrived = true;
        gs[0] = db_aggregate<false, true>(r_src, mesh_on, rowbase, v, c0, tb, &gs[1], none, nullptr, wave, lane, c.done, c.count, &arrived);
        if (!arrived) gs[0].x = __builtin_nanf("");
        const unsigned own_off = mesh_on ? rowbase + (unsigned)v * (DB_C * 4) + 4 * c0 : DB_OOB;
#pragma unroll
        for (int i = 0; i < 3; ++i) db_st4(r_ds, own_off == DB_OOB ? DB_OOB : own_off + 4 * DB_K * i, gs[i]);
    }
}

inline bool db_aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int db_check_shape(int b, int nv, int c, int k, int ell_w)
{
    if (b < 0 || nv < 0 || c <= 0 || k < 0) return GEOM_EINVAL;
    if (c != DB_C || k != DB_K || ell_w != DB_W || b > 16) return GEOM_EUNSUPPORTED;
    if ((int64_t)b * nv * DB_C >= (1LL << 29)) return GEOM_EUNSUPPORTED; // 32-bit byte offsets
    return 0;
}

} // namespace

#ifdef DB_PROBE_STAMPS
extern "C" int geom_db_probe_read(unsigned long long *dst, int n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(db_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int geom_deform_layer_fwd_f32(const geom_deform_fwd *args, void *stream)
{
    if (!args) return GEOM_EINVAL;
    geom_deform_fwd a = *args;
    const int code = db_check_shape(a.b, a.nv, a.c, a.k, a.ell_w);
    if (code) return code;
    if (a.b == 0 || a.nv == 0) return 0;
    if (!a.s_in || !a.ell_col || !a.ell_val || !a.x_out) return GEOM_EINVAL;
    if (a.training ? (!a.save_mean || !a.save_invstd) : (!a.run_mean || !a.run_var)) return GEOM_EINVAL;
    if (a.w_next && !a.s_out) return GEOM_EINVAL;
    if ((a.w_head != nullptr) != (a.s_head != nullptr) || (a.w_head && a.w_next)) return GEOM_EINVAL; // the head rides in the launch without a product
    if (a.tail_col && !a.tail_val) return GEOM_EINVAL;
    if (a.res && a.res_ld < DB_C) return GEOM_EINVAL; // (any pitch: a column slice of the block's 1155-wide input is read in place)
    if (!db_aligned16(a.s_in) || !db_aligned16(a.ell_col) || !db_aligned16(a.ell_val) || !db_aligned16(a.x_out) || !db_aligned16(a.z_out) ||
        !db_aligned16(a.s_out) || !db_aligned16(a.bias) || ((uintptr_t)a.res & 3) || !db_aligned16(a.w_next))
        return GEOM_EINVAL;
    if (!a.res) a.scale = 1.f;
    a.vpx = (a.nv + 7) / 8;
    const dim3 grid(8 * a.vpx), block(DB_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.w_next) hipLaunchKernelGGL((db_fwd_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((db_fwd_kernel<false>), grid, block, 0, s, a);
    return geom::launch_status();
}

extern "C" int geom_deform_layer_bwd_f32(const geom_deform_bwd *args, void *stream)
{
    if (!args) return GEOM_EINVAL;
    geom_deform_bwd a = *args;
    const int code = db_check_shape(a.b, a.nv, a.c, a.k, a.ell_w);
    if (code) return code;
    if (a.b == 0 || a.nv == 0) return 0;
    if (!a.z || !a.save_mean || !a.save_invstd || !a.dz) return GEOM_EINVAL;
    const bool product = a.dz_up != nullptr;
    if (product ? (!a.ell_col_t || !a.ell_val_t || !a.ds_up || !a.wt_up) : (!a.g && !a.ds_head)) return GEOM_EINVAL;
    if (a.ds_head && (product || !a.w_head || (a.dw_head && !a.x_top) || !db_aligned16(a.x_top))) return GEOM_EINVAL;
    if (a.tail_col_t && !a.tail_val_t) return GEOM_EINVAL;
    if ((a.g_ld && a.g_ld < DB_C) || (a.g2_ld && a.g2_ld < DB_C)) return GEOM_EINVAL;
    if ((int64_t)a.b * a.nv * (a.g_ld > a.g2_ld ? a.g_ld : a.g2_ld) >= (1LL << 29)) return GEOM_EUNSUPPORTED;
    if (!db_aligned16(a.dz_up) || !db_aligned16(a.ell_col_t) || !db_aligned16(a.ell_val_t) || !db_aligned16(a.ds_up) || ((uintptr_t)a.g & 3) ||
        ((uintptr_t)a.g2 & 3) || !db_aligned16(a.z) || !db_aligned16(a.grad_res) || !db_aligned16(a.dz) || !db_aligned16(a.colsum) ||
        !db_aligned16(a.wt_up))
        return GEOM_EINVAL;
    if (!a.has_res) a.scale = 1.f;
    a.vpx = (a.nv + 7) / 8;
    const dim3 grid(8 * a.vpx), block(DB_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (product) hipLaunchKernelGGL((db_bwd_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((db_bwd_kernel<false>), grid, block, 0, s, a);
    return geom::launch_status();
}

// `count` backward layers in execution order (layers[0] = the top layer, read from memory: dz_up == NULL; layers[t].dz_up ==
// layers[t - 1].dz for t >= 1) in ONE launch; results = those of the separate geom_deform_layer_bwd_f32 calls, bit for bit.
// done: as for geom_deform_chain_fwd_f32 (its own nv * 32 zeroed ints).  ds_first (may be NULL): also the aggregation backward of
// the last step's dZ, [A^T . dZ[:, :64] | dZ[:, 64:]] -> ds_first [b,nv,192] (the support gradient of the chain's first layer;
// the bits of geom_zn_gcn_aggregate_ell_bwd_f32 without activation), as a last hand-off of the same launch.
extern "C" int geom_deform_chain_bwd_f32(int count, const geom_deform_bwd *layers, int *done, float *ds_first, void *stream)
{
    if (count <= 0 || count > GEOM_DEFORM_CHAIN_MAX || !layers || !done || ((uintptr_t)done & 127)) return GEOM_EINVAL;
    if (ds_first && (count < 2 || !db_aligned16(ds_first))) return GEOM_EINVAL; // (the transposed tables come with layers[1])
    DbBwdChain c{};
    for (int l = 0; l < count; ++l) {
        geom_deform_bwd a = layers[l];
        const int code = db_check_shape(a.b, a.nv, a.c, a.k, a.ell_w);
        if (code) return code;
        if (a.b != layers[0].b || a.nv != layers[0].nv) return GEOM_EINVAL;
        if (!a.z || !a.save_mean || !a.save_invstd || !a.dz) return GEOM_EINVAL;
        const bool product = a.dz_up != nullptr;
        if (product != (l > 0)) return GEOM_EINVAL;                                   // only the top layer reads its gradient from memory
        if (product && a.dz_up != layers[l - 1].dz) return GEOM_EINVAL;               // a chain
        if (product && l > 1 && (a.ell_col_t != layers[1].ell_col_t || a.ell_val_t != layers[1].ell_val_t ||
                                 a.tail_col_t != layers[1].tail_col_t || a.tail_val_t != layers[1].tail_val_t))
            return GEOM_EINVAL;
        if (product ? (!a.ell_col_t || !a.ell_val_t || !a.ds_up || !a.wt_up) : (!a.g && !a.ds_head)) return GEOM_EINVAL;
        if (a.ds_head && (product || !a.w_head || (a.dw_head && !a.x_top) || !db_aligned16(a.x_top))) return GEOM_EINVAL;
        if (a.tail_col_t && !a.tail_val_t) return GEOM_EINVAL;
        if ((a.g_ld && a.g_ld < DB_C) || (a.g2_ld && a.g2_ld < DB_C)) return GEOM_EINVAL;
        if ((int64_t)a.b * a.nv * (a.g_ld > a.g2_ld ? a.g_ld : a.g2_ld) >= (1LL << 29)) return GEOM_EUNSUPPORTED;
        if (!db_aligned16(a.dz_up) || !db_aligned16(a.ell_col_t) || !db_aligned16(a.ell_val_t) || !db_aligned16(a.ds_up) || ((uintptr_t)a.g & 3) ||
            ((uintptr_t)a.g2 & 3) || !db_aligned16(a.z) || !db_aligned16(a.grad_res) || !db_aligned16(a.dz) || !db_aligned16(a.colsum) ||
            !db_aligned16(a.wt_up))
            return GEOM_EINVAL;
        for (int e = 0; e < l; ++e)
            if (layers[e].dz == a.dz) return GEOM_EINVAL;                              // (every step its own dZ: neighbours read it a step later)
        if (!a.has_res) a.scale = 1.f;
        a.vpx = (a.nv + 7) / 8;
        c.layer[l] = a;
    }
    if (layers[0].b == 0 || layers[0].nv == 0) return 0;
    if (!geom_deform_chain_fits(layers[0].nv)) return GEOM_EUNSUPPORTED;
    c.count = count, c.done = done, c.ds_first = ds_first;
    hipLaunchKernelGGL(db_bwd_chain_kernel, dim3(8 * c.layer[0].vpx), dim3(DB_THREADS), 0, static_cast<hipStream_t>(stream), c);
    return geom::launch_status();
}

// fwd[l] / bwd[l] (each count x 36 864 floats, either may be NULL) = the register-slice order of w[l] / w[l]^T that
// geom_deform_layer_fwd_f32 (w_next) / geom_deform_layer_bwd_f32 (wt_up) read; w = HOST array of count <= GEOM_DEFORM_MAX_PACK
// device pointers to [192,192] row-major matrices.  One launch for all layers of a block, once per step.
extern "C" int geom_deform_pack_weights_f32(int count, const float *const *w, float *fwd, float *bwd, void *stream)
{
    return geom_deform_pack_weights_zero_f32(count, w, fwd, bwd, nullptr, 0, stream);
}

// ... and `zero_words` 4-byte words at `zero` cleared by the same launch (the counters of the step's chain launches)
extern "C" int geom_deform_pack_weights_zero_f32(int count, const float *const *w, float *fwd, float *bwd, int *zero, int zero_words,
                                                 void *stream)
{
    if (count < 0 || count > GEOM_DEFORM_MAX_PACK || zero_words < 0 || (zero_words > 0 && !zero)) return GEOM_EINVAL;
    if (count == 0 || (!fwd && !bwd)) count = 0;
    if (count == 0 && zero_words == 0) return 0;
    if (count && (!w || !db_aligned16(fwd) || !db_aligned16(bwd))) return GEOM_EINVAL;
    DbPackArgs a{};
    for (int i = 0; i < count; ++i) {
        if (!w[i] || ((uintptr_t)w[i] & 3)) return GEOM_EINVAL;
        a.w[i] = w[i];
    }
    a.fwd = fwd, a.bwd = bwd, a.count = count;
    const int total = count * 2 * DB_C * DB_C;
    a.pack_blocks = (total + 255) / 256, a.zero = zero, a.zero_words = zero_words;
    hipLaunchKernelGGL(db_pack_kernel, dim3(a.pack_blocks + (zero_words + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return geom::launch_status();
}

// Whether a chain launch over nv vertices can run on the current device: every workgroup must be resident at once (a
// workgroup waits for its neighbours' INSIDE the launch).
extern "C" int geom_deform_chain_fits(int nv)
{
    if (nv <= 0) return 0;
    static int slots[64] = {0}; // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!slots[dev]) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        int per_cu_b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, db_fwd_chain_kernel, DB_THREADS, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, db_bwd_chain_kernel, DB_THREADS, 0) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return 0;
        per_cu = per_cu < per_cu_b ? per_cu : per_cu_b;
        slots[dev] = per_cu * prop.multiProcessorCount > 0 ? per_cu * prop.multiProcessorCount : -1;
    }
    return slots[dev] >= 8 * ((nv + 7) / 8);
}

// `count` consecutive layers (layers[l + 1].s_in == layers[l].s_out, all but possibly the last with a product) in ONE launch;
// results = those of `count` geom_deform_layer_fwd_f32 calls, bit for bit.  done: nv * 32 ints, ZERO on entry
// (geom_deform_pack_weights_zero_f32 of the same step), 128-byte aligned.  GEOM_EUNSUPPORTED when the launch does not fit the
// device (geom_deform_chain_fits) -- the caller issues the layers one by one.
extern "C" int geom_deform_chain_fwd_f32(int count, const geom_deform_fwd *layers, int *done, void *stream)
{
    if (count <= 0 || count > GEOM_DEFORM_CHAIN_MAX || !layers || !done || ((uintptr_t)done & 127)) return GEOM_EINVAL;
    DbFwdChain c{};
    for (int l = 0; l < count; ++l) {
        geom_deform_fwd a = layers[l];
        const int code = db_check_shape(a.b, a.nv, a.c, a.k, a.ell_w);
        if (code) return code;
        if (a.b != layers[0].b || a.nv != layers[0].nv || a.ell_col != layers[0].ell_col || a.ell_val != layers[0].ell_val ||
            a.tail_col != layers[0].tail_col || a.tail_val != layers[0].tail_val)
            return GEOM_EINVAL;
        if (!a.s_in || !a.ell_col || !a.ell_val || !a.x_out) return GEOM_EINVAL;
        if (a.training ? (!a.save_mean || !a.save_invstd) : (!a.run_mean || !a.run_var)) return GEOM_EINVAL;
        if (a.w_next ? !a.s_out : l + 1 < count) return GEOM_EINVAL;                 // only the last layer may lack a product
        if (l > 0 && a.s_in != layers[l - 1].s_out) return GEOM_EINVAL;              // a chain
        if (l > 1 && a.s_out && a.s_out == layers[l - 1].s_out) return GEOM_EINVAL; // (ping-pong at least)
        if ((a.w_head != nullptr) != (a.s_head != nullptr) || (a.w_head && a.w_next)) return GEOM_EINVAL;
        if (a.tail_col && !a.tail_val) return GEOM_EINVAL;
        if (a.res && a.res_ld < DB_C) return GEOM_EINVAL;
        if (!db_aligned16(a.s_in) || !db_aligned16(a.ell_col) || !db_aligned16(a.ell_val) || !db_aligned16(a.x_out) || !db_aligned16(a.z_out) ||
            !db_aligned16(a.s_out) || !db_aligned16(a.bias) || ((uintptr_t)a.res & 3) || !db_aligned16(a.w_next))
            return GEOM_EINVAL;
        if (!a.res) a.scale = 1.f;
        a.vpx = (a.nv + 7) / 8;
        c.layer[l] = a;
    }
    if (layers[0].b == 0 || layers[0].nv == 0) return 0;
    if (!geom_deform_chain_fits(layers[0].nv)) return GEOM_EUNSUPPORTED;
    c.count = count, c.done = done;
    hipLaunchKernelGGL(db_fwd_chain_kernel, dim3(8 * c.layer[0].vpx), dim3(DB_THREADS), 0, static_cast<hipStream_t>(stream), c);
    return geom::launch_status();
}
