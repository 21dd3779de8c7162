// geom_gemm_f32: C[M][N] = op(A) . op(B) in exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32), ANY sizes, leading
// dimensions and alignments -- the products of the layers whose widths the tuned kernels of dense_gemm.hip do not take
// (they are written for 192 output columns): the sixteen ZERON_GCN layers of the mesh encoder (reference models.py:299-348:
// widths 3, 60, 120, 150, 200, 210, 250, 300) and its GCNMax head (300 -> latent), forward `torch.mm(input, weight)`
// (reference layers.py:30, 66) and the two gradients autograd derives from it.  The library serves those shapes badly: the
// weight gradients (18 432 summed rows against a 300 x 300 output) run as ~100 workgroups without a K split, 73-97 us each
// -- 1.25 of the 3.2 ms of an encoder step (tools/probe/encoder_profile.sh).
//
// One kernel, three operand forms.  A panel of an operand whose SUMMED index is contiguous in memory ("k-contiguous": the
// rows of X in X . W, the rows of G and of W in G . W^T) is kept in LDS as [row][36] and a lane reads FOUR consecutive k of
// its row with one ds_read_b128; a panel whose summed index is the slow one ("k-major": W in X . W, both operands of
// X^T . G) is kept as [k][68] and read one float per MFMA.  The MFMA k-steps of a 32-deep stage are permuted so that both
// forms agree: step s = 4 q + j gives lane group g the summed index 16 q + 4 g + j.  Bank arithmetic: [row][36] -- the 8
// lanes a b128 read serves per cycle sit 36 floats apart = 4 banks: 32 distinct banks; [k][68] -- lane groups g, g + 1 sit
// 4 * 68 floats = 16 banks apart, 16 lanes each: 32 distinct banks.
// Workgroup = 64 x 64 output tile, four waves 2 x 2, a wave 32 x 32 = 2 x 2 MFMA tiles; global -> registers -> LDS, double
// buffered, one barrier per stage; loads are buffer loads whose out-of-range lanes get an offset beyond the resource (zero
// fill, no branches) and whose vector width follows the divisibility of the contiguous extent (16 / 8 / 4 bytes: 963- and
// 3-float rows take the narrow ones).  The accumulators are taken with the operands swapped (D^T = B^T A^T) so that a lane
// holds four consecutive COLUMNS of one row.  A long sum against few tiles (the weight gradients) is split over the summed
// index into `splits` partial tiles in a caller-provided workspace, added up in split order by a second launch:
// bit-reproducible.  Small tiles on purpose: the shapes are ragged (N = 300 is 4.7 tiles), and 5 x 288 = 1440 workgroups
// balance over the chip where 3 x 144 do not.
#include "geom_common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int GA_THREADS = 256;
constexpr int GA_T = 64;   // tile edge of the base tile (rows and columns); the tall tile is 128 x 64
constexpr int GA_BK = 32;  // granularity of the split chunks (a multiple of both stage depths)
constexpr int ga_pk(int bk) { return bk + 4; }     // [row][k] panel pitch (36 / 20 floats: 8 lanes x 4 banks apart)
constexpr int ga_pm(int rows) { return rows + 4; } // [k][row] panel pitch (68 / 132: 4 mod 8 -- see the bank arithmetic above)
constexpr unsigned GA_OOB = 0x80000000u;
constexpr int ga_panel(int rows, int bk) { return rows * ga_pk(bk) > bk * ga_pm(rows) ? rows * ga_pk(bk) : bk * ga_pm(rows); } // floats per buffer

struct AnyArgs {
    const float *a, *b;
    float *c;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    int splits, k_chunk;   // split s sums k in [s * k_chunk, min(K, (s + 1) * k_chunk)) into c + s * split_stride
    int64_t split_stride;
    int64_t a_bytes, b_bytes, c_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ga_rsrc(const void *p, int64_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// One operand's ROWS x 32 panel of a stage: V floats per load, PASSES loads per thread.
// KM = false: the source is [x][k] (k contiguous), LDS [x][36];  KM = true: the source is [k][x] (x contiguous), LDS [k][ROWS + 4].
template <bool KM, int V, int ROWS, int BK>
struct Panel {
    static constexpr int PASSES = ROWS * BK / V / GA_THREADS;
    static_assert(PASSES >= 1, "a stage must give every thread a load");
    unsigned off[PASSES];   // byte offset of the load at stage 0 (GA_OOB: the row / column is outside the operand)
    unsigned lds[PASSES];   // float offset inside a panel buffer
    int kl[PASSES];         // the load's first summed index inside a stage
    unsigned step;          // bytes per stage
    unsigned v[2][PASSES][V]; // two register sets: a stage's loads are requested TWO stages ahead

    __device__ __forceinline__ void prepare(int x0, int X, int64_t ld, int k_begin)
    {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int idx = tid + GA_THREADS * p;
            int x, k;
            if (KM) {
                constexpr int PER = ROWS / V; // loads per k row
                k = idx / PER, x = (idx % PER) * V;
                lds[p] = (unsigned)(k * ga_pm(ROWS) + x);
            } else {
                constexpr int PER = BK / V; // loads per x row
                x = idx / PER, k = (idx % PER) * V;
                lds[p] = (unsigned)(x * ga_pk(BK) + k);
            }
            kl[p] = k;
            const bool in = x0 + x < X; // (X % V == 0 for the k-major form: a vector is inside or outside as a whole)
            const int64_t e = KM ? (int64_t)(k_begin + k) * ld + x0 + x : (int64_t)(x0 + x) * ld + k_begin + k;
            off[p] = in ? (unsigned)(e * 4) : GA_OOB;
        }
        step = (unsigned)((KM ? (int64_t)BK * ld : (int64_t)BK) * 4);
    }

    // k_left = summed indices from this stage's first one to the end of the split's range
    template <int SET>
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, int stage, int k_left)
    {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
#ifdef GA_PROBE_NO_GLOBAL // tools/probe: no operand loads (the registers keep whatever they hold)
            if (stage > 1) continue;
#endif
            const unsigned o = (kl[p] < k_left && off[p] != GA_OOB) ? off[p] + (unsigned)stage * step : GA_OOB;
            if constexpr (V == 4) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0);
                v[SET][p][0] = t.x, v[SET][p][1] = t.y, v[SET][p][2] = t.z, v[SET][p][3] = t.w;
            } else if constexpr (V == 2) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0);
                v[SET][p][0] = t.x, v[SET][p][1] = t.y;
            } else {
                v[SET][p][0] = __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0);
            }
        }
    }

    template <int SET>
    __device__ __forceinline__ void store(float *panel) const
    {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            unsigned *d = reinterpret_cast<unsigned *>(panel) + lds[p];
            if constexpr (V == 4) *reinterpret_cast<u32x4 *>(d) = (u32x4){v[SET][p][0], v[SET][p][1], v[SET][p][2], v[SET][p][3]};
            else if constexpr (V == 2) *reinterpret_cast<u32x2 *>(d) = (u32x2){v[SET][p][0], v[SET][p][1]};
            else *d = v[SET][p][0];
        }
    }
};

// A_KM: A is stored [K][M] (the transposed operand of X^T . G); B_KN: B is stored [K][N] (W in X . W, G in X^T . G), else [N][K].
// WM: MFMA row-blocks per wave -- 2: the 64 x 64 tile that ships (4 = a tall 128 x 64 tile: measured and rejected, any_plan).
template <bool A_KM, bool B_KN, int VA, int VB, int VC, int WM, int BK>
__global__ __launch_bounds__(GA_THREADS) void any_gemm_kernel(AnyArgs q)
{
    constexpr int TM = 32 * WM, TN = GA_T, PK = ga_pk(BK);
    constexpr int PA_F = ga_panel(TM, BK), PB_F = ga_panel(TN, BK);
    constexpr int PMA = ga_pm(TM), PMB = ga_pm(TN);
    __shared__ __attribute__((aligned(16))) float lds[2 * PA_F + 2 * PB_F];
    // block -> (tile row, split) on "its" XCD, tile column fastest: the workgroups that share an A panel run next to each
    // other on one XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % q.tiles_n, u = (slot / q.tiles_n) * 8 + xcd;
    if (u >= q.tiles_m * q.splits) return;
    const int mt = u % q.tiles_m, split = u / q.tiles_m;
    const int m0 = mt * TM, n0 = nt * TN;
    const int k_begin = split * q.k_chunk, k_end = min(q.K, k_begin + q.k_chunk);
    const int nst = (k_end - k_begin + BK - 1) / BK;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int wr = 16 * WM * wm, wc = 32 * wn; // the wave's first row / column inside the tile

    f32x4 acc[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t ra = ga_rsrc(q.a, q.a_bytes), rb = ga_rsrc(q.b, q.b_bytes);
    Panel<A_KM, VA, TM, BK> pa;
    Panel<B_KN, VB, TN, BK> pb;
    pa.prepare(m0, q.M, q.lda, k_begin);
    pb.prepare(n0, q.N, q.ldb, k_begin);
    float *const la = lds, *const lb = lds + 2 * PA_F;

    // Loads run TWO stages ahead of the MFMAs that consume them (two register sets, two LDS buffers): one stage of MFMAs is
    // 0.4 us, a round trip to L2 / HBM under load more -- with one stage of lead every stage ended waiting for its successor.
    const int k_len = k_end - k_begin;
    auto compute = [&](const float *as, const float *bs) {
#pragma unroll
        for (int qq = 0; qq < BK / 16; ++qq) {
            f32x4 a4[WM], b4[2];
#ifdef GA_PROBE_NO_FRAG // tools/probe: the fragments of the first quarter-stage feed every MFMA (no further LDS reads)
            if (qq > 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[i][j ^ 1][0], acc[i ^ 1][j][1], acc[i][j], 0, 0, 0);
                continue;
            }
#endif
            if constexpr (!A_KM) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a4[i] = *reinterpret_cast<const f32x4 *>(as + (wr + 16 * i + x) * PK + 16 * qq + 4 * g);
            }
            if constexpr (!B_KN) {
#pragma unroll
                for (int j = 0; j < 2; ++j) b4[j] = *reinterpret_cast<const f32x4 *>(bs + (wc + 16 * j + x) * PK + 16 * qq + 4 * g);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float af[WM], bf[2];
                const int kk = 16 * qq + 4 * g + s;
#pragma unroll
                for (int i = 0; i < WM; ++i) af[i] = A_KM ? as[kk * PMA + wr + 16 * i + x] : a4[i][s];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = B_KN ? bs[kk * PMB + wc + 16 * j + x] : b4[j][s];
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
        }
    };
    if (nst > 0) {
        pa.template issue<0>(ra, 0, k_len);
        pb.template issue<0>(rb, 0, k_len);
        if (nst > 1) {
            pa.template issue<1>(ra, 1, k_len - BK);
            pb.template issue<1>(rb, 1, k_len - BK);
        }
        pa.template store<0>(la);
        pb.template store<0>(lb);
        __syncthreads();
    }
    for (int st = 0; st < nst; st += 2) {
        { // stage st: LDS buffer 0; its successor's registers are set 1
            if (st + 2 < nst) {
                pa.template issue<0>(ra, st + 2, k_len - (st + 2) * BK);
                pb.template issue<0>(rb, st + 2, k_len - (st + 2) * BK);
            }
            compute(la, lb);
            if (st + 1 < nst) {
                pa.template store<1>(la + PA_F);
                pb.template store<1>(lb + PB_F);
            }
            __syncthreads();
        }
        if (st + 1 < nst) { // stage st + 1: LDS buffer 1; its successor's registers are set 0
            if (st + 3 < nst) {
                pa.template issue<1>(ra, st + 3, k_len - (st + 3) * BK);
                pb.template issue<1>(rb, st + 3, k_len - (st + 3) * BK);
            }
            compute(la + PA_F, lb + PB_F);
            if (st + 2 < nst) {
                pa.template store<0>(la);
                pb.template store<0>(lb);
            }
            __syncthreads();
        }
    }

    // lane (x, g) holds C[m0 + wr + 16 i + x][n0 + wc + 16 j + 4 g .. + 3]
    float *cbase = q.c + (int64_t)split * q.split_stride;
    const __amdgpu_buffer_rsrc_t rc = ga_rsrc(cbase, q.c_bytes);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = m0 + wr + 16 * i + x;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wc + 16 * j + 4 * g;
            const f32x4 v = acc[i][j];
            const unsigned o = (unsigned)(((int64_t)m * q.ldc + n) * 4);
            if constexpr (VC == 4) {
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                       rc, (m < q.M && n < q.N) ? o : GA_OOB, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rc, (m < q.M && n + r < q.N) ? o + 4u * r : GA_OOB, 0, 0);
            }
        }
    }
}

// out[m][n] = sum over the splits of part[s][m][n] (part rows are N floats long), in a FIXED order: lane l of the L lanes that
// share an output vector adds splits l, l + L, l + 2L ... (eight loads in flight), then the L lane sums are added in
// ascending lane order.  L = 1 for few splits; a 60 x 60 weight gradient summed over 144 splits by one thread per vector was
// 18 dependent batches of loads, 15 us for 2 MB.  V floats per vector.
template <int V, int L>
__global__ __launch_bounds__(256) void any_reduce_kernel(const float *__restrict__ part, int splits, int M, int N, float *__restrict__ out,
                                                         int64_t ldc)
{
    const int64_t count = (int64_t)M * N;
    const int l = threadIdx.x % L;
    const int64_t e = ((int64_t)blockIdx.x * (256 / L) + threadIdx.x / L) * V;
    const bool live = e < count;
    float t[V];
#pragma unroll
    for (int i = 0; i < V; ++i) t[i] = 0.f;
    if (live) {
        for (int s0 = l; s0 < splits; s0 += 8 * L) {
            float v[8][V];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u * L < splits ? s0 + u * L : l;
                if constexpr (V == 4) {
                    const float4 w = *reinterpret_cast<const float4 *>(part + (int64_t)s * count + e);
                    v[u][0] = w.x, v[u][1] = w.y, v[u][2] = w.z, v[u][3] = w.w;
                } else {
                    v[u][0] = part[(int64_t)s * count + e];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u * L < splits) {
#pragma unroll
                    for (int i = 0; i < V; ++i) t[i] = (s0 + u * L == l) ? v[u][i] : t[i] + v[u][i];
                }
        }
    }
    if constexpr (L > 1) { // the L lanes of a vector are neighbours in one wave: fold them in lane order
        float r[V];
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = __shfl(t[i], (int)(threadIdx.x & 63) / L * L);
#pragma unroll
        for (int j = 1; j < L; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) r[i] += __shfl(t[i], (int)(threadIdx.x & 63) / L * L + j);
#pragma unroll
        for (int i = 0; i < V; ++i) t[i] = r[i];
    }
    if (live && l == 0) {
        float *o = out + (e / N) * ldc + e % N; // (V == 4: N % 4 == 0, the four elements share a row)
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = t[i];
    }
}

struct AnyPlan {
    int tile_m; // 64: 64-row tiles, 32-deep stages; 16: 64-row tiles, 16-deep stages
    int tiles_m, tiles_n, splits, k_chunk;
};

inline AnyPlan any_plan(int m, int n, int k, bool vector_forms)
{
    AnyPlan p;
    p.tiles_n = (n + GA_T - 1) / GA_T;
    // 64 x 64 tiles; `tile_m` also carries the stage depth: 16 = 16-deep stages (20 KB of LDS, 8 workgroups per CU: the whole
    // grid of a big product is resident at once -- 37.6 vs 39.8 us at 18432 x 300 x 300), 64 = 32-deep stages.
    // Measured and rejected (tools/time_gemm_any.py, tools/probe/gemm_any_variants.sh): a tall 128 x 64 tile (twice the MFMAs
    // per fragment read and per staged byte: 42.2 vs 39.8 us), loads two stages ahead instead of one (no change); with the
    // operand loads AND half the fragment reads compiled out the launch still takes 37 of its 43 us -- 24 us of MFMA issue +
    // what 1440 ten-stage workgroups pay in prologue, epilogue and barriers.
    const int t64 = (m + 63) / 64;
    p.tiles_m = t64;
    p.tile_m = (vector_forms && (int64_t)t64 * p.tiles_n >= 1024) ? 16 : 64;
    static const char *force = getenv("GEOM_GEMM_TILE"); // tools: "64" / "16"
    if (force && vector_forms) p.tile_m = atoi(force) == 16 ? 16 : 64;
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    p.splits = 1, p.k_chunk = k > 0 ? k : 1;
    if (tiles < 512 && k >= 512) { // few tiles against a long sum: ~768 workgroups, chunks of whole stages, >= 64 deep
        int want = (int)((768 + tiles - 1) / tiles);
        if (want > k / 64) want = k / 64;
        if (want > 1) {
            int chunk = (k + want - 1) / want;
            chunk = (chunk + GA_BK - 1) / GA_BK * GA_BK;
            p.k_chunk = chunk;
            p.splits = (k + chunk - 1) / chunk;
        }
    }
    return p;
}

inline int vec_of(int extent, int64_t ld) { return (extent % 4 == 0 && ld % 4 == 0) ? 4 : (extent % 2 == 0 && ld % 2 == 0) ? 2 : 1; }

template <bool A_KM, bool B_KN, int VA, int VB>
void any_launch_c(const AnyArgs &q, int vc, int tile_m, dim3 grid, hipStream_t s)
{
    if constexpr (VA == 4 && VB == 4) { // the vector-load forms may take 16-deep stages (20 KB of LDS: 8 workgroups per CU)
        if (tile_m == 16) {
            if (vc == 4) hipLaunchKernelGGL((any_gemm_kernel<A_KM, B_KN, VA, VB, 4, 2, 16>), grid, dim3(GA_THREADS), 0, s, q);
            else hipLaunchKernelGGL((any_gemm_kernel<A_KM, B_KN, VA, VB, 1, 2, 16>), grid, dim3(GA_THREADS), 0, s, q);
            return;
        }
    }
    if (vc == 4) hipLaunchKernelGGL((any_gemm_kernel<A_KM, B_KN, VA, VB, 4, 2, 32>), grid, dim3(GA_THREADS), 0, s, q);
    else hipLaunchKernelGGL((any_gemm_kernel<A_KM, B_KN, VA, VB, 1, 2, 32>), grid, dim3(GA_THREADS), 0, s, q);
}

template <bool A_KM, bool B_KN, int VA>
void any_launch_b(const AnyArgs &q, int vb, int vc, int tile_m, dim3 grid, hipStream_t s)
{
    if (vb == 4) any_launch_c<A_KM, B_KN, VA, 4>(q, vc, tile_m, grid, s);
    else if (vb == 2) any_launch_c<A_KM, B_KN, VA, 2>(q, vc, tile_m, grid, s);
    else any_launch_c<A_KM, B_KN, VA, 1>(q, vc, tile_m, grid, s);
}

template <bool A_KM, bool B_KN>
void any_launch(const AnyArgs &q, int va, int vb, int vc, int tile_m, dim3 grid, hipStream_t s)
{
    if (va == 4) any_launch_b<A_KM, B_KN, 4>(q, vb, vc, tile_m, grid, s);
    else if (va == 2) any_launch_b<A_KM, B_KN, 2>(q, vb, vc, tile_m, grid, s);
    else any_launch_b<A_KM, B_KN, 1>(q, vb, vc, tile_m, grid, s);
}

} // namespace

extern "C" int64_t geom_gemm_workspace_floats(int m, int n, int k)
{
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const AnyPlan p = any_plan(m, n, k, false), t = any_plan(m, n, k, true); // whichever tile the call will take
    const int splits = p.splits > t.splits ? p.splits : t.splits;
    return splits > 1 ? (int64_t)splits * m * n : 0;
}

extern "C" int geom_gemm_f32(int m, int n, int k, const float *a, int64_t lda, int a_km, const float *b, int64_t ldb, int b_kn, float *c,
                             int64_t ldc, float *workspace, int64_t workspace_floats, void *stream)
{
    if (m < 0 || n < 0 || k < 0) return GEOM_EINVAL;
    if (m == 0 || n == 0) return 0;
    if (!c || ldc < n) return GEOM_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (k == 0) { // an empty sum
        return (int)hipMemset2DAsync(c, (size_t)ldc * 4, 0, (size_t)n * 4, (size_t)m, s) ? GEOM_EINVAL : 0;
    }
    if (!a || !b || lda < (a_km ? m : k) || ldb < (b_kn ? n : k)) return GEOM_EINVAL;
    AnyArgs q;
    q.a = a, q.b = b, q.c = c, q.lda = lda, q.ldb = ldb, q.ldc = ldc, q.M = m, q.N = n, q.K = k;
    q.a_bytes = ((int64_t)((a_km ? k : m) - 1) * lda + (a_km ? m : k)) * 4;
    q.b_bytes = ((int64_t)((b_kn ? k : n) - 1) * ldb + (b_kn ? n : k)) * 4;
    q.c_bytes = ((int64_t)(m - 1) * ldc + n) * 4;
    if (q.a_bytes >= 0x7fffffffLL || q.b_bytes >= 0x7fffffffLL || q.c_bytes >= 0x7fffffffLL) return GEOM_ETOOBIG; // 32-bit buffer offsets
    const int va = a_km ? vec_of(m, lda) : vec_of(k, lda), vb = b_kn ? vec_of(n, ldb) : vec_of(k, ldb);
    AnyPlan p = any_plan(m, n, k, va == 4 && vb == 4);
    if (p.splits > 1 && (!workspace || workspace_floats < (int64_t)p.splits * m * n)) p.splits = 1, p.k_chunk = k; // no room: one pass
    q.tiles_m = p.tiles_m, q.tiles_n = p.tiles_n, q.splits = p.splits, q.k_chunk = p.k_chunk, q.split_stride = 0;
    int vc = vec_of(n, ldc);
    if (p.splits > 1) {
        q.c = workspace, q.ldc = n, q.split_stride = (int64_t)m * n, q.c_bytes = (int64_t)m * n * 4;
        vc = vec_of(n, n);
        if (((int64_t)m * n) % 4) vc = 1; // the splits' tiles start at multiples of m * n floats
    }
    if (vc == 2) vc = 1;
    const int64_t units = (int64_t)p.tiles_m * p.splits;
    const int64_t blocks = (units + 7) / 8 * 8 * p.tiles_n;
    if (blocks > 0x7fffffffLL) return GEOM_ETOOBIG;
    const dim3 grid((unsigned)blocks);
    if (a_km && b_kn) any_launch<true, true>(q, va, vb, vc, p.tile_m, grid, s);
    else if (a_km) any_launch<true, false>(q, va, vb, vc, p.tile_m, grid, s);
    else if (b_kn) any_launch<false, true>(q, va, vb, vc, p.tile_m, grid, s);
    else any_launch<false, false>(q, va, vb, vc, p.tile_m, grid, s);
    if (p.splits > 1) {
        const int64_t count = (int64_t)m * n;
        // lanes per output vector: enough of them that the chip sees >= ~64 k loads in flight, at most 16
        const int64_t vecs = n % 4 == 0 ? count / 4 : count;
        const int lanes = (p.splits >= 32 && vecs < 16384) ? 16 : (p.splits >= 16 && vecs < 65536) ? 4 : 1;
        const dim3 rg((unsigned)((vecs + 256 / lanes - 1) / (256 / lanes)));
        if (n % 4 == 0) {
            if (lanes == 16) hipLaunchKernelGGL((any_reduce_kernel<4, 16>), rg, dim3(256), 0, s, workspace, p.splits, m, n, c, ldc);
            else if (lanes == 4) hipLaunchKernelGGL((any_reduce_kernel<4, 4>), rg, dim3(256), 0, s, workspace, p.splits, m, n, c, ldc);
            else hipLaunchKernelGGL((any_reduce_kernel<4, 1>), rg, dim3(256), 0, s, workspace, p.splits, m, n, c, ldc);
        } else {
            if (lanes == 16) hipLaunchKernelGGL((any_reduce_kernel<1, 16>), rg, dim3(256), 0, s, workspace, p.splits, m, n, c, ldc);
            else if (lanes == 4) hipLaunchKernelGGL((any_reduce_kernel<1, 4>), rg, dim3(256), 0, s, workspace, p.splits, m, n, c, ldc);
            else hipLaunchKernelGGL((any_reduce_kernel<1, 1>), rg, dim3(256), 0, s, workspace, p.splits, m, n, c, ldc);
        }
    }
    return geom::launch_status();
}
