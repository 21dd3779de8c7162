// Dense products of the 0N-GCN layers on the gfx950 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32).
//
// One 0N-GCN layer is  out = act([A . S[:, :k] | S[:, k:]] + bias)  with  S = X . W  (reference layers.py:107-116,
// 30-41, 140-152).  `X . W`, and in the backward `dX = G . W^T` and `dW = X^T . G`, are dense [b*V, Cin] x [Cin, C]
// contractions: 66 % of the step at the BASELINE shard when they are library calls.  The shapes are the awkward kind for
// a library -- 20 496 rows against a 192-wide output, a 963-wide (4-byte aligned, never 16) inner dimension, a
// weight gradient whose output is 192 x 192 against 20 496 summed rows -- and every one of them has work before or after
// it that only this code can fold in: the bias + ReLU + sign bits of the pass-through columns (so that the aggregation
// kernel touches k = C/3 columns instead of C), a composite gradient operand [A^T g' | g . relu'] that is never
// materialised, split-K partial sums that go to the end-of-pass reduction launch that exists anyway.
//
// fp32 MFMA runs at the fp32 VECTOR rate (64 flop/clk/SIMD = 157.3 TFLOP/s, MI355X_MICROARCH.md): one instruction is 32
// cycles for 2 operand registers, 16x less operand traffic per cycle than the bf16 forms -- so these kernels are bound by
// the matrix pipe itself and the design goal is the opposite of a bf16 GEMM's: keep ONE workgroup of 4 waves per CU
// issuing MFMAs back to back and get everything else (staging, LDS reads, epilogue) out of the way in the gaps.
//   * tile = RB row-blocks of 16 rows x NCW*64 output columns per workgroup; each of the 4 waves owns NCW column blocks
//     of all RB row-blocks (RB*NCW independent accumulators: no dependent-MFMA stalls), so a k-step of 4 needs RB + NCW
//     ds_read_b32 for RB*NCW MFMAs;
//   * the inner dimension advances 32 elements per stage through double-buffered LDS: the global loads of stage s+1 are
//     issued before the MFMAs of stage s and written to LDS after them, one barrier per stage;
//   * 20 496 rows = 1281 row-blocks = 5 per CU + ONE: the leftover row-blocks are not given to a few workgroups as a
//     sixth row-block (+20 % for them = for the launch) but spread fragment by fragment: workgroups 0..NCW-1 stage the
//     leftover rows as well and each of their waves computes one extra fragment (+1 MFMA on 15);
//   * operands are consumed in the layout they have in memory (row-major X and G, [Cin, C] weights): panels whose
//     inner index is contiguous are stored [row][34] in LDS, panels whose inner index is the slow one [t][R+16] -- both
//     conflict-free for the fragment reads (bank = 2*row + t resp. 16*g + x);
//   * a weight gradient (inner dimension = all rows) is split over the rows so that all CUs work; partial tiles go to a
//     workspace and are added up in slot order by a reduction kernel -- a fixed order, so results are bit-reproducible.
#include "geom_common.h"
#include "adam_math.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; }; // 16 bytes at 4-byte alignment (963-float rows)

constexpr int DG_THREADS = 256;
constexpr int DG_WAVES = 4;
constexpr int DG_BK = 32; // inner-dimension elements per LDS stage
constexpr int DG_KS = 8;  // MFMA k-steps (4 elements each) per stage
constexpr int SPLIT_RB = 6;        // weight-gradient tiles: 96 rows x 192 columns ...
constexpr int SPLIT_RB_NARROW = 3; // ... 48 rows for layers of <= 192 inputs (see split_geometry)
#ifndef DG_SPLIT_KP
#define DG_SPLIT_KP 1   // measured (profiles/r04_split_kernel_ablation.txt): two waves per SIMD buy nothing -- 66.8 vs 65.7 us
#endif
constexpr int SPLIT_KP_WIDE = DG_SPLIT_KP;   // k-parts (waves per SIMD) of the stand-alone weight-gradient launch of a wide layer, see split_stage
constexpr int LD_TC = 34; // stride of a [row][t] panel: fragment reads hit bank (2*row + t) % 32 -- all distinct

// stride of a [t][r] panel of R columns: the smallest s >= R with s % 32 == 16 (rows t and t+1 half a bank row apart)
__host__ __device__ constexpr int ld_rc(int r) { return ((r + 15) / 32) * 32 + 16; }

// ---------------------------------------------------------------------------------------------------------------------
// Panel staging.  A panel is the slice of one operand a stage needs: R rows x 32 inner elements.
//   TC ("t contiguous"): element (r, t) at g[r * ld + t] in memory, at lds[r * 34 + t] in LDS;
//   RC ("r contiguous"): element (r, t) at g[t * ld + r] in memory, at lds[t * ld_rc(R) + r] in LDS.
// Loads go to registers first (issued before a stage's MFMAs, stored to LDS after them).  Rows beyond the valid range
// are clamped (their results are never stored), inner positions beyond `tmax` are zero-filled.
// ---------------------------------------------------------------------------------------------------------------------

// A loader has three steps so that nothing between "issue" and the MFMAs waits for memory:
//   prepare(...)     per tile / chunk: the byte offset of every element this thread fetches relative to the stage's base,
//                    rows and columns CLAMPED into the valid range (clamped duplicates only feed outputs that are not stored);
//   issue_pass(p..)  unconditional load at base + min(offset, limit): `base` is wave-uniform (it carries the stage's inner
//                    offset), `limit` = the last valid position of the whole operand, so the stage that holds the end of the
//                    inner dimension reads valid memory without a branch;
//   store_pass(p..)  registers -> LDS; ZERO: inner positions >= tmax are zero-filled (only the A operand does that: one
//                    zero factor is enough).
// A pass is one load / store instruction per thread; the pipeline below spreads the passes over a stage's MFMAs.
__device__ __forceinline__ float ldg(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ f32x4 ldg4(const float *base, unsigned byte_off)
{ // 16 bytes at 4-byte alignment (still ONE global_load_dwordx4): the 963-float rows of the first layer's features
    const f4u v = *reinterpret_cast<const f4u *>(reinterpret_cast<const char *>(base) + byte_off);
    return (f32x4){v.x, v.y, v.z, v.w};
}
struct __attribute__((packed, aligned(4))) f3u { float x, y, z; };
__device__ __forceinline__ f3u ldg3(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const f3u *>(reinterpret_cast<const char *>(base) + byte_off);
}

// TC panel, 16-byte loads (rows 16-byte aligned, T % 4 == 0): 8 threads per row, 32 rows per pass
template <int R>
struct PanelTCv {
    static constexpr int PASSES = (R + 31) / 32;
    static constexpr int WIDTH = 4;
    f32x4 v[2][PASSES]; // two register sets: one being loaded while the other waits for its turn to go to LDS
    unsigned off[PASSES];
    // rowoff(r) -> ELEMENT offset of (clamped) row r
    template <typename RowOff>
    __device__ __forceinline__ void prepare(RowOff rowoff, unsigned /*ld*/)
    {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int r = p * 32 + (threadIdx.x >> 3);
            off[p] = ((unsigned)rowoff(r < R ? r : R - 1) + (threadIdx.x & 7) * 4u) * 4u;
        }
    }
    static __device__ __forceinline__ const float *stage_base(const float *g, unsigned /*ld*/, int t0) { return g + t0; }
    static __device__ __forceinline__ unsigned stage_limit(unsigned total, unsigned /*ld*/, int t0) { return (total - 4u - (unsigned)t0) * 4u; }
    template <int SET>
    __device__ __forceinline__ void issue_pass(int p, const float *base, unsigned limit) { v[SET][p] = ldg4(base, min(off[p], limit)); }
    template <bool ZERO, int SET>
    __device__ __forceinline__ void store_pass(int p, float *lds, int t0, int tmax) const
    {
        const int tl = (threadIdx.x & 7) * 4;
        const int r = p * 32 + (threadIdx.x >> 3);
        if (R % 32 == 0 || r < R) { // 34-float rows are 8-byte aligned: two ds_write_b64
            const bool tin = !ZERO || t0 + tl < tmax;
            float2 *d = reinterpret_cast<float2 *>(lds + r * LD_TC + tl);
            d[0] = make_float2(tin ? v[SET][p][0] : 0.f, tin ? v[SET][p][1] : 0.f);
            d[1] = make_float2(tin ? v[SET][p][2] : 0.f, tin ? v[SET][p][3] : 0.f);
        }
    }
};

// TC panel, 4-byte loads (rows that are only dword aligned: the 963-wide features): 32 threads per row, 8 rows per pass
template <int R>
struct PanelTCs {
    static constexpr int PASSES = R / 8;
    static constexpr int WIDTH = 1;
    float v[2][PASSES];
    unsigned off[PASSES];
    template <typename RowOff>
    __device__ __forceinline__ void prepare(RowOff rowoff, unsigned /*ld*/)
    {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) off[p] = ((unsigned)rowoff(p * 8 + (threadIdx.x >> 5)) + (threadIdx.x & 31)) * 4u;
    }
    static __device__ __forceinline__ const float *stage_base(const float *g, unsigned /*ld*/, int t0) { return g + t0; }
    static __device__ __forceinline__ unsigned stage_limit(unsigned total, unsigned /*ld*/, int t0) { return (total - 1u - (unsigned)t0) * 4u; }
    template <int SET>
    __device__ __forceinline__ void issue_pass(int p, const float *base, unsigned limit) { v[SET][p] = ldg(base, min(off[p], limit)); }
    template <bool ZERO, int SET>
    __device__ __forceinline__ void store_pass(int p, float *lds, int t0, int tmax) const
    {
        const int tl = threadIdx.x & 31;
        const bool tin = !ZERO || t0 + tl < tmax;
        lds[(p * 8 + (threadIdx.x >> 5)) * LD_TC + tl] = tin ? v[SET][p] : 0.f;
    }
};

// RC panel, 16-byte loads along r: element e = thread + NT * p of the 32 x R/4 grid (NT = threads of the workgroup)
template <int R, int NT = DG_THREADS>
struct PanelRCv {
    static constexpr int Q = R / 4;
    static constexpr int PASSES = (32 * Q + NT - 1) / NT;
    static constexpr bool EXACT = (32 * Q) % NT == 0;
    static constexpr int WIDTH = 4;
    f32x4 v[2][PASSES];
    unsigned off[PASSES];
    // coloff(r4) -> ELEMENT offset of the (clamped) 4-column group r4 within a row; ld = row pitch in elements
    template <typename ColOff>
    __device__ __forceinline__ void prepare(ColOff coloff, unsigned ld)
    {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int e = threadIdx.x + p * NT;
            const int tl = EXACT ? e / Q : min(e / Q, 31);
            off[p] = ((unsigned)tl * ld + (unsigned)coloff(e % Q)) * 4u;
        }
    }
    static __device__ __forceinline__ const float *stage_base(const float *g, unsigned ld, int t0) { return g + (int64_t)t0 * ld; }
    static __device__ __forceinline__ unsigned stage_limit(unsigned total, unsigned ld, int t0) { return (total - 4u - (unsigned)t0 * ld) * 4u; }
    template <int SET>
    __device__ __forceinline__ void issue_pass(int p, const float *base, unsigned limit) { v[SET][p] = ldg4(base, min(off[p], limit)); }
    template <bool ZERO, int SET>
    __device__ __forceinline__ void store_pass(int p, float *lds, int t0, int tmax) const
    {
        // past the panel's last row (EXACT == false) the thread holds a copy of row 31's element (prepare clamps): it stores the
        // same value to the same address -- no branch inside the MFMA stream
        const int e = threadIdx.x + p * NT;
        const int tl = EXACT ? e / Q : min(e / Q, 31), r4 = e % Q;
        f32x4 x = v[SET][p];
        if (ZERO && !(t0 + tl < tmax)) x = (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(lds + tl * ld_rc(R) + r4 * 4) = x;
    }
};

// RC panel, 4-byte loads along r (963-wide rows): element e = thread + NT * p of the 32 x R grid
template <int R, int NT = DG_THREADS>
struct PanelRCs {
    static constexpr int PASSES = (32 * R + NT - 1) / NT;
    static constexpr bool EXACT = (32 * R) % NT == 0;
    static constexpr int WIDTH = 1;
    float v[2][PASSES];
    unsigned off[PASSES];
    template <typename ColOff>
    __device__ __forceinline__ void prepare(ColOff coloff, unsigned ld)
    {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int e = threadIdx.x + p * NT;
            const int tl = EXACT ? e / R : min(e / R, 31);
            off[p] = ((unsigned)tl * ld + (unsigned)coloff(e % R)) * 4u;
        }
    }
    static __device__ __forceinline__ const float *stage_base(const float *g, unsigned ld, int t0) { return g + (int64_t)t0 * ld; }
    static __device__ __forceinline__ unsigned stage_limit(unsigned total, unsigned ld, int t0) { return (total - 1u - (unsigned)t0 * ld) * 4u; }
    template <int SET>
    __device__ __forceinline__ void issue_pass(int p, const float *base, unsigned limit) { v[SET][p] = ldg(base, min(off[p], limit)); }
    template <bool ZERO, int SET>
    __device__ __forceinline__ void store_pass(int p, float *lds, int t0, int tmax) const
    {
        const int e = threadIdx.x + p * NT;
        const int tl = EXACT ? e / R : min(e / R, 31), r = e % R;      // (clamped duplicates: see PanelRCv)
        lds[tl * ld_rc(R) + r] = (!ZERO || t0 + tl < tmax) ? v[SET][p] : 0.f;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// One stage of the pipeline = 8 k-steps of MFMAs out of LDS buffer `cur`, with everything else of the stage issued in
// their shadow (one wave per SIMD: an MFMA occupies the matrix pipe for 32 cycles, the wave can issue ~5 other
// instructions meanwhile -- MI355X_MICROARCH.md):
//   k-steps 0-3: the loads of the stage AFTER NEXT are issued into staging register set SET;
//   k-steps 4-6: the other register set (the NEXT stage's operand slices, loaded during the previous stage: a full stage
//                = 1.6 us of latency budget; half a stage measured as `s_waitcnt vmcnt` stalls) goes to LDS buffer `wr`;
//                then ONE barrier;
//   every k-step: the fragments of the following k-step are requested first -- the last one reads `wr`, i.e. the first
//                k-step of the next stage, so the MFMA stream never stops between stages.
// Two LDS buffers and the single barrier are enough: between two barriers every wave reads only `cur` (its last request, for
// k-step 7, is issued in k-step 6 in front of the barrier) and writes only `wr`, which nobody reads before the barrier.
// 2 x 40 KB per workgroup: TWO workgroups fit a CU's 160 KB -- see the combined backward launch.
// The accumulator of fragment (rb, jj) of this wave holds, in lane l = 16*g + x, C[16*rb + x][16*cb + 4*g + r] for
// r = 0..3 (cb = the wave's jj-th column block): the weight-side fragment is passed as the instruction's first operand, so
// a lane's four registers are four CONSECUTIVE output columns (one 16-byte store).
// ---------------------------------------------------------------------------------------------------------------------
template <int RB, int NCW, bool EXTRA>
struct Frags {
    float a[2][RB + 1], b[2][NCW + 1];
};

template <int RB, int NCW, bool A_TC, bool B_TC, int RA, int RBW, bool EXTRA>
__device__ __forceinline__ void fetch_frags(Frags<RB, NCW, EXTRA> &f, int buf, const float *la, const float *lb, int s, int wave,
                                            int xcb)
{
    const int x = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
    const float *pa = A_TC ? la + x * LD_TC + g : la + g * ld_rc(RA) + x;
    const float *pb = B_TC ? lb + (wave * NCW * 16 + x) * LD_TC + g : lb + g * ld_rc(RBW) + wave * NCW * 16 + x;
#pragma unroll
    for (int i = 0; i < RB; ++i) f.a[buf][i] = A_TC ? pa[i * 16 * LD_TC + 4 * s] : pa[4 * s * ld_rc(RA) + i * 16];
#pragma unroll
    for (int j = 0; j < NCW; ++j) f.b[buf][j] = B_TC ? pb[j * 16 * LD_TC + 4 * s] : pb[4 * s * ld_rc(RBW) + j * 16];
    if (EXTRA) { // the leftover row-block: rows RB*16.. of the A panel, column block xcb
        const float *pbx = B_TC ? lb + (xcb * 16 + x) * LD_TC + g : lb + g * ld_rc(RBW) + xcb * 16 + x;
        f.a[buf][RB] = A_TC ? pa[RB * 16 * LD_TC + 4 * s] : pa[4 * s * ld_rc(RA) + RB * 16];
        f.b[buf][NCW] = B_TC ? pbx[4 * s] : pbx[4 * s * ld_rc(RBW)];
    }
}

struct StageIO {
    int st_t0, st_tmax;      // inner range of the stage held in the staging registers (zero-fill of the A operand)
    const float *a_base, *b_base; // stage bases + clamp limits of the loads to issue
    unsigned a_limit, b_limit;
};

template <int RB, int NCW, bool A_TC, bool B_TC, int RA, int RBW, bool EXTRA, int A_FLOATS, int SET, class PA, class PB, class Tail>
__device__ __forceinline__ void gemm_stage(const float *cur, float *wr, f32x4 (&acc)[RB][NCW], f32x4 &accx,
                                           Frags<RB, NCW, EXTRA> &f, PA &pa, PB &pb, const StageIO &io, int wave, int xcb,
                                           Tail tail)
{
    constexpr int U = PA::PASSES + PB::PASSES; // staging instructions per thread and direction
#pragma unroll
    for (int s = 0; s < DG_KS; ++s) {
        // fragments of the next k-step (of the next stage after the last one)
#ifndef DG_PROBE_NO_FETCH
        if (s + 1 < DG_KS) fetch_frags<RB, NCW, A_TC, B_TC, RA, RBW, EXTRA>(f, (s + 1) & 1, cur, cur + A_FLOATS, s + 1, wave, xcb);
        else fetch_frags<RB, NCW, A_TC, B_TC, RA, RBW, EXTRA>(f, 0, wr, wr + A_FLOATS, 0, wave, xcb);
#endif
        // this k-step's share of the staging work
#ifdef DG_PROBE_NO_ISSUE
        if (false) {
#else
        if (s < 4) {
#endif
#pragma unroll
            for (int u = s * U / 4; u < (s + 1) * U / 4; ++u) {
                if (u < PA::PASSES) pa.template issue_pass<SET>(u, io.a_base, io.a_limit);
                else pb.template issue_pass<SET>(u - PA::PASSES, io.b_base, io.b_limit);
            }
#ifdef DG_PROBE_NO_STORE
        } else if (false) {
#else
        } else if (s < 7) {
#endif
#pragma unroll
            for (int u = (s - 4) * U / 3; u < (s - 3) * U / 3; ++u) {
                if (u < PA::PASSES) pa.template store_pass<true, SET ^ 1>(u, wr, io.st_t0, io.st_tmax);
                else pb.template store_pass<false, SET ^ 1>(u - PA::PASSES, wr + A_FLOATS, io.st_t0, io.st_tmax);
            }
        }
        // the last k-step also carries the (scalar, branch-free) bookkeeping of the next stage: between two stages the
        // matrix pipe would otherwise sit idle for ~100 scalar instructions
        if (s == DG_KS - 1) tail();
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < NCW; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[s & 1][j], f.a[s & 1][i], acc[i][j], 0, 0, 0);
        if (EXTRA) accx = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[s & 1][NCW], f.a[s & 1][RB], accx, 0, 0, 0);
        // one MFMA, then what fits into its 32 cycles
#pragma unroll
        for (int m = 0; m < RB * NCW + (EXTRA ? 1 : 0); ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // DS read
            if (s < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // VMEM read
            else __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // DS write
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0); // VALU / SALU
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == 6) {
#ifndef DG_PROBE_NO_BARRIER
            __syncthreads(); // `wr` is complete; its first fragments are requested in the next k-step
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row kernel:  C[i][j] = sum_t A[i][t] * B(t, j),  A row-major [I, T]; tiles of RB*16 rows.
//   B_TC = false: B(t, j) = b[t * ldb + j]   (forward: W [Cin, C])
//   B_TC = true : B(t, j) = b[j * ldb + t]   (input gradient: W again, read as W^T)
// Workgroup w takes row tiles w, w + G, ...; all column chunks (NCW*64 wide) of a tile in turn.  The rows beyond the last
// full tile ("leftover" row-blocks, fewer than RB) are spread over the first workgroups as extra fragments (see top).
// ---------------------------------------------------------------------------------------------------------------------
enum { EPI_PLAIN = 0, EPI_ZN = 1 };

struct RowArgs {
    const float *a;
    int64_t lda;
    const float *b;
    int64_t ldb;
    float *c;
    int64_t ldc;
    int I, J, T;
    int n_tiles;   // full tiles of RB row-blocks
    int left_rb;   // leftover row-blocks (< RB), handled as extras
    int n_chunks;  // ceil(J / (NCW*64))
    // EPI_ZN: columns < ksplit go raw to sup[i * ksplit + j]; columns >= ksplit get bias + ReLU and go to c; one sign bit
    // per element of c's pass-through columns to mask[i * (J/16) + j/16] (bit j % 16)
    int ksplit;
    float *sup;
    const float *bias;
    unsigned short *mask;
};

template <int RB, int NCW, bool B_TC, bool A_VEC, int EPI, bool HAS_X>
__device__ __forceinline__ void rows_body(const RowArgs &q, float *lds, const int w, const int G)
{
    constexpr int RA = (RB + 1) * 16;     // rows of the A panel (the last 16 = a leftover row-block, when staged)
    constexpr int RAL = HAS_X ? RA : RB * 16;
    constexpr int CW = NCW * 64;          // output columns per chunk
    constexpr int A_FLOATS = RA * LD_TC;
    constexpr int B_FLOATS = B_TC ? CW * LD_TC : 32 * ld_rc(CW);
    constexpr int BUF = A_FLOATS + B_FLOATS;
    const int wave = threadIdx.x >> 6;
    // leftover row-block handled by this workgroup (every chunk of it): workgroups [NCW*e, NCW*e + NCW) take leftover
    // row-block e, wave v of workgroup NCW*e + u the column block 4*u + v of each chunk
    const int xrow0 = HAS_X ? (q.n_tiles * RB + w / NCW) * 16 : 0;
    const int xcb = (w % NCW) * DG_WAVES + wave;
    const int nst = (q.T + DG_BK - 1) / DG_BK;
    int my_tiles = w < q.n_tiles ? (q.n_tiles - 1 - w) / G + 1 : 0;
    if (my_tiles == 0 && HAS_X) my_tiles = 1; // extras only (fewer full tiles than workgroups): an empty tile carries them
    const int total = my_tiles * q.n_chunks * nst;
    if (total == 0) return;

    typedef typename std::conditional<A_VEC, PanelTCv<RAL>, PanelTCs<RAL>>::type PA;
    typedef typename std::conditional<B_TC, PanelTCv<CW>, PanelRCv<CW>>::type PB;
    PA pa;
    PB pb;
    const unsigned a_total = (unsigned)q.I * (unsigned)q.lda;
    const unsigned b_total = B_TC ? (unsigned)q.J * (unsigned)q.ldb : (unsigned)q.T * (unsigned)q.ldb;
    // the loader runs two stages ahead of the MFMAs; past the end it stays on the last stage (harmless re-loads)
    int l_tile = w, l_chunk = 0, l_st = 0, l_count = 0;
    auto prepare_a = [&]() {
        const int tile = l_tile;
        pa.prepare([=](int r) -> unsigned {
            // the leftover row-block (panel rows RB*16..) is fetched with every tile of the workgroup and used by the first
            int row = r < RB * 16 ? (tile < q.n_tiles ? tile : 0) * RB * 16 + r : xrow0 + (r - RB * 16);
            row = row < q.I ? row : q.I - 1;
            return (unsigned)row * (unsigned)q.lda;
        }, (unsigned)q.lda);
    };
    auto prepare_b = [&]() {
        const int j0 = l_chunk * CW;
        if constexpr (B_TC) {
            pb.prepare([=](int r) -> unsigned { const int j = j0 + r; return (unsigned)(j < q.J ? j : q.J - 1) * (unsigned)q.ldb; }, (unsigned)q.ldb);
        } else {
            pb.prepare([=](int r4) -> unsigned { const int j = j0 + r4 * 4; return (unsigned)(j < q.J ? j : q.J - 4); }, (unsigned)q.ldb);
        }
    };
    StageIO io;
    auto aim = [&]() { // bases / limits of the loader's current stage
        io.a_base = PA::stage_base(q.a, (unsigned)q.lda, l_st * DG_BK);
        io.a_limit = PA::stage_limit(a_total, (unsigned)q.lda, l_st * DG_BK);
        io.b_base = PB::stage_base(q.b, (unsigned)q.ldb, l_st * DG_BK);
        io.b_limit = PB::stage_limit(b_total, (unsigned)q.ldb, l_st * DG_BK);
    };
    // step the loader to the next stage (it stays on the last one at the end: harmless re-loads).  The frequent part is
    // branch-free -- it runs inside the MFMA stream; `wrapped` tells the caller that a new chunk / tile begins.
    bool wrapped = false;
    auto step = [&]() {
        const int more = l_count + 1 < total ? 1 : 0;
        l_count += more;
        const int nl = l_st + more;
        wrapped = nl == nst;
        l_st = wrapped ? 0 : nl;
    };
    auto rewire = [&]() { // the rare part: offsets of the new chunk / tile
        if (!wrapped) return;
        if (++l_chunk == q.n_chunks) {
            l_chunk = 0, l_tile += G;
            prepare_a();
        }
        if (q.n_chunks > 1) prepare_b();
    };
    auto advance = [&]() { step(); rewire(); };
    // prologue: stages 0 and 1 requested together (sets 0 and 1), stage 0 -> LDS buffer 0
    prepare_a();
    prepare_b();
    aim();
#pragma unroll
    for (int p = 0; p < PA::PASSES; ++p) pa.template issue_pass<0>(p, io.a_base, io.a_limit);
#pragma unroll
    for (int p = 0; p < PB::PASSES; ++p) pb.template issue_pass<0>(p, io.b_base, io.b_limit);
    advance();
    aim();
    io.st_t0 = l_st * DG_BK; // the stage in set 1: stored during stage 0
#pragma unroll
    for (int p = 0; p < PA::PASSES; ++p) pa.template issue_pass<1>(p, io.a_base, io.a_limit);
#pragma unroll
    for (int p = 0; p < PB::PASSES; ++p) pb.template issue_pass<1>(p, io.b_base, io.b_limit);
#pragma unroll
    for (int p = 0; p < PA::PASSES; ++p) pa.template store_pass<true, 0>(p, lds, 0, q.T);
#pragma unroll
    for (int p = 0; p < PB::PASSES; ++p) pb.template store_pass<false, 0>(p, lds + A_FLOATS, 0, q.T);
    advance();
    aim();
    io.st_tmax = q.T;
    __syncthreads();
    Frags<RB, NCW, HAS_X> fr;
    fetch_frags<RB, NCW, true, B_TC, RA, CW, HAS_X>(fr, 0, lds, lds + A_FLOATS, 0, wave, xcb);

    const int x = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
    f32x4 acc[RB][NCW], accx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NCW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int c_tile = w, c_chunk = 0, c_st = 0, c_pass = 0;
    int buf = 0; // LDS buffer of the stage being multiplied (it % 2)

    // after a stage: the registers of set SET now hold the stage the loader was aimed at; aim at the one after it, and
    // finish the chunk when this was its last stage
    auto after_stage = [&]() {
        rewire();
        if (++c_st < nst) return;
        c_st = 0;
        const int j0 = c_chunk * CW;
        auto emit = [&](const f32x4 &val, int row, int cb) {
            const int j = j0 + cb * 16 + 4 * g;
            if (row >= q.I || j >= q.J) return;
            f32x4 v = val;
            if (EPI == EPI_ZN) {
                if (j < q.ksplit) { // aggregated columns: raw support, compact [I, ksplit]
                    *reinterpret_cast<f32x4 *>(q.sup + (int64_t)row * q.ksplit + j) = v;
                    return;
                }
                const f32x4 bb = q.bias ? *reinterpret_cast<const f32x4 *>(q.bias + j) : (f32x4){0.f, 0.f, 0.f, 0.f};
                unsigned bits = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = v[r] + bb[r];
                    v[r] = t > 0.f ? t : 0.f;
                    bits |= (t > 0.f ? 1u : 0u) << r;
                }
                if (q.mask) { // 16 sign bits per (row, column block): the four g-lanes of a row combine their nibbles
                    unsigned m = bits << (4 * g);
                    m |= __shfl_xor(m, 16);
                    m |= __shfl_xor(m, 32);
                    if (g == 0) q.mask[(int64_t)row * (q.J >> 4) + (j >> 4)] = (unsigned short)m;
                }
            }
            float *dst = q.c + (int64_t)row * q.ldc + j;
            if (j + 3 < q.J) {
                *reinterpret_cast<f4u *>(dst) = f4u{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j + r < q.J) dst[r] = v[r];
            }
        };
        if (c_tile < q.n_tiles) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int jj = 0; jj < NCW; ++jj) emit(acc[i][jj], (c_tile * RB + i) * 16 + x, wave * NCW + jj);
        }
        if (HAS_X && c_pass == 0) emit(accx, xrow0 + x, xcb);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int jj = 0; jj < NCW; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accx = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (++c_chunk == q.n_chunks) c_chunk = 0, c_tile += G, ++c_pass;
    };

    // stage `it` multiplies LDS buffer it % 3, requests stage it + 2 into register set it % 2 and moves stage it + 1 (set
    // (it + 1) % 2) to buffer (it + 1) % 3: two stages per trip so that the register sets are compile-time names
    for (int it = 0; it < total; it += 2) {
        {
            const float *cur = lds + buf * BUF;
            float *wr = lds + (buf ^ 1) * BUF;
            gemm_stage<RB, NCW, true, B_TC, RA, CW, HAS_X, A_FLOATS, 0>(cur, wr, acc, accx, fr, pa, pb, io, wave, xcb, [&]() {
                // set 0 now holds the stage the loader was aimed at: it goes to LDS in the next stage; aim at the one after
                io.st_t0 = l_st * DG_BK;
                step();
                aim();
                buf ^= 1;
            });
            after_stage();
        }
        if (it + 1 < total) {
            const float *cur = lds + buf * BUF;
            float *wr = lds + (buf ^ 1) * BUF;
            gemm_stage<RB, NCW, true, B_TC, RA, CW, HAS_X, A_FLOATS, 1>(cur, wr, acc, accx, fr, pa, pb, io, wave, xcb, [&]() {
                io.st_t0 = l_st * DG_BK;
                step();
                aim();
                buf ^= 1;
            });
            after_stage();
        }
    }
}

template <int RB, int NCW, bool B_TC, bool A_VEC, int EPI>
__global__ __launch_bounds__(DG_THREADS) void dense_rows_kernel(RowArgs q)
{
    constexpr int RA = (RB + 1) * 16;
    constexpr int CW = NCW * 64;
    __shared__ __attribute__((aligned(16))) float lds[2 * (RA * LD_TC + (B_TC ? CW * LD_TC : 32 * ld_rc(CW)))];
    // the workgroups that carry a leftover row-block run their own instantiation of the whole loop (one more accumulator)
    if ((int)blockIdx.x < q.left_rb * NCW) rows_body<RB, NCW, B_TC, A_VEC, EPI, true>(q, lds, blockIdx.x, gridDim.x);
    else rows_body<RB, NCW, B_TC, A_VEC, EPI, false>(q, lds, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// Split kernel (weight gradient):  P[slot][i][j] = sum over this slot's rows t of  A[t][i0 + i] * B[t][j],
// A = X [T, I] and B = G [T, J <= NCW*64], both read as stored (the summed index is the slow one).  Output tiles of
// SPLIT_RB*16 rows of dW, each split s_full ways over T; the I % (SPLIT_RB*16) leftover rows (963 = 10 * 96 + 3) are
// single row-blocks with their own, smaller number of splits (a sixth of the work per split) and a 1-row-block body.
// Workgroup w < full_tiles * s_full: tile = w % full_tiles, split = w / full_tiles (neighbouring workgroups = the tiles
// of one split = the same rows of G: they share it through L2); the others: leftover row-block (w - nfull) / s_left.
// ---------------------------------------------------------------------------------------------------------------------
struct SplitArgs {
    const float *a;
    int64_t lda;
    const float *b;
    int64_t ldb;
    float *part;  // [slots][SPLIT_RB*16][NCW*64]; a leftover slot uses its first 16 rows
    int I, J, T;
    int full_tiles, s_full, left_rb, s_left;
    float *colsum; // optional [s_full][J]: column sums of B over each split's rows (the bias gradient), tile 0 only
};

// The split body keeps only the X panel in LDS.  G (the operand every wave needs a DIFFERENT 48-column slice of, and that
// all workgroups of a split share through L2) goes straight into fragment registers: lane (x, g) of a wave loads the 12
// bytes G[t0 + 4s + g][48*wave + 3x .. +2] -- 16 lanes x 12 B = one contiguous 192-byte run per row -- and MFMA number u of
// the k-step takes component u as its column-side operand, so accumulator (i, u) of the lane holds
// C[16 i + x'][48*wave + 12 g' + 3 r + u]: twelve CONSECUTIVE output columns per lane.  The registers form a ring over two
// stages x 8 k-steps: the slot a k-step has just consumed is refilled at once with the same k-step of the stage after
// next.  Measured before (probe builds with the LDS traffic removed): the G panel's LDS writes + fragment reads were ~15 of
// the first layer's 77 us.
// KP = k-parts: with KP == 2 the workgroup has EIGHT waves, two per SIMD -- wave v and wave v + 4 own the same 48 output
// columns and share the X panel, the first takes the even k-steps of every stage, the second the odd ones, each into its
// own accumulators (added up through LDS at the end: split_body).  One wave per SIMD cannot hide its own stalls (fragment
// reads, the stage barrier, s_waitcnt in front of the LDS stores): PMC showed the matrix pipe 64-73 % busy with one wave;
// two waves on a SIMD issue into each other's gaps.  A wave then runs KS = 8 / KP k-steps per stage: loads of the stage
// after next in its first KS/2 steps, LDS stores in the following ones, barrier behind step KS - 2, the first fragments of
// the next stage requested in the last step.
// X fragments of one k-step with WIDE LDS reads: the tile's rows are dealt to the MFMA row-blocks so that a lane's RB
// fragment values are neighbours in the [t][r] panel -- lane x of row-blocks 4q .. 4q+3 holds rows 64q + 4x + {0,1,2,3} (one
// ds_read_b128), a remaining pair rows 2x + {0,1} (ds_read_b64), a remaining single row x.  Which row an accumulator holds
// is only a matter of where the epilogue stores it (split_row); every output element is still the same sum in the same
// order.  6 row-blocks: 2 LDS reads per k-step instead of 6 -- on this chip an LDS read is not hidden behind the MFMAs, it
// takes issue cycles away from them (profiles/r05_mfma_corun.txt).
template <int RB>
__device__ __forceinline__ int split_row(int i, int x)
{
    constexpr int N4 = RB / 4, N2 = (RB % 4) / 2;
    if (i < 4 * N4) return 64 * (i >> 2) + 4 * x + (i & 3);
    if (i < 4 * N4 + 2 * N2) return 64 * N4 + 2 * x + (i - 4 * N4);
    return 64 * N4 + 32 * N2 + x;
}

template <int RB>
__device__ __forceinline__ void split_frags(float (&f)[RB], const float *row, int x)
{
    constexpr int N4 = RB / 4, N2 = (RB % 4) / 2, N1 = RB % 2;
#pragma unroll
    for (int q = 0; q < N4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 64 * q + 4 * x);
        f[4 * q] = v[0], f[4 * q + 1] = v[1], f[4 * q + 2] = v[2], f[4 * q + 3] = v[3];
    }
    if constexpr (N2 == 1) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(row + 64 * N4 + 2 * x);
        f[4 * N4] = v[0], f[4 * N4 + 1] = v[1];
    }
    if constexpr (N1 == 1) f[RB - 1] = row[64 * N4 + 32 * N2 + x];
}

template <int RB, int RA, int A_FLOATS, int SET, int KP, class PA, class Tail>
__device__ __forceinline__ void split_stage(const float *cur, float *wr, f32x4 (&acc)[RB][3], float (&fa)[2][RB], f3u (&bq)[2][DG_KS / KP],
                                            PA &pa, const StageIO &io, unsigned b_lane, unsigned b_step, int kpart, Tail tail)
{
    constexpr int U = PA::PASSES;
    constexpr int KS = DG_KS / KP;             // k-steps of this wave per stage
    constexpr int LOADS = KS / 2;              // steps [0, LOADS): issue the loads; [LOADS, KS - 1): LDS stores
    constexpr int STORES = KS - 1 - LOADS;
    const int x = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
    const float *b_base = io.b_base; // by value: `tail` re-aims io in the last k-step, the refills below belong to this aim
    const unsigned b_limit = io.b_limit;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#ifndef DG_PROBE_NO_FETCH
        { // X fragments of this wave's next k-step (of the next stage after the last one): global k-step KP * s' + kpart
            const float *src = s + 1 < KS ? cur : wr;
            const int sn = (s + 1 < KS ? KP * (s + 1) : 0) + kpart;
            split_frags<RB>(fa[(s + 1) & 1], src + (4 * sn + g) * ld_rc(RA), x);
        }
#endif
#ifdef DG_PROBE_NO_ISSUE
        if (false) {
#else
        if (s < LOADS) {
#endif
#pragma unroll
            for (int u = s * U / LOADS; u < (s + 1) * U / LOADS; ++u) pa.template issue_pass<SET>(u, io.a_base, io.a_limit);
#ifdef DG_PROBE_NO_STORE
        } else if (false) {
#else
        } else if (s < KS - 1) {
#endif
#pragma unroll
            for (int u = (s - LOADS) * U / STORES; u < (s - LOADS + 1) * U / STORES; ++u)
                pa.template store_pass<true, SET ^ 1>(u, wr, io.st_t0, io.st_tmax);
        }
        if (s == KS - 1) tail();
        const f3u b = bq[SET][s];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, fa[s & 1][i], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, fa[s & 1][i], acc[i][1], 0, 0, 0);
            acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, fa[s & 1][i], acc[i][2], 0, 0, 0);
        }
#ifndef DG_PROBE_NO_G
        bq[SET][s] = ldg3(b_base, min(b_lane + (unsigned)(KP * s) * b_step, b_limit)); // the same k-step of the stage after next
#endif
#pragma unroll
        for (int m = 0; m < RB * 3; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // DS read
            if (s < LOADS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // VMEM read
            else __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);           // DS write
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0); // VALU / SALU
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == KS - 2) {
#ifndef DG_PROBE_NO_BARRIER
            __syncthreads();
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int RB, int RBP, bool A_VEC, int KP = 1>
__device__ __forceinline__ void split_body(const SplitArgs &q, float *lds, int i0, int split, int nsplit, int slot, bool want_cs)
{
    constexpr int RA = RB * 16;
    constexpr int CW = 192;
    constexpr int A_FLOATS = 32 * ld_rc(RA);
    constexpr int BUF = A_FLOATS;
    constexpr int NT = DG_THREADS * KP;
    constexpr int KS = DG_KS / KP;
    const int wave = (threadIdx.x >> 6) & 3;     // which 48 output columns
    const int kpart = KP == 1 ? 0 : (int)(threadIdx.x >> 8); // which k-steps of a stage (waves v and v + 4 share a SIMD)
    const int x = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
    // rows of the summed dimension in units of 4 (one MFMA k-step); T % 4 != 0 is zero-filled by the X loader
    const int n4 = (q.T + 3) / 4;
    const int t_begin = (int)((int64_t)n4 * split / nsplit) * 4;
    const int t_end = min((int)((int64_t)n4 * (split + 1) / nsplit) * 4, q.T);
    const int nst = (t_end - t_begin + DG_BK - 1) / DG_BK;

    f32x4 acc[RB][3];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f; // column sums of G over this split's rows == g (mod 4), columns 48*wave + 3x ..

    typedef typename std::conditional<A_VEC, PanelRCv<RA, NT>, PanelRCs<RA, NT>>::type PA;
    PA pa;
    if (nst > 0) {
        if constexpr (A_VEC) pa.prepare([=](int r4) -> unsigned { const int i = i0 + r4 * 4; return (unsigned)(i + 3 < q.I ? i : q.I - 4); }, (unsigned)q.lda);
        else pa.prepare([=](int r) -> unsigned { const int i = i0 + r; return (unsigned)(i < q.I ? i : q.I - 1); }, (unsigned)q.lda);
        const unsigned a_total = (unsigned)q.T * (unsigned)q.lda, b_total = (unsigned)q.T * (unsigned)q.ldb;
        // G: this lane's 12 bytes of row g of a k-step, columns clamped into the row (J % 4 == 0 but maybe not % 3); the wave's
        // first k-step of a stage is k-step `kpart`
        const int jc = min(wave * 48 + 3 * x, q.J - 3);
        const unsigned b_step = 4u * (unsigned)q.ldb * 4u; // four rows per k-step
        const unsigned b_lane = ((unsigned)g * (unsigned)q.ldb + (unsigned)jc) * 4u + (unsigned)kpart * b_step;
        StageIO io;
        int l_st = 0;
        auto aim = [&]() {
            const int t0 = t_begin + l_st * DG_BK;
            io.a_base = PA::stage_base(q.a, (unsigned)q.lda, t0);
            io.a_limit = PA::stage_limit(a_total, (unsigned)q.lda, t0);
            io.b_base = q.b + (int64_t)t0 * q.ldb;
            io.b_limit = (b_total - 3u - (unsigned)t0 * (unsigned)q.ldb) * 4u;
        };
        auto advance = [&]() { l_st = l_st + 1 < nst ? l_st + 1 : l_st; };
        f3u bq[2][KS];
        float fa[2][RB];
        // prologue: stages 0 and 1 requested together; X of stage 0 -> LDS buffer 0
        aim();
#pragma unroll
        for (int p = 0; p < PA::PASSES; ++p) pa.template issue_pass<0>(p, io.a_base, io.a_limit);
#pragma unroll
        for (int s = 0; s < KS; ++s) bq[0][s] = ldg3(io.b_base, min(b_lane + (unsigned)(KP * s) * b_step, io.b_limit));
        advance();
        aim();
        io.st_t0 = t_begin + l_st * DG_BK;
#pragma unroll
        for (int p = 0; p < PA::PASSES; ++p) pa.template issue_pass<1>(p, io.a_base, io.a_limit);
#pragma unroll
        for (int s = 0; s < KS; ++s) bq[1][s] = ldg3(io.b_base, min(b_lane + (unsigned)(KP * s) * b_step, io.b_limit));
#pragma unroll
        for (int p = 0; p < PA::PASSES; ++p) pa.template store_pass<true, 0>(p, lds, t_begin, t_end);
        advance();
        aim();
        io.st_tmax = t_end;
        __syncthreads();
        split_frags<RB>(fa[0], lds + (4 * kpart + g) * ld_rc(RA), x);
        int buf = 0;
        auto tail = [&]() {
            io.st_t0 = t_begin + l_st * DG_BK;
            advance();
            aim();
            buf ^= 1;
        };
        // rows of G beyond t_end belong to the next split (their X factors are zero-filled): the bias gradient counts only
        // this split's rows
        auto colsum_stage = [&](const f3u (&b)[KS], int st) {
            const int t0 = t_begin + st * DG_BK + g + 4 * kpart;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bool in = t0 + 4 * KP * s < t_end;
                cs0 += in ? b[s].x : 0.f, cs1 += in ? b[s].y : 0.f, cs2 += in ? b[s].z : 0.f;
            }
        };
        for (int st = 0; st < nst; st += 2) {
            {
                const float *cur = lds + buf * BUF;
                float *wr = lds + (buf ^ 1) * BUF;
                if (want_cs) colsum_stage(bq[0], st);
                split_stage<RB, RA, A_FLOATS, 0, KP>(cur, wr, acc, fa, bq, pa, io, b_lane, b_step, kpart, tail);
            }
            if (st + 1 < nst) {
                const float *cur = lds + buf * BUF;
                float *wr = lds + (buf ^ 1) * BUF;
                if (want_cs) colsum_stage(bq[1], st + 1);
                split_stage<RB, RA, A_FLOATS, 1, KP>(cur, wr, acc, fa, bq, pa, io, b_lane, b_step, kpart, tail);
            }
        }
    }
    if constexpr (KP == 2) {
        // the odd k-steps' accumulators (and column sums) go to their even partners through LDS, lane for lane: float4
        // (wave, i, u) of lane l at [((wave * H + i') * 3 + u) * 64 + l] -- 1 KB contiguous per wave instruction -- H = 2
        // row-blocks per round, so that the exchange needs no more LDS than the kernel's two panels hold
        constexpr int H = 2;
        static_assert(DG_WAVES * H * 3 * 64 * 4 <= 2 * 32 * ld_rc(RBP * 16), "the exchange reuses the panel buffers");
        f32x4 *xch = reinterpret_cast<f32x4 *>(lds);
        const int l = threadIdx.x & 63;
#pragma unroll
        for (int i0r = 0; i0r < RB; i0r += H) {
            __syncthreads(); // the panels (first round) / the partners' reads of the previous round are done
            if (kpart == 1) {
#pragma unroll
                for (int i = i0r; i < RB && i < i0r + H; ++i)
#pragma unroll
                    for (int u = 0; u < 3; ++u) xch[((wave * H + (i - i0r)) * 3 + u) * 64 + l] = acc[i][u];
            }
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int i = i0r; i < RB && i < i0r + H; ++i)
#pragma unroll
                    for (int u = 0; u < 3; ++u) acc[i][u] = acc[i][u] + xch[((wave * H + (i - i0r)) * 3 + u) * 64 + l];
            }
        }
        if (want_cs) {
            __syncthreads();
            float *xf = lds;
            if (kpart == 1) xf[(wave * 3 + 0) * 64 + l] = cs0, xf[(wave * 3 + 1) * 64 + l] = cs1, xf[(wave * 3 + 2) * 64 + l] = cs2;
            __syncthreads();
            if (kpart == 0) cs0 += xf[(wave * 3 + 0) * 64 + l], cs1 += xf[(wave * 3 + 1) * 64 + l], cs2 += xf[(wave * 3 + 2) * 64 + l];
        }
        if (kpart == 1) return;
    }
    // partial tile: the lane's twelve consecutive columns of row 16 i + x
    float *dst = q.part + (int64_t)slot * (RBP * 16) * CW;
    const int j0 = wave * 48 + 12 * g;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float e[12];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 3; ++u) e[3 * r + u] = acc[i][u][r];
        f32x4 *d = reinterpret_cast<f32x4 *>(dst + split_row<RB>(i, x) * CW + j0);
        d[0] = (f32x4){e[0], e[1], e[2], e[3]};
        d[1] = (f32x4){e[4], e[5], e[6], e[7]};
        d[2] = (f32x4){e[8], e[9], e[10], e[11]};
    }
    if (want_cs) { // fold the four row classes (lanes x, x+16, x+32, x+48) in a fixed order; lanes g == 0 own the result
        float t0 = cs0, t1 = cs1, t2 = cs2;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            t0 += __shfl(cs0, x + 16 * k), t1 += __shfl(cs1, x + 16 * k), t2 += __shfl(cs2, x + 16 * k);
        }
        const int j = wave * 48 + 3 * x;
        if (g == 0 && j + 2 < q.J) {
            float *o = q.colsum + (int64_t)split * q.J + j;
            o[0] = t0, o[1] = t1, o[2] = t2;
        }
    }
}

// A_VEC: the leftover row-blocks may use 16-byte loads too (I % 4 == 0).  FULL tiles always do: their column groups never
// reach the end of a row, so 4-byte alignment is all they need (the 963-float rows of the first layer's features).
template <int RB, int NCW, bool A_VEC, int KP = 1>
__device__ __forceinline__ void split_dispatch(const SplitArgs &q, float *lds, const int w)
{
    static_assert(NCW == 3, "the split body owns 48 columns per wave");
    const int nfull = q.full_tiles * q.s_full;
    if (w < nfull) {
        const int tile = w % q.full_tiles, split = w / q.full_tiles;
        split_body<RB, RB, true, KP>(q, lds, tile * RB * 16, split, q.s_full, tile * q.s_full + split,
                                     q.colsum != nullptr && tile == 0);
    } else {
        const int e = (w - nfull) / q.s_left, split = (w - nfull) % q.s_left;
        split_body<1, RB, A_VEC, KP>(q, lds, (q.full_tiles * RB + e) * 16, split, q.s_left, w, false);
    }
}

// KP = 2: eight waves per workgroup, see split_stage -- built for the stand-alone launch of the 963-wide first layer (one
// workgroup per CU) on the suspicion that a single wave per SIMD leaves the matrix pipe idle during its own stalls.  The
// ablation says otherwise (tools/probe/run_probes.sh, profiles/r04_split_kernel_ablation.txt): with EVERYTHING but the
// MFMAs removed from the loop the launch still takes 59.5 us of its 65.7 -- 50 us of MFMA issue at the full 2.39 GHz the chip
// holds under this kernel + ~9.5 us of launch, first loads and the 18.5 MB burst of partial tiles at the end -- and a
// second wave per SIMD changes nothing (66.8).  Kept as a build option (-DDG_SPLIT_KP=2), the shipped value is 1.
template <int RB, int NCW, bool A_VEC, int KP>
__global__ __launch_bounds__(DG_THREADS * KP) void dense_split_kernel(SplitArgs q)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 32 * ld_rc(RB * 16)];
    split_dispatch<RB, NCW, A_VEC, KP>(q, lds, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// Both gradients of a layer's product in ONE launch, two workgroups per CU: workgroups [0, n_row) compute the input
// gradient dX = G . W^T (row tiles), workgroups [n_row, n_row + slots) the split partial sums of dW = X^T . G.  A workgroup
// of either kind uses 4 waves, <= 256 registers and <= 80 KB of LDS, so one of each shares a CU: two waves per SIMD whose
// stalls (prologue, epilogue, barriers, LDS round trips) are covered by the other's MFMAs.  Measured at the BASELINE shard,
// hidden layer (20 496 x 192 x 192): 31 us for the pair against 22 + 23 one after the other (library: 15 + 25).
// ---------------------------------------------------------------------------------------------------------------------
template <int RB, int SRB, bool X_VEC>
__global__ __launch_bounds__(DG_THREADS, 2) void dense_bwd_pair_kernel(RowArgs r, SplitArgs q, int n_row)
{
    constexpr int LDS_ROWS = 2 * ((RB + 1) * 16 * LD_TC + 192 * LD_TC);
    constexpr int LDS_SPLIT = 2 * 32 * ld_rc(SRB * 16);
    __shared__ __attribute__((aligned(16))) float lds[LDS_ROWS > LDS_SPLIT ? LDS_ROWS : LDS_SPLIT];
    const int w = blockIdx.x;
    if (w < n_row) {
        if (w < r.left_rb * 3) rows_body<RB, 3, true, true, EPI_PLAIN, true>(r, lds, w, n_row);
        else rows_body<RB, 3, true, true, EPI_PLAIN, false>(r, lds, w, n_row);
    } else {
        split_dispatch<SRB, 3, X_VEC>(q, lds, w - n_row);
    }
}

// out[i][j] = sum over the splits of tile(i) of part[slot][i % RA][j], in slot order.  One thread per 4 columns.
struct ReduceJob {
    const float *part;
    float *out;
    int I, J, RA, CW, full_tiles, s_full, s_left; // rows beyond the full tiles: 16-row tiles of s_left splits each
    int outs;                                     // outputs (of 4 columns) per workgroup, see reduce_outs()
};
struct ReduceJobs {
    ReduceJob job[GEOM_DENSE_MAX_REDUCE_JOBS];
    int first[GEOM_DENSE_MAX_REDUCE_JOBS + 1]; // flat grid: job j owns workgroups [first[j], first[j + 1])
    int n;
};
// Optional: the optimiser step on the gradients this launch finishes (geometrics_amd.optim.FusedAdam.fuse_into_backward):
// job i's output IS the gradient of parameter p[i] (same [I, J] layout), so the thread that writes a gradient element also
// updates the parameter and its two moments -- the stand-alone Adam launch (7.6 us) and its re-read of the gradients go
// away.  Same arithmetic and the same device-side step protocol as adam.hip (adam_math.h): same bits.
struct ReduceAdam {
    float *p[GEOM_DENSE_MAX_REDUCE_JOBS], *m[GEOM_DENSE_MAX_REDUCE_JOBS], *v[GEOM_DENSE_MAX_REDUCE_JOBS]; // null: no update
    float lr, b1, b2, eps;
    float *state; // null: no optimiser step in this launch
    int vec;      // every p / m / v pointer is 16-byte aligned: float4 accesses
};

// `outs` outputs (of 4 columns) x 256 / outs slot groups per workgroup: group sg adds its contiguous share of the slots in
// slot order (eight loads in flight), the group sums are then added in group order -- a fixed tree for a given (I, J,
// splits).  The more slots a job has, the more groups share them (reduce_outs): 64 x 4 for the 25 splits of the 963-wide
// layer, 16 x 16 for the 128 splits of a 192-wide one (whose 19 MB of partial tiles used to hang on 144 workgroups of 32-deep
// serial sums: 6.6 us alone against 3.9 for the first layer's), 4 x 64 for the 1 288 partial rows of a bias gradient.
constexpr int RED_THREADS = 256;
inline int reduce_outs(int slots) { return slots <= 32 ? 64 : slots <= 64 ? 32 : slots <= 128 ? 16 : slots <= 256 ? 8 : 4; }
// workgroup `block` of `nblocks` of a reduction; part = RED_THREADS f32x4 of LDS, st = 3 floats of LDS
template <bool ADAM> // false: no optimiser step (`adam` is not touched: a by-value ReduceAdam indexed by job lives in scratch)
__device__ __forceinline__ void reduce_body(const ReduceJobs &jobs, const ReduceAdam &adam, const int block, const int nblocks,
                                            f32x4 *part, float *st)
{
    float *const state = ADAM ? adam.state : nullptr;
    // flat grid: every workgroup holds outputs (a job smaller than the largest one used to leave whole workgroups idle:
    // 5 800 of 8 670 at the BASELINE shard)
    int jb = 0;
    while (jb + 1 < jobs.n && block >= jobs.first[jb + 1]) ++jb;
    const ReduceJob q = jobs.job[jb];
    const int bx = block - jobs.first[jb];
    const int jq = q.J >> 2;
    const int outs = q.outs, groups = RED_THREADS / outs;
    // the optimiser's step state is REQUESTED first and published to the workgroup at the barrier the partial sums need
    // anyway: its round trip rides under theirs.  (Thread 0 -- the one that signs the arrival at the end -- reads it with
    // agent-scope atomic loads and stores it to LDS in front of that barrier: "read the state, then arrive" stays a data
    // dependency, adam_math.h.)
    float s_t = 0.f, s_b1 = 0.f, s_b2 = 0.f;
    if (state && threadIdx.x == 0) {
        s_t = __hip_atomic_load(state + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_b1 = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_b2 = __hip_atomic_load(state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int o = threadIdx.x % outs, sg = threadIdx.x / outs;
    const int e = bx * outs + o;
    const bool live = e < q.I * jq;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    int i = 0, j = 0;
    if (live) {
        i = e / jq, j = (e % jq) * 4;
        int tile = i / q.RA, il = i % q.RA;
        const bool full = tile < q.full_tiles;
        const int n = full ? q.s_full : q.s_left;
        int slot0 = tile * q.s_full;
        if (!full) {
            const int r = i - q.full_tiles * q.RA;
            slot0 = q.full_tiles * q.s_full + (r >> 4) * q.s_left, il = r & 15;
        }
        const int64_t pitch = (int64_t)q.RA * q.CW;
        const float *p = q.part + ((int64_t)slot0 * q.RA + il) * q.CW + j;
        const int s0 = (int)((int64_t)n * sg / groups), s1 = (int)((int64_t)n * (sg + 1) / groups);
        // eight loads in flight per round (a group's share of the slots is 3-7 here: ONE round trip), added in slot order
        for (int s = s0; s < s1; s += 8) {
            f32x4 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s + k < s1) a[k] = *reinterpret_cast<const f32x4 *>(p + (int64_t)(s + k) * pitch);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s + k < s1) t = t + a[k];
        }
    }
    // the parameter and its moments (16 bytes at a time) are requested in front of the barrier as well
    const int64_t at = (int64_t)i * q.J + j;
    float *pp = state ? adam.p[jb] : nullptr;
    const bool vec = pp && adam.vec;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, bm = a, cv = a;
    if (vec && sg == 0 && live) {
        a = *reinterpret_cast<const f32x4 *>(pp + at);
        bm = *reinterpret_cast<const f32x4 *>(adam.m[jb] + at);
        cv = *reinterpret_cast<const f32x4 *>(adam.v[jb] + at);
    }
    part[sg * outs + o] = t;
    if (state && threadIdx.x == 0) st[0] = s_t, st[1] = s_b1, st[2] = s_b2;
    __syncthreads();
    geom::AdamStep as{};
    if (state) as = geom::adam_step_from(st, adam.lr, adam.b1, adam.b2); // uniform per workgroup
    if (sg == 0 && live) {
        f32x4 r = part[o];
        for (int k = 1; k < groups; ++k) r = r + part[k * outs + o];
        float *dst = q.out + at;
        if (vec) {
            f32x4 *p4 = reinterpret_cast<f32x4 *>(pp + at), *m4 = reinterpret_cast<f32x4 *>(adam.m[jb] + at),
                  *v4 = reinterpret_cast<f32x4 *>(adam.v[jb] + at);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pa = a[k], pb = bm[k], pc = cv[k];
                geom::adam_update(pa, r[k], pb, pc, adam.b1, adam.b2, adam.eps, 1.f, as.step_size, as.bc2_sqrt);
                a[k] = pa, bm[k] = pb, cv[k] = pc;
            }
            *reinterpret_cast<f32x4 *>(dst) = r;
            *m4 = bm, *v4 = cv, *p4 = a;
        } else {
            dst[0] = r[0], dst[1] = r[1], dst[2] = r[2], dst[3] = r[3];
            if (pp) {
                float *pm = adam.m[jb] + at, *pv = adam.v[jb] + at;
                pp += at;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a = pp[k], b = pm[k], c = pv[k];
                    geom::adam_update(a, r[k], b, c, adam.b1, adam.b2, adam.eps, 1.f, as.step_size, as.bc2_sqrt);
                    pm[k] = b, pv[k] = c, pp[k] = a;
                }
            }
        }
    }
    if (state) geom::adam_arrive(state, as, block, nblocks);
}

__global__ __launch_bounds__(RED_THREADS) void dense_reduce_kernel(ReduceJobs jobs, ReduceAdam adam)
{
    __shared__ f32x4 part[RED_THREADS];
    __shared__ float st[3];
    reduce_body<true>(jobs, adam, blockIdx.x, gridDim.x, part, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------

int num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// rows-kernel geometry: RB row-blocks per tile such that the busiest workgroup does the least work
struct RowGeo {
    int rb, n_tiles, left_rb, grid;
};
RowGeo row_geometry(int rows, int ncw, int cus)
{
    const int n_rb = (rows + 15) / 16;
    RowGeo best{0, 0, 0, 0};
    long best_cost = -1;
    const int cand[] = {5, 6, 4, 3, 2, 1};
    for (int rb : cand) {
        int n_tiles = n_rb / rb, left = n_rb % rb;
        if (left * ncw > cus || (n_tiles == 0 && left == 0)) continue;
        const int grid = n_tiles < cus ? (n_tiles > left * ncw ? n_tiles : left * ncw) : cus;
        if (grid <= 0) continue;
        const int passes = (n_tiles + grid - 1) / grid;
        // cost in fragments per wave of the busiest workgroup: passes * rb * ncw (+1 with a leftover fragment)
        const long cost = (long)(passes > 0 ? passes : 1) * rb * ncw + (left ? 1 : 0);
        if (best_cost < 0 || cost < best_cost) best = RowGeo{rb, n_tiles, left, grid}, best_cost = cost;
    }
    return best;
}

template <int NCW, bool B_TC, bool A_VEC, int EPI>
int launch_rows_rb(const RowArgs &q, const RowGeo &geo, hipStream_t s)
{
    const dim3 grid(geo.grid), block(DG_THREADS);
    switch (geo.rb) {
    case 1: hipLaunchKernelGGL((dense_rows_kernel<1, NCW, B_TC, A_VEC, EPI>), grid, block, 0, s, q); break;
    case 2: hipLaunchKernelGGL((dense_rows_kernel<2, NCW, B_TC, A_VEC, EPI>), grid, block, 0, s, q); break;
    case 3: hipLaunchKernelGGL((dense_rows_kernel<3, NCW, B_TC, A_VEC, EPI>), grid, block, 0, s, q); break;
    case 4: hipLaunchKernelGGL((dense_rows_kernel<4, NCW, B_TC, A_VEC, EPI>), grid, block, 0, s, q); break;
    case 5: hipLaunchKernelGGL((dense_rows_kernel<5, NCW, B_TC, A_VEC, EPI>), grid, block, 0, s, q); break;
    case 6: hipLaunchKernelGGL((dense_rows_kernel<6, NCW, B_TC, A_VEC, EPI>), grid, block, 0, s, q); break;
    default: return GEOM_EINVAL;
    }
    return geom::launch_status();
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

} // namespace

// support = x . w (+ the 0N-GCN pass-through epilogue)
extern "C" int geom_dense_fwd_f32(int rows, int cin, int c, const float *x, const float *w, int ksplit, const float *bias,
                                  float *out, float *sup, uint16_t *mask, void *stream)
{
    if (rows < 0 || cin <= 0 || c <= 0) return GEOM_EINVAL;
    if (c % 16 != 0 || c > 192 || (int64_t)rows * cin >= (1LL << 30)) return GEOM_EUNSUPPORTED;
    if (rows == 0) return 0;
    if (!x || !w || !out || !aligned16(w) || !aligned16(out)) return GEOM_EINVAL;
    const bool zn = ksplit > 0;
    if (zn && (ksplit % 16 != 0 || ksplit >= c || !sup || !aligned16(sup) || (bias && !aligned16(bias)))) return GEOM_EINVAL;
    const RowGeo geo = row_geometry(rows, 3, num_cus());
    RowArgs q{x, cin, w, c, out, c, rows, c, cin, geo.n_tiles, geo.left_rb, 1, zn ? ksplit : 0, sup, bias, mask};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool avec = cin % 4 == 0 && aligned16(x);
    if (zn) return avec ? launch_rows_rb<3, false, true, EPI_ZN>(q, geo, s) : launch_rows_rb<3, false, false, EPI_ZN>(q, geo, s);
    return avec ? launch_rows_rb<3, false, true, EPI_PLAIN>(q, geo, s) : launch_rows_rb<3, false, false, EPI_PLAIN>(q, geo, s);
}

// grad_x = g . w^T   (g [rows, c], w [cin, c], grad_x [rows, cin])
extern "C" int geom_dense_bwd_input_f32(int rows, int cin, int c, const float *g, const float *w, float *grad_x, void *stream)
{
    if (rows < 0 || cin <= 0 || c <= 0) return GEOM_EINVAL;
    if (c % 4 != 0 || (int64_t)rows * (cin > c ? cin : c) >= (1LL << 30)) return GEOM_EUNSUPPORTED;
    if (rows == 0) return 0;
    if (!g || !w || !grad_x || !aligned16(g) || !aligned16(w)) return GEOM_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ncw = cin > 192 ? 4 : 3;
    const RowGeo geo = row_geometry(rows, ncw, num_cus());
    RowArgs q{g, c, w, c, grad_x, cin, rows, cin, c, geo.n_tiles, geo.left_rb, (cin + ncw * 64 - 1) / (ncw * 64), 0, nullptr,
              nullptr, nullptr};
    return ncw == 4 ? launch_rows_rb<4, true, true, EPI_PLAIN>(q, geo, s) : launch_rows_rb<3, true, true, EPI_PLAIN>(q, geo, s);
}

namespace {
struct SplitGeo {
    int rb, full_tiles, left_rb, s_full, s_left, slots;
};
// Tile height of a weight gradient's partial sums.  The workgroup count is fixed by the chip (one per CU), so tiles x splits
// ~ 256 and the bytes of partial tiles written here and read back by the reduction launch are splits x (cin x c x 4): the
// MORE output tiles, the FEWER splits of the summed rows and the less traffic.  A 192-wide layer: 2 tiles of 96 rows x 128
// splits = 18.9 MB, 4 tiles of 48 rows x 64 splits = 9.4 MB for the same MFMA work per workgroup (a stage is then 72 MFMAs
// per wave instead of 144: still enough shadow for the X panel and the G loads, measured profiles/r04_split_tiles.txt).
// The 963-wide layer keeps 96-row tiles (25 splits already): with 48-row tiles x 12 splits its launch -- the step's
// longest -- went from 65.7 to 71.4 us for 1.2 us less in the reduction launch.
inline int split_rb(int cin) { return cin <= 192 ? SPLIT_RB_NARROW : SPLIT_RB; }
SplitGeo split_geometry(int cin, int rows, int cus)
{
    SplitGeo g;
    g.rb = split_rb(cin);
    const int ra = g.rb * 16;
    g.full_tiles = cin / ra;
    g.left_rb = (cin % ra + 15) / 16;
    const int n4 = (rows + 3) / 4;
    // a leftover row-block is 1/rb of a full tile: equal work per workgroup <=> s_left = s_full / rb
    g.s_full = g.full_tiles ? (int)((int64_t)cus * g.rb / (g.full_tiles * g.rb + g.left_rb)) : 0;
    if (g.full_tiles && g.s_full < 1) g.s_full = 1;
    g.s_left = 0;
    if (g.left_rb) {
        const int rest = cus - g.full_tiles * g.s_full;
        g.s_left = rest / g.left_rb > 0 ? rest / g.left_rb : 1;
    }
    if (g.s_full > n4) g.s_full = n4;
    if (g.s_left > n4) g.s_left = n4;
    g.slots = g.full_tiles * g.s_full + g.left_rb * g.s_left;
    return g;
}
} // namespace

extern "C" int64_t geom_dense_bwd_weight_workspace_floats(int rows, int cin, int c)
{
    if (rows <= 0 || cin <= 0 || c <= 0 || c > 192) return 0;
    const SplitGeo g = split_geometry(cin, rows, num_cus());
    return (int64_t)g.slots * g.rb * 16 * 192 + (int64_t)(g.s_full > 0 ? g.s_full : g.s_left) * c;
}

// partial sums of grad_w = x^T . g into `workspace`; geom_dense_reduce_f32 finishes them (and the bias gradient)
extern "C" int geom_dense_bwd_weight_f32(int rows, int cin, int c, const float *x, const float *g, float *workspace,
                                         int want_colsum, void *stream)
{
    if (rows <= 0 || cin <= 0 || c <= 0) return GEOM_EINVAL;
    // c % 12: a lane owns whole 3-column groups of G (12-byte loads) and whole 4-column groups of the partial tile
    if (c % 12 != 0 || c > 192 || (int64_t)rows * (cin > c ? cin : c) >= (1LL << 30)) return GEOM_EUNSUPPORTED;
    if (!x || !g || !workspace || !aligned16(g) || !aligned16(workspace)) return GEOM_EINVAL;
    const SplitGeo geo = split_geometry(cin, rows, num_cus());
    if (want_colsum && geo.full_tiles == 0) return GEOM_EUNSUPPORTED;
    float *colsum = want_colsum ? workspace + (int64_t)geo.slots * geo.rb * 16 * 192 : nullptr;
    SplitArgs q{x, cin, g, c, workspace, cin, c, rows, geo.full_tiles, geo.s_full, geo.left_rb, geo.s_left, colsum};
    const dim3 grid(geo.slots);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool xvec = cin % 4 == 0 && aligned16(x);
    if (geo.rb == SPLIT_RB_NARROW) {
        const dim3 block(DG_THREADS);
        if (xvec) hipLaunchKernelGGL((dense_split_kernel<SPLIT_RB_NARROW, 3, true, 1>), grid, block, 0, s, q);
        else hipLaunchKernelGGL((dense_split_kernel<SPLIT_RB_NARROW, 3, false, 1>), grid, block, 0, s, q);
    } else {
        const dim3 block(DG_THREADS * SPLIT_KP_WIDE);
        if (xvec) hipLaunchKernelGGL((dense_split_kernel<SPLIT_RB, 3, true, SPLIT_KP_WIDE>), grid, block, 0, s, q);
        else hipLaunchKernelGGL((dense_split_kernel<SPLIT_RB, 3, false, SPLIT_KP_WIDE>), grid, block, 0, s, q);
    }
    return geom::launch_status();
}

// Both gradients of one layer: grad_x = g . w^T and the partial sums of grad_w = x^T . g (finished by geom_dense_reduce_f32).
// One launch with two workgroups per CU (dense_bwd_pair_kernel) when both halves fit that mode: cin <= 192 (one column chunk
// for the input gradient) and 16-byte aligned x rows; two launches otherwise (the 963-wide first layer).
extern "C" int geom_dense_bwd_f32(int rows, int cin, int c, const float *x, const float *g, const float *w, float *grad_x,
                                  float *workspace, int want_colsum, void *stream)
{
    if (rows <= 0 || cin <= 0 || c <= 0) return GEOM_EINVAL;
    if (c % 12 != 0 || c > 192 || (int64_t)rows * (cin > c ? cin : c) >= (1LL << 30)) return GEOM_EUNSUPPORTED;
    if (!x || !g || !w || !grad_x || !workspace || !aligned16(g) || !aligned16(w) || !aligned16(workspace)) return GEOM_EINVAL;
    const int cus = num_cus();
    const bool xvec = cin % 4 == 0 && aligned16(x);
    const RowGeo rg = row_geometry(rows, 3, cus);
    const SplitGeo sg = split_geometry(cin, rows, cus);
    if (cin > 192 || !xvec || (want_colsum && sg.full_tiles == 0) || rg.rb < 2 || rg.rb > 5) { // rb 6: 82 KB of LDS, no pairing
        int code = geom_dense_bwd_input_f32(rows, cin, c, g, w, grad_x, stream);
        if (code) return code;
        return geom_dense_bwd_weight_f32(rows, cin, c, x, g, workspace, want_colsum, stream);
    }
    RowArgs r{g, c, w, c, grad_x, cin, rows, cin, c, rg.n_tiles, rg.left_rb, 1, 0, nullptr, nullptr, nullptr};
    float *colsum = want_colsum ? workspace + (int64_t)sg.slots * sg.rb * 16 * 192 : nullptr;
    SplitArgs q{x, cin, g, c, workspace, cin, c, rows, sg.full_tiles, sg.s_full, sg.left_rb, sg.s_left, colsum};
    const dim3 grid(rg.grid + sg.slots), block(DG_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    static_assert(SPLIT_RB_NARROW == 3, "cin <= 192 (the pair mode's range) takes the narrow split tiles");
    switch (rg.rb) {
    case 2: hipLaunchKernelGGL((dense_bwd_pair_kernel<2, SPLIT_RB_NARROW, true>), grid, block, 0, s, r, q, rg.grid); break;
    case 3: hipLaunchKernelGGL((dense_bwd_pair_kernel<3, SPLIT_RB_NARROW, true>), grid, block, 0, s, r, q, rg.grid); break;
    case 4: hipLaunchKernelGGL((dense_bwd_pair_kernel<4, SPLIT_RB_NARROW, true>), grid, block, 0, s, r, q, rg.grid); break;
    case 5: hipLaunchKernelGGL((dense_bwd_pair_kernel<5, SPLIT_RB_NARROW, true>), grid, block, 0, s, r, q, rg.grid); break;
    default: return GEOM_EINVAL;
    }
    return geom::launch_status();
}

namespace {
// flat grid of dense_reduce_kernel: only workgroups that hold outputs are launched
void flat_grid(ReduceJobs &jobs, int n)
{
    jobs.n = n;
    jobs.first[0] = 0;
    for (int j = 0; j < n; ++j) {
        ReduceJob &q = jobs.job[j];
        q.outs = reduce_outs(q.s_full > q.s_left ? q.s_full : q.s_left);
        jobs.first[j + 1] = jobs.first[j] + (q.I * (q.J >> 2) + q.outs - 1) / q.outs;
    }
}

// the jobs of geom_dense_reduce2_f32 / geom_dense_reduce_adam_f32: weight gradients first (job l = layer l), then the
// column-sum jobs (job count + i)
int build_reduce_jobs(ReduceJobs &jobs, int &n, int &widest, int count, const int *rows, const int *cin, const int *c,
                      const float *const *workspaces, float *const *grad_w, int ncs, const float *const *cs_partials,
                      const int *cs_rows, const int *cs_cols, float *const *cs_outs)
{
    if (count < 0 || ncs < 0) return GEOM_EINVAL;
    if (count && (!rows || !cin || !c || !workspaces || !grad_w)) return GEOM_EINVAL;
    if (ncs && (!cs_partials || !cs_rows || !cs_cols || !cs_outs)) return GEOM_EINVAL;
    if (count + ncs > GEOM_DENSE_MAX_REDUCE_JOBS) return GEOM_ETOOBIG;
    n = 0, widest = 0;
    for (int l = 0; l < count; ++l) {
        if (!workspaces[l] || !grad_w[l] || rows[l] <= 0 || cin[l] <= 0 || c[l] <= 0 || c[l] > 192 || c[l] % 4) return GEOM_EINVAL;
        const SplitGeo g = split_geometry(cin[l], rows[l], num_cus());
        jobs.job[n++] = ReduceJob{workspaces[l], grad_w[l], cin[l], c[l], g.rb * 16, 192, g.full_tiles, g.s_full, g.s_left, 0};
        widest = cin[l] * (c[l] / 4) > widest ? cin[l] * (c[l] / 4) : widest;
    }
    for (int i = 0; i < ncs; ++i) {
        if (!cs_partials[i] || !cs_outs[i] || cs_rows[i] < 0 || cs_cols[i] <= 0 || cs_cols[i] % 4) return GEOM_EINVAL;
        jobs.job[n++] = ReduceJob{cs_partials[i], cs_outs[i], 1, cs_cols[i], 1, cs_cols[i], 1, cs_rows[i], 0, 0};
    }
    flat_grid(jobs, n);
    return 0;
}
} // namespace

// Weight gradients of `count` layers out of their workspaces AND `ncs` pending column-sum jobs (cs_outs[i][0..cs_cols[i]) =
// column sums of cs_partials[i], cs_rows[i] x cs_cols[i] row-major: the per-workgroup bias-gradient partials of the
// aggregation backward, what geom_colsum_batch_f32 does in a launch of its own) in ONE launch.  Host arrays; count + ncs
// jobs in total <= GEOM_DENSE_MAX_REDUCE_JOBS.
extern "C" int geom_dense_reduce2_f32(int count, const int *rows, const int *cin, const int *c, const float *const *workspaces,
                                      float *const *grad_w, float *const *grad_bias, int ncs, const float *const *cs_partials,
                                      const int *cs_rows, const int *cs_cols, float *const *cs_outs, void *stream)
{
    if (grad_bias) return GEOM_EINVAL; // bias gradients travel as column-sum jobs here
    if (count + ncs == 0) return 0;
    ReduceJobs jobs;
    int n, widest;
    const int code = build_reduce_jobs(jobs, n, widest, count, rows, cin, c, workspaces, grad_w, ncs, cs_partials, cs_rows, cs_cols, cs_outs);
    if (code) return code;
    ReduceAdam adam{};
    hipLaunchKernelGGL(dense_reduce_kernel, dim3(jobs.first[n]), dim3(RED_THREADS), 0, static_cast<hipStream_t>(stream), jobs, adam);
    return geom::launch_status();
}

// The same launch + the Adam step of the parameters whose gradients it finishes: w_p/w_m/w_v[l] = parameter, first and
// second moment of layer l's weight (same [cin, c] layout as grad_w[l]), b_p/b_m/b_v[i] likewise for column-sum job i
// (entries may be NULL: gradient only).  `state` = the optimiser's device-side step state (geom_adam_step_f32); it is
// advanced once by this launch.  grad_scale is 1 (single-process step; a data-parallel step reduces the gradients across
// ranks first and uses geom_adam_step_f32).
extern "C" int geom_dense_reduce_adam_f32(int count, const int *rows, const int *cin, const int *c,
                                          const float *const *workspaces, float *const *grad_w, float *const *w_p,
                                          float *const *w_m, float *const *w_v, int ncs, const float *const *cs_partials,
                                          const int *cs_rows, const int *cs_cols, float *const *cs_outs, float *const *b_p,
                                          float *const *b_m, float *const *b_v, float lr, float beta1, float beta2, float eps,
                                          float *state, void *stream)
{
    if (count + ncs == 0 || !state) return GEOM_EINVAL;
    if ((count && (!w_p || !w_m || !w_v)) || (ncs && (!b_p || !b_m || !b_v))) return GEOM_EINVAL;
    ReduceJobs jobs;
    int n, widest;
    const int code = build_reduce_jobs(jobs, n, widest, count, rows, cin, c, workspaces, grad_w, ncs, cs_partials, cs_rows, cs_cols, cs_outs);
    if (code) return code;
    ReduceAdam adam{};
    for (int l = 0; l < count; ++l) {
        adam.p[l] = w_p[l], adam.m[l] = w_m[l], adam.v[l] = w_v[l];
        if (w_p[l] && (!w_m[l] || !w_v[l])) return GEOM_EINVAL;
    }
    for (int i = 0; i < ncs; ++i) {
        adam.p[count + i] = b_p[i], adam.m[count + i] = b_m[i], adam.v[count + i] = b_v[i];
        if (b_p[i] && (!b_m[i] || !b_v[i])) return GEOM_EINVAL;
    }
    adam.lr = lr, adam.b1 = beta1, adam.b2 = beta2, adam.eps = eps, adam.state = state;
    uintptr_t bits = 0;
    for (int j = 0; j < n; ++j)
        if (adam.p[j]) bits |= (uintptr_t)adam.p[j] | (uintptr_t)adam.m[j] | (uintptr_t)adam.v[j] | (uintptr_t)jobs.job[j].out;
    adam.vec = (bits & 15) == 0;
    hipLaunchKernelGGL(dense_reduce_kernel, dim3(jobs.first[n]), dim3(RED_THREADS), 0, static_cast<hipStream_t>(stream), jobs, adam);
    return geom::launch_status();
}

// grad_w[i] (and grad_bias[i], may be NULL) of `count` layers out of their workspaces, ONE launch
extern "C" int geom_dense_reduce_f32(int count, const int *rows, const int *cin, const int *c, const float *const *workspaces,
                                     float *const *grad_w, float *const *grad_bias, void *stream)
{
    if (count < 0 || count > GEOM_DENSE_MAX_REDUCE_JOBS / 2) return GEOM_ETOOBIG;
    if (count == 0) return 0;
    if (!rows || !cin || !c || !workspaces || !grad_w) return GEOM_EINVAL;
    ReduceJobs jobs;
    int n = 0, widest = 0;
    for (int l = 0; l < count; ++l) {
        if (!workspaces[l] || !grad_w[l] || rows[l] <= 0 || cin[l] <= 0 || c[l] <= 0 || c[l] > 192 || c[l] % 4) return GEOM_EINVAL;
        const SplitGeo g = split_geometry(cin[l], rows[l], num_cus());
        jobs.job[n++] = ReduceJob{workspaces[l], grad_w[l], cin[l], c[l], g.rb * 16, 192, g.full_tiles, g.s_full, g.s_left, 0};
        widest = cin[l] * (c[l] / 4) > widest ? cin[l] * (c[l] / 4) : widest;
        if (grad_bias && grad_bias[l]) { // column sums: a 1-row "tile" per split, pitch = c
            if (g.full_tiles == 0) return GEOM_EUNSUPPORTED;
            jobs.job[n++] = ReduceJob{workspaces[l] + (int64_t)g.slots * g.rb * 16 * 192, grad_bias[l], 1, c[l], 1, c[l], 1,
                                      g.s_full, 0, 0};
        }
    }
    flat_grid(jobs, n);
    hipLaunchKernelGGL(dense_reduce_kernel, dim3(jobs.first[n]), dim3(RED_THREADS), 0, static_cast<hipStream_t>(stream), jobs,
                       ReduceAdam{});
    return geom::launch_status();
}
