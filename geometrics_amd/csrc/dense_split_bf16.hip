// EXPERIMENT (round-5 review item 5; not on any default route): the first layer's forward product X . W ([rows, 963] x [963, 192])
// on the BF16 matrix cores with EXACT fp32 products.
//
// gfx950 has no fp32-rate shortcut (no xf32 / tf32): v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate, and the seven dense
// products are two thirds of the headline step.  An fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 mantissa
// bits: a = a0 + a1 + a2, a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)), and a product of two bf16 numbers is exact
// in fp32 (16 bits).  So a . b = sum_{i,j} a_i b_j with nine exact products; the three with i + j >= 3 are below 2^-24 of
// |a b| and are dropped (six MFMAs per tile and 32 k instead of 8 fp32 ones: 3/8 of the fp32 matrix-core time at most).
// ACCURACY (tools/probe/split_bf16_accuracy.py, an emulation in numpy; tests/test_split_bf16_gpu.py on the device): with the
// leading term a0 b0 in one fp32 accumulator and the five small terms in a second one, added once at the end, the result is
// CLOSER to the float64 product than the native fp32 MFMA chain on the same inputs (the big accumulator is rounded 31 times
// instead of 241), and the six-term sum is indistinguishable from the nine-term one.
//
// Shape of the kernel: one workgroup = 4 waves = a 96-row x 192-column output tile (213.5 -> 214 workgroups at 20 496 rows: one
// round on 256 CUs), wave (wy, wx) = 48 rows x 96 columns = 3 x 6 tiles of v_mfma_f32_16x16x32_bf16, two accumulators per
// tile.  Per 32-k block: the 96 x 32 fp32 piece of X is loaded with dword-aligned 16-byte buffer loads,
// split in registers and written as three bf16 planes to LDS; W comes pre-split and transposed ([plane][column][k], bf16:
// geom_split_bf16_planes_f32, once per weight update) so that a lane's 8 consecutive k are one 16-byte read; LDS rows are
// padded to 80 bytes (16 rows x 16 B cover all 64 banks); double-buffered, the next block's global loads travel under the
// current block's 108 MFMAs per wave.
#include "geom_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_THREADS = 256;
constexpr int SB_N = 192;          // output columns (the layer width)
constexpr int SB_TM = 96;          // rows per workgroup
constexpr int SB_KB = 32;          // k per block = one MFMA
constexpr int SB_PITCH = 80;       // bytes per LDS row of 32 bf16 (64 used)
constexpr int SB_A_PLANE = SB_TM * SB_PITCH;   // 7 680 B
constexpr int SB_B_PLANE = SB_N * SB_PITCH;    // 15 360 B
constexpr int SB_BUF = 3 * SB_A_PLANE + 3 * SB_B_PLANE; // 69 120 B per stage

__device__ __forceinline__ unsigned bf16_bits(float x) // round to nearest even (finite inputs)
{
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_value(unsigned bits) { return __uint_as_float(bits << 16); }

// a -> (a0, a1, a2) as bf16 bit patterns, a0 + a1 + a2 == a exactly
__device__ __forceinline__ void split3(float a, unsigned &b0, unsigned &b1, unsigned &b2)
{
    b0 = bf16_bits(a);
    const float r1 = a - bf16_value(b0);
    b1 = bf16_bits(r1);
    const float r2 = r1 - bf16_value(b1);
    b2 = bf16_bits(r2);
}

// planes [3][n][kpad] bf16 (kpad = k rounded up to 32, zero filled) of w [k][n] row-major: thread = one (column, k)
__global__ __launch_bounds__(256) void split_planes_kernel(int k, int n, int kpad, const float *w, unsigned short *planes)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * kpad) return;
    const int col = (int)(i / kpad), kk = (int)(i - (int64_t)col * kpad);
    unsigned b0 = 0, b1 = 0, b2 = 0;
    if (kk < k) split3(w[(size_t)kk * n + col], b0, b1, b2);
    planes[i] = (unsigned short)b0;
    planes[(size_t)n * kpad + i] = (unsigned short)b1;
    planes[2 * (size_t)n * kpad + i] = (unsigned short)b2;
}

struct SbArgs {
    const float *a;             // [m][k] row-major fp32
    const unsigned short *bt;   // [3][192][kpad] bf16
    float *c;                   // [m][192]
    int m, k, kpad, terms;      // terms: 6 (i + j <= 2) or 9
};

__global__ __launch_bounds__(SB_THREADS, 1) void split_bf16_gemm_kernel(SbArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wy = wave >> 1, wx = wave & 1;       // wave tile: rows 48 wy .., columns 96 wx ..
    const int lr = lane & 15, lg = lane >> 4;      // fragment coordinates: row / column lr, k group lg (8 consecutive k)
    const int row0 = blockIdx.x * SB_TM;
    const int nkb = p.kpad / SB_KB;

    // ---- global -> register staging of one k-block (buffer loads: 32-bit offsets against wave-uniform descriptors, rows beyond
    // the matrix read zeros; the offsets are formed once and advance by a constant per block)
    // X: thread (r = tid / 8, k4 = tid % 8) owns k = 4 k4 .. 4 k4 + 3 of rows r, r + 32, r + 64: three 16-byte loads (a
    // multi-dword buffer load needs dword alignment only: the 963-float rows are 4-byte aligned), eight lanes per 128-byte row piece
    const int ar = tid >> 3, ak4 = tid & 7;
    const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.a), 0, (int)((int64_t)p.m * p.k * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.bt), 0, (int)((int64_t)3 * SB_N * p.kpad * 2), 0x00020000);
    unsigned a_off[3], b_off[9], b_lds[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = row0 + ar + 32 * i;
        a_off[i] = row < p.m ? (unsigned)(((int64_t)row * p.k + 4 * ak4) * 4) : 0x80000000u;
    }
    // W planes: 3 x 192 x 64 B per block = 2 304 chunks of 16 B, nine per thread: chunk q = tid + 256 t -> (plane, column, piece)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int q = tid + SB_THREADS * t, plane = q / 768, rem = q - plane * 768, col = rem >> 2, piece = rem & 3;
        b_off[t] = (unsigned)((((int64_t)plane * SB_N + col) * p.kpad + 8 * piece) * 2);
        b_lds[t] = 3 * SB_A_PLANE + plane * SB_B_PLANE + col * SB_PITCH + 16 * piece;
    }
    u32x4 av[3];
    u32x4 bv[9];
    auto load_block = [&](int kb) {
        const int kk = kb * SB_KB + 4 * ak4;
        const bool whole = kk + 3 < p.k; // (only the last block has a tail: 963 = 30 x 32 + 3)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned off = a_off[i] + (unsigned)kb * (SB_KB * 4);
            if (whole) {
                av[i] = __builtin_amdgcn_raw_buffer_load_b128(r_a, off, 0, 0);
            } else { // the tail: element by element (a 16-byte load would run into the next row)
                av[i].x = kk + 0 < p.k ? __builtin_amdgcn_raw_buffer_load_b32(r_a, off, 0, 0) : 0u;
                av[i].y = kk + 1 < p.k ? __builtin_amdgcn_raw_buffer_load_b32(r_a, off + 4, 0, 0) : 0u;
                av[i].z = kk + 2 < p.k ? __builtin_amdgcn_raw_buffer_load_b32(r_a, off + 8, 0, 0) : 0u;
                av[i].w = 0u;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) bv[t] = __builtin_amdgcn_raw_buffer_load_b128(r_b, b_off[t] + (unsigned)kb * (SB_KB * 2), 0, 0);
    };
    auto store_block = [&](unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            unsigned e[4][3];
            split3(__uint_as_float(av[i].x), e[0][0], e[0][1], e[0][2]);
            split3(__uint_as_float(av[i].y), e[1][0], e[1][1], e[1][2]);
            split3(__uint_as_float(av[i].z), e[2][0], e[2][1], e[2][2]);
            split3(__uint_as_float(av[i].w), e[3][0], e[3][1], e[3][2]);
            unsigned char *dst = buf + (ar + 32 * i) * SB_PITCH + 8 * ak4;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint2 *>(dst + pl * SB_A_PLANE) = make_uint2(e[0][pl] | (e[1][pl] << 16), e[2][pl] | (e[3][pl] << 16));
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) *reinterpret_cast<u32x4 *>(buf + b_lds[t]) = bv[t];
    };

    f32x4 hi[3][6], lo[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) hi[r][c] = lo[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_block(0);
    store_block(lds);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        unsigned char *cur = lds + (kb & 1) * SB_BUF, *nxt = lds + ((kb + 1) & 1) * SB_BUF;
        if (kb + 1 < nkb) load_block(kb + 1); // in flight under this block's MFMAs
        // ---- this block's products: A fragments of the wave's 3 row tiles x 3 planes, then column tile by column tile with the
        // NEXT tile's W fragments requested in front of this tile's MFMAs; per term the three row tiles in turn, so that an
        // accumulator is touched every third MFMA (a dependent MFMA waits for its predecessor's passes)
        bf16x8 af[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[r][pl] = *reinterpret_cast<const bf16x8 *>(cur + pl * SB_A_PLANE + (48 * wy + 16 * r + lr) * SB_PITCH + 16 * lg);
        const unsigned char *bbase = cur + 3 * SB_A_PLANE + (96 * wx + lr) * SB_PITCH + 16 * lg;
        bf16x8 bf[3], bn[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bf16x8 *>(bbase + pl * SB_B_PLANE);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (c + 1 < 6) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bn[pl] = *reinterpret_cast<const bf16x8 *>(bbase + pl * SB_B_PLANE + 16 * (c + 1) * SB_PITCH);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) hi[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][0], bf[0], hi[r][c], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][0], bf[1], lo[r][c], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][1], bf[0], lo[r][c], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][1], bf[1], lo[r][c], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][0], bf[2], lo[r][c], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][2], bf[0], lo[r][c], 0, 0, 0);
            if (p.terms == 9) {
#pragma unroll
                for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][1], bf[2], lo[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][2], bf[1], lo[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 3; ++r) lo[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][2], bf[2], lo[r][c], 0, 0, 0);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[pl] = bn[pl];
        }
        if (kb + 1 < nkb) store_block(nxt);
        __syncthreads();
    }
    // ---- C = hi + lo; D layout of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = row0 + 48 * wy + 16 * r + 4 * lg + e, col = 96 * wx + 16 * c + lr;
                if (row < p.m) p.c[(size_t)row * SB_N + col] = hi[r][c][e] + lo[r][c][e];
            }
}

} // namespace

// planes [3][n][kpad] bf16 (kpad = k rounded up to a multiple of 32; geom_split_bf16_kpad) <- w [k][n] fp32: w == plane 0 +
// plane 1 + plane 2 exactly, transposed so that 8 consecutive k of a column are 16 contiguous bytes.
extern "C" int geom_split_bf16_kpad(int k) { return (k + 31) & ~31; }
extern "C" int geom_split_bf16_planes_f32(int k, int n, const float *w, uint16_t *planes, void *stream)
{
    if (k <= 0 || n <= 0) return GEOM_EINVAL;
    if (!w || !planes || ((uintptr_t)planes & 15)) return GEOM_EINVAL;
    const int kpad = geom_split_bf16_kpad(k);
    const int64_t total = (int64_t)n * kpad;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), k, n, kpad, w,
                       reinterpret_cast<unsigned short *>(planes));
    return geom::launch_status();
}

// c [m, 192] = a [m, k] . w [k, 192] with w given as planes (geom_split_bf16_planes_f32): exact fp32 products on the bf16 matrix
// cores, fp32 accumulation (two accumulators per element); terms = 6 (products with i + j <= 2) or 9.  n must be 192.
extern "C" int geom_gemm_split_bf16_f32(int m, int k, int n, const float *a, const uint16_t *planes, float *c, int terms, void *stream)
{
    if (m < 0 || k <= 0) return GEOM_EINVAL;
    if (n != SB_N || (terms != 6 && terms != 9)) return GEOM_EUNSUPPORTED;
    if (m == 0) return 0;
    if (!a || !planes || !c || ((uintptr_t)planes & 15) || ((uintptr_t)a & 3) || ((uintptr_t)c & 3)) return GEOM_EINVAL;
    if ((int64_t)m * k >= (1LL << 31)) return GEOM_ETOOBIG;
    SbArgs p{a, reinterpret_cast<const unsigned short *>(planes), c, m, k, geom_split_bf16_kpad(k), terms};
    static bool configured[64];
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !configured[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(split_bf16_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SB_BUF) != hipSuccess)
            return GEOM_EINVAL;
        configured[dev] = true;
    }
    hipLaunchKernelGGL(split_bf16_gemm_kernel, dim3((m + SB_TM - 1) / SB_TM), dim3(SB_THREADS), 2 * SB_BUF, static_cast<hipStream_t>(stream), p);
    return geom::launch_status();
}
