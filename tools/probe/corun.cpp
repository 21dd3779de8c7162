// What does a memory / VALU / LDS wave cost (and get) beside an MFMA wave on the SAME SIMD?  One workgroup of 8 waves per CU:
// waves 0-3 stream v_mfma_f32_16x16x4_f32 (3 accumulators in rotation), waves 4-7 run one of several instruction mixes.
// Each role alone and both together; per role the median wave duration in shader ticks.      hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MIX, int NOPS>
__global__ __launch_bounds__(512, 2) void corun(int n_mfma, int n_other, float *buf, unsigned long long *stamps, int run_mfma, int run_other, int prio)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    lds[tid] = tid, lds[tid + 512] = 1.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (run_mfma) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
            float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
            for (int i = 0; i < n_mfma; ++i) {
                if (NOPS == 100) { // v_mfma_f32_32x32x2_f32: 64 cycles each, the same flop rate -- does IT leave issue slots?
                    typedef float f32x16 __attribute__((ext_vector_type(16)));
                    static f32x16 c0, c1;
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
                    }
                    if (i == n_mfma - 1 && c0[0] + c1[1] == 12345.f) buf[tid] = c0[0];
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    // NOPS idle cycles behind every MFMA: the wave does not present its next MFMA while the pipe is busy anyway
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    if (NOPS == 3) asm volatile("s_nop 3"); if (NOPS == 4) asm volatile("s_nop 4"); if (NOPS == 5) asm volatile("s_nop 5"); if (NOPS == 6) asm volatile("s_nop 6"); if (NOPS == 7) asm volatile("s_nop 7");
                    
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                    if (NOPS == 3) asm volatile("s_nop 3"); if (NOPS == 4) asm volatile("s_nop 4"); if (NOPS == 5) asm volatile("s_nop 5"); if (NOPS == 6) asm volatile("s_nop 6"); if (NOPS == 7) asm volatile("s_nop 7");
                    
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                    if (NOPS == 3) asm volatile("s_nop 3"); if (NOPS == 4) asm volatile("s_nop 4"); if (NOPS == 5) asm volatile("s_nop 5"); if (NOPS == 6) asm volatile("s_nop 6"); if (NOPS == 7) asm volatile("s_nop 7");
                    
                }
            }
            if (a0[0] + a1[1] + a2[2] == 12345.f) buf[tid] = a0[0];
        }
    } else if (run_other) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 24, 0x00020000);
        const unsigned off = (blockIdx.x * 4096 + (tid - 256) * 16) & ((1 << 24) - 1);
        float acc = lane;
        u32x4 s = {0, 0, 0, 0};
        for (int i = 0; i < n_other; ++i) {
            if (MIX == 1) { // 64 independent-ish VALU
#pragma unroll
                for (int k = 0; k < 64; ++k) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
            } else if (MIX == 2) { // 8 buffer loads (L2-resident), one wait
                u32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, off + k * 65536 + (i & 3) * 4096, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            } else if (MIX == 3) { // 8 LDS reads
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<f32x4 *>(lds + ((tid * 4 + k * 1024 + i * 4) & 8188 & ~3));
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k][0];
            } else if (MIX == 4) { // 8 buffer stores
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){(unsigned)i, 1u, 2u, 3u}, r, off + k * 65536 + (i & 3) * 4096 + (8 << 20), 0, 0);
            }
        }
        if (acc + s.x == 12345.f) buf[tid] = acc;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MIX, int NOPS>
void run(const char *name, int n_mfma, int n_other, float *buf, unsigned long long *dst)
{
    for (int mode = 0; mode < 4; ++mode) { // 0: mfma alone, 1: other alone, 2: both, 3: both + prio
        const int rm = mode != 1, ro = mode != 0;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((corun<MIX, NOPS>), dim3(256), dim3(512), 0, 0, n_mfma, n_other, buf, dst, rm, ro, mode == 3);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), dst, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> m, o;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? m : o).push_back(h[b * 8 + w]);
        std::sort(m.begin(), m.end()), std::sort(o.begin(), o.end());
        const char *modes[] = {"mfma alone ", "other alone", "both       ", "both + prio"};
        printf("%-32s %s   mfma waves %8llu ticks   other waves %8llu ticks\n", name, modes[mode], m[m.size() / 2], o[o.size() / 2]);
    }
}

int main()
{
    float *buf;
    unsigned long long *st;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 0, 64 << 20);
    hipMalloc(&st, 256 * 8 * 8);
    // 100 x 48 MFMAs = 4800 MFMAs ~ 153 600 cycles
    // 100 x 48 MFMAs = 4800 MFMAs = 153 600 cycles at 32 per MFMA
    run<1, 0>("64 VALU x 600", 100, 600, buf, st);
    run<2, 0>("8 buffer loads x 200", 100, 200, buf, st);
    run<3, 0>("8 LDS reads x 600", 100, 600, buf, st);
    run<4, 0>("8 buffer stores x 100", 100, 100, buf, st);
    run<1, 100>("64 VALU x 600 | 32x32x2", 100, 600, buf, st);
    run<2, 100>("8 buffer loads x 200 | 32x32x2", 100, 200, buf, st);
    run<1, 3>("64 VALU x 600 | s_nop 3", 100, 600, buf, st);
    run<1, 5>("64 VALU x 600 | s_nop 5", 100, 600, buf, st);
    run<1, 7>("64 VALU x 600 | s_nop 7", 100, 600, buf, st);
    run<2, 5>("8 buffer loads x 200 | s_nop 5", 100, 200, buf, st);
    return 0;
}
