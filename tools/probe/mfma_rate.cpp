// What does the fp32 matrix pipe deliver, and at which clock?  A kernel of NOTHING but v_mfma_f32_16x16x4_f32 on independent
// accumulators (no memory, no LDS): every SIMD of every CU issues MFMAs back to back, so
//     TFLOP/s = CUs x 4 SIMDs x 64 flop/clk x f      and      f = (MFMAs per wave x 32 cycles) / kernel time.
// Swept over the LENGTH of the busy period (one launch of 0.1 ms .. back-to-back launches for ~1 s): the guide's 155 TFLOP/s
// (MI355X_MICROARCH.md:41,389) is what a short burst on an idle chip reaches; under sustained fp32-MFMA load the power
// management settles lower -- the figure the step's dense kernels have to be priced against.  tools/probe/mfma_rate.sh
// samples the reported sclk next to it.     hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate.cpp -o tools/probe/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_only(float *out, int iters, float a, float b)
{
    f32x4 acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < ACC; ++i) s += acc[i];
    if (s[0] == 12345.678f) out[threadIdx.x] = s[1] + s[2] + s[3];   // never true: keeps the chain alive
}

int main(int argc, char **argv)
{
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    cus = prop.multiProcessorCount;
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    constexpr int ACC = 16;                      // 16 independent accumulators per wave: no dependent-MFMA stall
    printf("# %s, %d CUs, nominal clock %.0f MHz; one 4-wave workgroup per CU (one wave per SIMD), %d accumulators per wave\n",
           prop.name, cus, prop.clockRate / 1e3, ACC);
    printf("# busy period   launches   ms total   TFLOP/s   implied shader clock (GHz)\n");
    const double flop_per_iter = 8.0 * ACC * 2.0 * 16 * 16 * 4;       // per wave and loop iteration
    const int iters_short = 200;                                      // 200 x 128 MFMAs x 32 cycles = 0.8 M cycles ~ 0.35 ms
    mfma_only<ACC><<<cus, 256>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    struct { const char *name; int iters, launches; double idle_ms; } cases[] = {
        {"0.1 ms burst, idle before", 60, 1, 300}, {"0.35 ms burst, idle before", iters_short, 1, 300},
        {"3.5 ms", 2000, 1, 300}, {"35 ms", 2000, 10, 300}, {"0.35 s", 2000, 100, 300}, {"1 s", 2000, 300, 0},
        {"1 s (again, warm)", 2000, 300, 0}, {"0.35 ms burst right after", iters_short, 1, 0}};
    for (auto &c : cases) {
        if (c.idle_ms > 0) {
            hipDeviceSynchronize();
            struct timespec ts = {0, (long)(c.idle_ms * 1e6)};
            nanosleep(&ts, nullptr);
        }
        hipEventRecord(e0, nullptr);
        for (int l = 0; l < c.launches; ++l) mfma_only<ACC><<<cus, 256>>>(out, c.iters, 1.f, 1.f);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = flop_per_iter * c.iters * c.launches * 4.0 * cus;
        const double cycles = 8.0 * ACC * 32.0 * c.iters * c.launches;       // per SIMD, back-to-back issue
        printf("%-28s %5d   %9.3f   %7.1f   %.3f\n", c.name, c.launches, ms, flops / (ms * 1e-3) / 1e12, cycles / (ms * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
