// issue rate of un-packed vs packed fp32 VALU adds/multiplies on gfx950 (wave-instructions per SIMD-cycle)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float a, float b, int iters)
{
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float2v v = {x[i], x[i + 1]};
                float2v aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(v) : "v"(aa), "v"(bb));
                x[i] = v.x, x[i + 1] = v.y;
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *o; hipMalloc(&o, 256 * 4 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wg = 1; wg <= 8; wg *= 2) {   // 256-thread WGs per CU: waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wg), dim3(256), 0, 0, o, 1.0001f, 0.5f, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256 * wg), dim3(256), 0, 0, o, 1.0001f, 0.5f, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double laneops = 256.0 * wg * 256 * iters * 32.0;  // mul+add on 16 values
            printf("%s  %d waves/SIMD: %.3f ms  %.1f T lane-op/s\n", mode ? "packed  " : "unpacked", wg, ms, laneops / ms / 1e9);
        }
    return 0;
}
