// Timing probe for csrc/dense_gemm.hip: builds the kernels with -DDG_PROBE_* knobs (work removed piece by piece; results
// are then WRONG, only the time is read) and times the forward product at the BASELINE shapes with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I include -I geometrics_amd/csrc [-DDG_PROBE_x] tools/probe/dense_probe.cpp -o probe_x
#include "../../geometrics_amd/csrc/dense_gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv)
{
    const int rows = 20496, c = 192;
    std::vector<int> cins = {963, 192};
    if (argc > 2) { cins.clear(); for (int a = 2; a < argc; ++a) cins.push_back(atoi(argv[a])); }
    for (int cin : cins) {
        float *x[3], *w, *out, *g, *dx, *ws;
        for (int i = 0; i < 3; ++i) hipMalloc(&x[i], (size_t)rows * cin * 4);
        hipMalloc(&w, (size_t)cin * c * 4);
        hipMalloc(&out, (size_t)rows * c * 4);
        hipMalloc(&g, (size_t)rows * c * 4);
        hipMalloc(&dx, (size_t)rows * cin * 4);
        hipMalloc(&ws, (size_t)geom_dense_bwd_weight_workspace_floats(rows, cin, c) * 4);
        std::vector<float> h((size_t)rows * (cin > c ? cin : c));
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
        for (int i = 0; i < 3; ++i) hipMemcpy(x[i], h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, h.data(), (size_t)cin * c * 4, hipMemcpyHostToDevice);
        hipMemcpy(g, h.data(), (size_t)rows * c * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int iters = 30;
        hipStream_t sa, sb;
        hipStreamCreate(&sa);
        hipStreamCreate(&sb);
        for (int which = 0; which < 4; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                for (int i = 0; i < 3; ++i) {
                    if (which == 0) geom_dense_fwd_f32(rows, cin, c, x[i], w, 0, nullptr, out, nullptr, nullptr, nullptr);
                    if (which == 1) geom_dense_bwd_input_f32(rows, cin, c, g, w, dx, nullptr);
                    if (which == 2) geom_dense_bwd_weight_f32(rows, cin, c, x[i], g, ws, 0, nullptr);
                }
                hipDeviceSynchronize();
                hipEventRecord(e0, nullptr);
                for (int i = 0; i < iters; ++i) {
                    if (which == 3) { // dX and dW of one layer side by side on two streams
                        geom_dense_bwd_input_f32(rows, cin, c, g, w, dx, sa);
                        geom_dense_bwd_weight_f32(rows, cin, c, x[i % 3], g, ws, 0, sb);
                    }
                    if (which == 0) geom_dense_fwd_f32(rows, cin, c, x[i % 3], w, 0, nullptr, out, nullptr, nullptr, nullptr);
                    if (which == 1) geom_dense_bwd_input_f32(rows, cin, c, g, w, dx, nullptr);
                    if (which == 2) geom_dense_bwd_weight_f32(rows, cin, c, x[i % 3], g, ws, 0, nullptr);
                }
                if (which == 3) { hipStreamSynchronize(sa); hipStreamSynchronize(sb); }
                hipEventRecord(e1, nullptr);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double us = best * 1e3 / iters, flop = 2.0 * rows * cin * c;
            printf("%s cin %4d %-4s %7.1f us  %6.1f TFLOP/s\n", argc > 1 ? argv[1] : "", cin, which == 0 ? "fwd" : which == 1 ? "dX" : which == 2 ? "dW" : "dX||dW", us,
                   flop / us / 1e6);
        }
        for (int i = 0; i < 3; ++i) hipFree(x[i]);
        hipFree(w), hipFree(out), hipFree(g), hipFree(dx), hipFree(ws);
    }
    return 0;
}
