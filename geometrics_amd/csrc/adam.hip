// Multi-tensor Adam for the replicated 0N-GCN parameters (the optimiser the reference drivers use:
// GEOMetrics.py:73, optim.Adam(lr=1e-4)).  One launch for every parameter tensor, the step
// counter and the running beta powers live in device memory (updated by a 1-thread tick kernel),
// so the whole update is HIP-graph replayable with no host scalars baked in.  (Folding the tick into the update
// kernel through a last-workgroup-arrives counter was measured twice: with the 6144-workgroup grid the same-address
// atomics serialise to 50 us; with a 66-workgroup grid the update itself slows to 8-9 us -- no better than 4.6 + 4.5.)
// Update rule = torch.optim.Adam (no weight decay, no amsgrad):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
#include "geom_common.h"

namespace {

struct AdamTensors {
    float *p[GEOM_ADAM_MAX_TENSORS];
    const float *g[GEOM_ADAM_MAX_TENSORS];
    float *m[GEOM_ADAM_MAX_TENSORS];
    float *v[GEOM_ADAM_MAX_TENSORS];
    int64_t n[GEOM_ADAM_MAX_TENSORS];
    int count;
};

// state[0] = t (as float), state[1] = b1^t, state[2] = b2^t
__global__ void adam_tick_kernel(float b1, float b2, float *state)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const bool first = state[0] == 0.f;
        state[0] += 1.f;
        state[1] = first ? b1 : state[1] * b1;
        state[2] = first ? b2 : state[2] * b2;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, float lr, float b1, float b2, float eps,
                                                   float grad_scale, const float *state)
{
    const int which = blockIdx.y;
    const float bc1 = 1.f - state[1];
    const float bc2_sqrt = sqrtf(1.f - state[2]);
    const float step_size = lr / bc1;
    float *p = t.p[which], *m = t.m[which], *v = t.v[which];
    const float *g = t.g[which];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.n[which]; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

} // namespace

extern "C" int geom_adam_step_f32(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                                  float *const *exp_avg_sq, const int64_t *sizes, float lr, float beta1, float beta2,
                                  float eps, float grad_scale, float *state, void *stream)
{
    if (count < 0 || count > GEOM_ADAM_MAX_TENSORS) return GEOM_ETOOBIG;
    if (count == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !state) return GEOM_EINVAL;
    AdamTensors t;
    int64_t longest = 0;
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || sizes[i] < 0) return GEOM_EINVAL;
        t.p[i] = params[i];
        t.g[i] = grads[i];
        t.m[i] = exp_avg[i];
        t.v[i] = exp_avg_sq[i];
        t.n[i] = sizes[i];
        if (sizes[i] > longest) longest = sizes[i];
    }
    t.count = count;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, s, beta1, beta2, state);
    int64_t blocks = (longest + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks, count), dim3(256), 0, s, t, lr, beta1, beta2, eps, grad_scale,
                       state);
    return geom::launch_status();
}
