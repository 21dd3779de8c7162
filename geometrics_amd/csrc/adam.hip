// Multi-tensor Adam for the replicated 0N-GCN parameters (the optimiser the reference drivers use:
// GEOMetrics.py:73, optim.Adam(lr=1e-4)).  ONE launch for up to GEOM_ADAM_MAX_TENSORS parameter tensors; the step
// counter and the running beta powers live in device memory, so the update is HIP-graph replayable with no host
// scalars baked in.
//
// Who advances the step state?  Every workgroup reads {t, b1^t, b2^t} at its start and derives the bias corrections
// of step t+1 from it; the state may only change after the LAST workgroup has read it.  Round 1 used a 1-thread tick
// kernel in front (a 4.6 us launch floor per step); a single last-arriver counter was measured and rejected (6144
// same-address atomics serialise to ~50 us).  Here arrivals go through a two-level tree: workgroup w arrives at
// leaf counter w % 64, the last arriver of a leaf arrives at the root, the last arriver of the root writes the new
// state and re-arms the counters -- at most ceil(N/64) + 64 same-address atomics on any word (~1 us for the bench's
// 254 workgroups), all off the critical path except the final hop.  `advance` = 0 skips the protocol: that is how an
// optimiser with more than 64 tensors issues several launches that all use the bias corrections of ONE step (only
// the last launch advances).
// Update rule = torch.optim.Adam (no weight decay, no amsgrad):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
#include "geom_common.h"
#include "adam_math.h"

namespace {

using geom::adam_update;

struct AdamTensors {
    float *p[GEOM_ADAM_MAX_TENSORS];
    const float *g[GEOM_ADAM_MAX_TENSORS];
    float *m[GEOM_ADAM_MAX_TENSORS];
    float *v[GEOM_ADAM_MAX_TENSORS];
    int64_t n[GEOM_ADAM_MAX_TENSORS];
    int first_block[GEOM_ADAM_MAX_TENSORS + 1]; // workgroups [first_block[i], first_block[i+1]) own tensor i
    int count;
};

// state: [0] t (float), [1] b1^t, [2] b2^t, [3] root arrivals, [4..67] leaf arrivals (uint words)
__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, float lr, float b1, float b2, float eps,
                                                   float grad_scale, float *state, int advance)
{
    // the tensor this workgroup works on: the last i with first_block[i] <= blockIdx.x (uniform binary search, <= 6 steps;
    // empty tensors own no workgroup and are skipped by it)
    int lo = 0, hi = t.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= t.first_block[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int which = lo;
    const int local = blockIdx.x - t.first_block[which];

    __shared__ float st[3];
    const geom::AdamStep as = geom::adam_read_state(state, st, lr, b1, b2);   // see adam_math.h for the ordering argument
    const float step_size = as.step_size, bc2_sqrt = as.bc2_sqrt;

    float *p = t.p[which], *m = t.m[which], *v = t.v[which];
    const float *g = t.g[which];
    const int64_t n = t.n[which];
    const int64_t base = ((int64_t)local * 256 + threadIdx.x) * 4;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (base + 4 <= n && vec) {
        float4 pp = *reinterpret_cast<float4 *>(p + base);
        const float4 gg = *reinterpret_cast<const float4 *>(g + base);
        float4 mm = *reinterpret_cast<float4 *>(m + base);
        float4 vv = *reinterpret_cast<float4 *>(v + base);
        adam_update(pp.x, gg.x, mm.x, vv.x, b1, b2, eps, grad_scale, step_size, bc2_sqrt);
        adam_update(pp.y, gg.y, mm.y, vv.y, b1, b2, eps, grad_scale, step_size, bc2_sqrt);
        adam_update(pp.z, gg.z, mm.z, vv.z, b1, b2, eps, grad_scale, step_size, bc2_sqrt);
        adam_update(pp.w, gg.w, mm.w, vv.w, b1, b2, eps, grad_scale, step_size, bc2_sqrt);
        *reinterpret_cast<float4 *>(m + base) = mm;
        *reinterpret_cast<float4 *>(v + base) = vv;
        *reinterpret_cast<float4 *>(p + base) = pp;
    } else {
        for (int64_t i = base; i < n && i < base + 4; ++i) {
            float pi = p[i], mi = m[i], vi = v[i];
            adam_update(pi, g[i], mi, vi, b1, b2, eps, grad_scale, step_size, bc2_sqrt);
            m[i] = mi;
            v[i] = vi;
            p[i] = pi;
        }
    }

    if (!advance) return;
    geom::adam_arrive(state, as, blockIdx.x, gridDim.x);
}

} // namespace

extern "C" int geom_adam_step_f32(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                                  float *const *exp_avg_sq, const int64_t *sizes, float lr, float beta1, float beta2,
                                  float eps, float grad_scale, float *state, int advance, void *stream)
{
    if (count < 0 || count > GEOM_ADAM_MAX_TENSORS) return GEOM_ETOOBIG;
    if (count == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !state) return GEOM_EINVAL;
    AdamTensors t;
    int64_t blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || sizes[i] < 0) return GEOM_EINVAL;
        t.p[i] = params[i];
        t.g[i] = grads[i];
        t.m[i] = exp_avg[i];
        t.v[i] = exp_avg_sq[i];
        t.n[i] = sizes[i];
        t.first_block[i] = (int)blocks;
        blocks += (sizes[i] + 1023) / 1024; // 256 threads x 4 elements
        if (blocks > 0x3fffffff) return GEOM_ETOOBIG;
    }
    t.first_block[count] = (int)blocks;
    t.count = count;
    if (blocks == 0) blocks = 1; // only empty tensors: one workgroup still advances the state
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t, lr, beta1, beta2, eps, grad_scale, state,
                       advance);
    return geom::launch_status();
}
