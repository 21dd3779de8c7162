// Adam arithmetic + the device-side step state shared by the stand-alone optimiser launch (adam.hip) and the end-of-pass
// reduction launch that applies the update to the gradients it has just finished (dense_gemm.hip): ONE definition, so the
// two paths produce the same bits.  Update rule = torch.optim.Adam (no weight decay, no amsgrad):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;  p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// state: [0] t (float), [1] b1^t, [2] b2^t, [3] root arrivals, [32 * (1 + l)] arrivals of leaf l (uint words).
#pragma once
#include "geom_common.h"

namespace geom {

constexpr int ADAM_LEAVES = 64;
constexpr int ADAM_LEAF_STRIDE = 32; // words: every leaf counter on a cache line of its own (the root shares line 0 with the
                                     // state).  With the leaves packed into two lines, the 2 000 arrivals of the reduction
                                     // launch all went through one L2 channel one after the other
static_assert(GEOM_ADAM_STATE_WORDS >= ADAM_LEAF_STRIDE * (ADAM_LEAVES + 1), "state layout: {t, b1^t, b2^t, root}, leaf[64] x 32 words");

__device__ __forceinline__ void adam_update(float &p, float g, float &m, float &v, float b1, float b2, float eps,
                                            float grad_scale, float step_size, float bc2_sqrt)
{
    const float gi = g * grad_scale;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi;
    v = vi;
    p -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}

struct AdamStep {
    float t_old, b1t, b2t, step_size, bc2_sqrt;
};

// the step quantities from the published state st = {t, b1^t, b2^t} (LDS, after the barrier that published it)
__device__ __forceinline__ AdamStep adam_step_from(const float *st, float lr, float b1, float b2)
{
    AdamStep a;
    a.t_old = st[0];
    a.b1t = a.t_old == 0.f ? b1 : st[1] * b1;
    a.b2t = a.t_old == 0.f ? b2 : st[2] * b2;
    a.step_size = lr / (1.f - a.b1t);
    a.bc2_sqrt = sqrtf(1.f - a.b2t);
    return a;
}

// The step state is read ONCE per workgroup, by the thread that later signs the workgroup's arrival, with agent-scope
// atomic loads, and handed to the other threads through LDS (`st`, 3 floats): that thread's loads have RETURNED (their
// values were stored to LDS in front of the barrier) before it can reach its arrival atomic -- the order "read the state,
// then arrive" is a data dependency, not an assumption about issue order.  Contains a __syncthreads().
__device__ __forceinline__ AdamStep adam_read_state(const float *state, float *st, float lr, float b1, float b2)
{
    if (threadIdx.x == 0) {
        st[0] = __hip_atomic_load(state + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st[1] = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st[2] = __hip_atomic_load(state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return adam_step_from(st, lr, b1, b2);
}

// Arrival tree (call with every thread of every workgroup of the launch, after the workgroup's last use of the state;
// contains a __syncthreads()): workgroup w arrives at leaf counter w % 64, the last arriver of a leaf arrives at the root,
// the last arriver of the root writes the new state and re-arms the counters.  Relaxed atomics on purpose: a release here
// would write back the L2 lines this workgroup just dirtied with p / m / v (measured: 13.9 us for the launch with acq_rel
// arrivals against ~4 us), and nothing reads those before the kernel boundary anyway.
__device__ __forceinline__ void adam_arrive(float *state, const AdamStep &a, unsigned block, unsigned nblk)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("" ::: "memory");
        unsigned *cnt = reinterpret_cast<unsigned *>(state) + 3;
        unsigned *leaves_base = reinterpret_cast<unsigned *>(state) + ADAM_LEAF_STRIDE;
        const unsigned leaf = block % ADAM_LEAVES;
        const unsigned leaf_total = (nblk - leaf + ADAM_LEAVES - 1) / ADAM_LEAVES; // workgroups mapped to this leaf
        const unsigned leaves = nblk < ADAM_LEAVES ? nblk : ADAM_LEAVES;
        unsigned *mine = leaves_base + leaf * ADAM_LEAF_STRIDE;
        if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == leaf_total - 1) {
            __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == leaves - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(state + 0, a.t_old + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(state + 1, a.b1t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(state + 2, a.b2t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

} // namespace geom
