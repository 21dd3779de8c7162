// Random face draws of batch_sample (reference utils.py:604-612, 627-628): the workgroup body shared by the stand-alone
// launch (sample_loss.hip) and the fused surface-prepare launch (tri_distance.hip: draws + triangle-record prep in ONE
// launch -- both depend only on the vertex positions).
#pragma once
#include "geom_common.h"
#include "tri_math.h"

namespace {

__device__ __forceinline__ geom::V3 draw_ld3(const float *p) { return geom::mk(p[0], p[1], p[2]); }

constexpr int DRAW_THREADS = 1024;
constexpr int DRAW_MAX_FACES = 16384; // 64 KiB of LDS

// Philox4x32-10 (Salmon et al. 2011): counter-based, so a sample's three uniforms depend only on
// (seed, stream position, mesh, sample index) -- no generator state to carry, and a captured HIP graph
// draws fresh numbers on every replay because the stream position lives in device memory.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * ctr.z;
        ctr = make_uint4((unsigned)(p1 >> 32) ^ ctr.y ^ key.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ ctr.w ^ key.y, (unsigned)p0);
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return ctr;
}
__device__ __forceinline__ float u01(unsigned x) { return (x >> 8) * 0x1p-24f; } // [0,1), 24 random bits

// rng_state = {seed, stream position, arrival counter, first GLOBAL mesh index of this shard}: the LAST workgroup of
// a call to finish advances the position (every workgroup has read it by then) and re-arms the counter -- no separate
// tick launch.  The counter is keyed on the global mesh index, so data-parallel ranks that share a seed draw exactly
// what one process holding the whole batch would draw (and never the same samples for different meshes).

// chunk / mesh: which 1024 samples of which mesh; groups: workgroups of this launch that run this body (the arrival count)
__device__ __forceinline__ void draw_samples_body(int chunk, int mesh, unsigned long long groups, int nv, const float *verts,
                                                  int nf, const int64_t *faces, int num, const float *uniforms, int64_t plane,
                                                  unsigned long long *rng_state, int64_t *choices, float *u, float *v,
                                                  float *points)
{
    __shared__ float cdf[DRAW_MAX_FACES];
    __shared__ float wave_total[DRAW_THREADS / GEOM_WAVE];
    const float *V = verts + (size_t)mesh * nv * 3;
    // the random stream's state is requested first: read after the CDF it would add a dependent round trip to a launch
    // that is a chain of them (face ids -> corners -> scan -> search -> face ids -> corners -> store)
    unsigned long long seed = 0ull, pos = 0ull, mesh0 = 0ull;
    if (rng_state) {
        seed = rng_state[0];
        pos = rng_state[1];
        mesh0 = rng_state[3];
    }
    const int per = (nf + DRAW_THREADS - 1) / DRAW_THREADS; // consecutive faces per thread
    const int f0 = threadIdx.x * per;
    float run = 0.f;
    const int f1 = min(f0 + per, nf);
#pragma unroll 4
    for (int f = f0; f < f1; ++f) { // local inclusive sums
        const geom::V3 v0 = draw_ld3(V + 3 * faces[3 * (size_t)f + 0]);
        const geom::V3 v1 = draw_ld3(V + 3 * faces[3 * (size_t)f + 1]);
        const geom::V3 v2 = draw_ld3(V + 3 * faces[3 * (size_t)f + 2]);
        const geom::V3 x = v0 - v1, y = v1 - v2;
        const float ca = x.y * y.z - x.z * y.y, cb = x.z * y.x - x.x * y.z, cc = x.x * y.y - x.y * y.x;
        run += sqrtf((ca * ca + cb * cb) + cc * cc) / 2.f;
        cdf[f] = run;
    }
    // exclusive offset of this thread: wave scan of the thread totals, then scan of the 16 wave totals
    const int lane = threadIdx.x & (GEOM_WAVE - 1), wave = threadIdx.x >> 6;
    float incl = run;
    for (int off = 1; off < GEOM_WAVE; off <<= 1) {
        const float t = __shfl_up(incl, off, GEOM_WAVE);
        if (lane >= off) incl += t;
    }
    if (lane == GEOM_WAVE - 1) wave_total[wave] = incl;
    __syncthreads();
    float base = 0.f;
    for (int w = 0; w < wave; ++w) base += wave_total[w];
    const float offset = base + (incl - run);
    for (int f = f0; f < min(f0 + per, nf); ++f) cdf[f] += offset;
    __syncthreads();
    const float total = cdf[nf - 1];

    const int i = chunk * DRAW_THREADS + threadIdx.x;
    if (rng_state) {
        __syncthreads(); // every thread of this workgroup holds the position before the workgroup reports in
        if (threadIdx.x == 0) {
            if (atomicAdd(&rng_state[2], 1ull) == groups - 1ull) { // last one in: nobody reads the old position any more
                rng_state[2] = 0ull;
                rng_state[1] = pos + 1ull;
            }
        }
    }
    if (i >= num) return;
    const int64_t o = (int64_t)mesh * num + i;
    float r0, r1, r2;
    if (rng_state) { // in-kernel Philox: counter = (sample, mesh, stream position), key = seed
        const uint4 r = philox4x32_10(make_uint4((unsigned)i, (unsigned)(mesh0 + mesh), (unsigned)pos, (unsigned)(pos >> 32)),
                                      make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
        r0 = u01(r.x), r1 = u01(r.y), r2 = u01(r.z);
    } else {
        r0 = uniforms[o], r1 = uniforms[plane + o], r2 = uniforms[2 * plane + o];
    }
    const float target = r0 * total;
    int lo = 0, hi = nf - 1; // first f with cdf[f] > target, clamped to the last face
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > target) hi = mid;
        else lo = mid + 1;
    }
    choices[o] = lo;
    const float su = sqrtf(r1);
    u[o] = su;
    v[o] = r2;
    if (points) { // the sampled point itself, same expression as sample_fwd_kernel (utils.py:630)
        const geom::V3 x = draw_ld3(V + 3 * faces[3 * (size_t)lo + 0]);
        const geom::V3 y = draw_ld3(V + 3 * faces[3 * (size_t)lo + 1]);
        const geom::V3 z = draw_ld3(V + 3 * faces[3 * (size_t)lo + 2]);
        const float w0 = 1.f - su;
        const float w1 = su * (1.f - r2);
        const float w2 = su * r2;
        const geom::V3 pt = (x * w0 + y * w1) + z * w2;
        points[3 * o + 0] = pt.x;
        points[3 * o + 1] = pt.y;
        points[3 * o + 2] = pt.z;
    }
}


} // namespace
