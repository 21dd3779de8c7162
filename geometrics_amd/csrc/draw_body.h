// Random face draws of batch_sample (reference utils.py:604-612, 627-628): the workgroup body shared by the stand-alone
// launch (sample_loss.hip) and the fused surface-prepare launch (tri_distance.hip: draws + triangle-record prep in ONE
// launch -- both depend only on the vertex positions).
#pragma once
#include "geom_common.h"
#include "nn_scan.h"
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

// SORTED (the culled Chamfer scan of the same step, nn_scan.h) draws the samples of a mesh ALREADY IN A VISITING ORDER:
// ascending position of their face in the triangle order, so that neighbours in the list are neighbours in space as far as
// that order is coherent -- without sorting anything.  The face of sample i is found with the i-th SMALLEST of `num`
// uniforms, and order statistics of uniforms can be generated directly: with E_1..E_{num+1} independent Exp(1) variates,
// U_(i) = (E_1 + .. + E_i) / (E_1 + .. + E_{num+1}) has exactly the joint law of the sorted uniforms.  One exponential
// per sample (-log of its Philox uniform), a fixed-order block scan, one division.  The multiset of samples is distributed
// as with independent draws (the reference's torch.multinomial with replacement); only their ORDER is no longer random,
// which is why this mode exists inside the fused surface loss only (its samples never leave it) and not in batch_sample.
// Round 3 sorted independent draws by face rank instead (histogram, scan, scatter, re-rank in ONE workgroup per mesh):
// 28 us against 13 for the launch, more than the culled scan saved.
// The CDF is built over the faces in visiting order (position j = face face_order[j]); the launch also writes the
// visiting-order copy of the sampled points and its run spheres (the "index" nn_culled_body reads; the sample order is the
// identity).
struct DrawSort {
    const int *face_order; // [nf] visiting position -> face (null: the faces' own order)
    const int *pfaces;     // [nf,3] optional: the corners of face face_order[j] at position j (static per face list and order:
                           // one dependent load less in each of the launch's two gather chains)
    float *xs;             // [b][nn_cull_stride(num)] out
    float4 *sph;           // [b][num / 16] out
};
constexpr int DRAW_SORT_CHUNKS = 4; // samples per mesh on the sorted route: at most 4096

// chunk / mesh: which 1024 samples of which mesh; groups: workgroups of this launch that run this body (the arrival count)
template <bool SORTED>
__device__ __forceinline__ void draw_samples_body(int chunk, int mesh, unsigned long long groups, int nv, const float *verts,
                                                  int nf, const int64_t *faces, int num, const float *uniforms, int64_t plane,
                                                  unsigned long long *rng_state, int64_t *choices, float *u, float *v,
                                                  float *points, const DrawSort &srt)
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
    const int lane = threadIdx.x & (GEOM_WAVE - 1), wave = threadIdx.x >> 6;
    __shared__ float chunk_part[SORTED ? DRAW_SORT_CHUNKS : 1][DRAW_THREADS / GEOM_WAVE]; // SORTED: spacing sums per chunk and wave
    __shared__ float scan_part[DRAW_THREADS / GEOM_WAVE];
    // SORTED: the i-th order statistic of `num` uniforms from exponential spacings (see the top of the file).  Thread t
    // holds the exponentials of samples t, t + 1024, ... (one per chunk); chunk totals by a fixed tree, then the inclusive
    // scan inside this workgroup's own chunk.
    float sorted_u = 0.f, incl_e = 0.f;
    if constexpr (SORTED) {
        const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
        auto expo = [&](int j) { // Exp(1) from sample j's first Philox word; j == num: the extra spacing behind the last sample
            const uint4 r = philox4x32_10(make_uint4((unsigned)j, (unsigned)(mesh0 + mesh), (unsigned)pos, (unsigned)(pos >> 32)), key);
            return -__logf(1.f - u01(r.x)); // 1 - u in (0, 1]: finite, >= 0
        };
        float mine = 0.f; // this thread's exponential in this workgroup's chunk
#pragma unroll
        for (int c = 0; c < DRAW_SORT_CHUNKS; ++c) {
            const int j = c * DRAW_THREADS + threadIdx.x;
            float e = j < num ? expo(j) : 0.f;
            if (j == num) e = expo(num); // the (num + 1)-th spacing lives in the slot behind the last sample (num < capacity)
            if (c == chunk) mine = j < num ? e : 0.f;
            float t = e;
#pragma unroll
            for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if (lane == 0) chunk_part[c][wave] = t;
        }
        incl_e = mine;
        for (int off = 1; off < GEOM_WAVE; off <<= 1) {
            const float t = __shfl_up(incl_e, off, GEOM_WAVE);
            if (lane >= off) incl_e += t;
        }
        if (lane == GEOM_WAVE - 1) scan_part[wave] = incl_e;
    }
    const int per = (nf + DRAW_THREADS - 1) / DRAW_THREADS; // consecutive faces per thread
    const int f0 = threadIdx.x * per;
    float run = 0.f;
    const int f1 = min(f0 + per, nf);
#pragma unroll 4
    for (int f = f0; f < f1; ++f) { // local inclusive sums
        int ff = f;
        bool real = true;
        int64_t c0, c1, c2;
        if (SORTED && srt.face_order && srt.pfaces) { // position f of the visiting order, corners listed in that order
            c0 = srt.pfaces[3 * f + 0], c1 = srt.pfaces[3 * f + 1], c2 = srt.pfaces[3 * f + 2];
        } else {
            if (SORTED && srt.face_order) { // through the order; an entry that is no face weighs nothing
                ff = srt.face_order[f];
                real = ff >= 0 && ff < nf;
                ff = real ? ff : 0;
            }
            c0 = faces[3 * (size_t)ff + 0], c1 = faces[3 * (size_t)ff + 1], c2 = faces[3 * (size_t)ff + 2];
        }
        const geom::V3 v0 = draw_ld3(V + 3 * c0);
        const geom::V3 v1 = draw_ld3(V + 3 * c1);
        const geom::V3 v2 = draw_ld3(V + 3 * c2);
        const geom::V3 x = v0 - v1, y = v1 - v2;
        const float ca = x.y * y.z - x.z * y.y, cb = x.z * y.x - x.x * y.z, cc = x.x * y.y - x.y * y.x;
        run += real ? sqrtf((ca * ca + cb * cb) + cc * cc) / 2.f : 0.f;
        cdf[f] = run;
    }
    // exclusive offset of this thread: wave scan of the thread totals, then scan of the 16 wave totals
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
    if constexpr (SORTED) { // (chunk_part / scan_part were written in front of the CDF scan's barrier)
        float before = 0.f, all = 0.f; // spacings of the chunks in front of this one / of all samples + the extra one
#pragma unroll
        for (int c = 0; c < DRAW_SORT_CHUNKS; ++c) {
            float t = 0.f;
            for (int w = 0; w < DRAW_THREADS / GEOM_WAVE; ++w) t += chunk_part[c][w];
            if (c < chunk) before += t;
            all += t;
        }
        for (int w = 0; w < wave; ++w) before += scan_part[w];
        sorted_u = fminf((before + incl_e) / all, 0x1.fffffep-1f); // in (0, 1): ascending in the sample index
    }

    if (rng_state) {
        __syncthreads(); // every thread of this workgroup holds the position before the workgroup reports in
        if (threadIdx.x == 0) {
            if (atomicAdd(&rng_state[2], 1ull) == groups - 1ull) { // last one in: nobody reads the old position any more
                rng_state[2] = 0ull;
                rng_state[1] = pos + 1ull;
            }
        }
    }
    // sample i of the mesh: the draw and its outputs
    const int i = chunk * DRAW_THREADS + threadIdx.x;
    geom::V3 pt = geom::mk(0.f, 0.f, 0.f);
    if (i < num) {
        const int64_t o = (int64_t)mesh * num + i;
        float r0, r1, r2;
        if (rng_state) { // in-kernel Philox: counter = (sample, mesh, stream position), key = seed
            const uint4 r = philox4x32_10(make_uint4((unsigned)i, (unsigned)(mesh0 + mesh), (unsigned)pos, (unsigned)(pos >> 32)),
                                          make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
            r0 = u01(r.x), r1 = u01(r.y), r2 = u01(r.z);
        } else {
            r0 = uniforms[o], r1 = uniforms[plane + o], r2 = uniforms[2 * plane + o];
        }
        if (SORTED) r0 = sorted_u;
        const float target = r0 * total;
        int lo = 0, hi = nf - 1; // first f with cdf[f] > target, clamped to the last face
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > target) hi = mid;
            else lo = mid + 1;
        }
        const int slot = lo; // position in the visiting order (SORTED) / the face itself
        if (SORTED && srt.face_order) {
            const int ff = srt.face_order[lo];
            lo = ff >= 0 && ff < nf ? ff : 0;
        }
        choices[o] = lo;
        const float su = sqrtf(r1);
        u[o] = su;
        v[o] = r2;
        if (points || SORTED) { // the sampled point itself, same expression as sample_fwd_kernel (utils.py:630)
            int64_t c0, c1, c2;
            if (SORTED && srt.face_order && srt.pfaces) c0 = srt.pfaces[3 * slot + 0], c1 = srt.pfaces[3 * slot + 1], c2 = srt.pfaces[3 * slot + 2];
            else c0 = faces[3 * (size_t)lo + 0], c1 = faces[3 * (size_t)lo + 1], c2 = faces[3 * (size_t)lo + 2];
            const geom::V3 x = draw_ld3(V + 3 * c0);
            const geom::V3 y = draw_ld3(V + 3 * c1);
            const geom::V3 z = draw_ld3(V + 3 * c2);
            const float w0 = 1.f - su;
            const float w1 = su * (1.f - r2);
            const float w2 = su * r2;
            pt = (x * w0 + y * w1) + z * w2;
            if (points) {
                points[3 * o + 0] = pt.x;
                points[3 * o + 1] = pt.y;
                points[3 * o + 2] = pt.z;
            }
        }
    }
    if constexpr (SORTED) { // whole waves: the run spheres are 16-lane reductions (threads past `num` take part with zeros)
        if (chunk * DRAW_THREADS < num)
            nn_cull_emit(pt.x, pt.y, pt.z, num, i, srt.xs + (size_t)mesh * nn_cull_stride(num), srt.sph + (size_t)mesh * (num / NNS_GROUP));
    }
}

} // namespace
