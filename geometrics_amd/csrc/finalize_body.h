// The forward-side FINALIZE body of the surface loss (see surface_gather.hip for what it does and why): shared by the
// stand-alone launch (surface_finalize_kernel, 1024 threads) and by the role workgroups of the fused surface scan
// (tri_distance.hip: surface_scan_kernel, 512 threads).  THREADS is the workgroup size; the RESULTS do not depend on it:
//   * the loss is summed by 1024 VIRTUAL threads (a physical thread runs 1024 / THREADS of them) and folded by the same
//     tree -- same bits from either workgroup shape;
//   * the ordering is a function of the data only (ascending point id inside every face).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "geom_common.h"
#include "tri_math.h"

#ifdef SCAN_TILE_STAMPS // tools/probe only: phase boundaries of the finalize body inside the fused scan launch (thread 0's clock)
__device__ long long scan_role_phases[16 * 16];
#define FIN_PHASE(k) do { if (threadIdx.x == 0 && blockIdx.x < 16) scan_role_phases[16 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define FIN_PHASE(k) do { } while (0)
#endif

namespace geom_finalize {

using geom::V3;
enum { OTHER_NONE = 0, OTHER_NN = 1, OTHER_TRI = 2 };
constexpr int FIN_VIRTUAL = 1024;               // virtual threads of the loss sum
constexpr int FIN_VWAVES = FIN_VIRTUAL / GEOM_WAVE;
constexpr int FIN_REG_POINTS = 8192;            // points per mesh the REGS variant keeps in registers
constexpr int SCRATCH_CH = 6;                   // the scratch variant: ids per thread whose loads are in flight together (binning)
constexpr int ORD_CH = 12;                      // ... in the two ordering passes
constexpr int GROUP_LOOKBACK = 64;              // longest run of one face's samples a role looks back over (beyond: ranking pass)

__device__ __forceinline__ V3 ld3(const float *p) { return geom::mk(p[0], p[1], p[2]); }

struct FinalizeArgs {
    const int64_t *choices;
    const float *u, *v, *points, *gt;
    const int *idx_g, *idx_p, *index;
    const float *closest, *weights;
    const float *sq_sample, *sq_other; // [b,num], [b,n_gt]: the squared distances the loss sums
    float scale_sample, scale_other;   // loss = scale_sample * sum(sq_sample) + scale_other * sum(sq_other)
    float coef_sample, coef_other;     // gradient coefficients of the two kinds of points (without 2 * upstream grad)
    int b, nf, num, n_gt, other, per, want_order, records_ready;
    int *off, *seg, *pface, *slot;
    float4 *rec;
    float *loss;
    int *status; // [b + 1] words of the scratch (surface_layout.h): 0 = this role's result is complete; null: no scratch
    const float *mesh_weight; // [b] or null: loss = sum_m w[m] * (scale_sample * sum(sq_sample[m]) + scale_other * sum(sq_other[m]))
};


// sum over the 1024 virtual threads: v[j] = value of virtual thread tid + j * THREADS; lds16 = FIN_VWAVES floats
template <int THREADS>
__device__ __forceinline__ float block_sum_virtual(float (&v)[FIN_VIRTUAL / THREADS], float *lds16, int tid)
{
    constexpr int V = FIN_VIRTUAL / THREADS;
#pragma unroll
    for (int j = 0; j < V; ++j)
        for (int d = GEOM_WAVE / 2; d > 0; d >>= 1) v[j] += __shfl_down(v[j], d, GEOM_WAVE);
    __syncthreads();
    if ((tid & (GEOM_WAVE - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) lds16[(tid >> 6) + j * (THREADS / GEOM_WAVE)] = v[j];
    }
    __syncthreads();
    float t = 0.f;
    if (tid < GEOM_WAVE) {
        t = tid < FIN_VWAVES ? lds16[tid] : 0.f;
        for (int d = GEOM_WAVE / 2; d > 0; d >>= 1) t += __shfl_down(t, d, GEOM_WAVE);
    }
    return t; // valid in thread 0
}

// block = the workgroup's role: [0, b) orders mesh `block`, b reduces the loss (want_order = 0: the loss role only).
// ord_lds: (nf + 1 + per + THREADS / 64 + 4) ints + 2 * FIN_VWAVES floats (finalize_lds_ints()).
// wait(): called by every thread of the workgroup (uniformly) in front of the first read of anything the scans of the
// SAME launch produce -- the stand-alone launch passes a no-op, the fused scan's role workgroups their counter wait.
// With ready records the sampled points (their faces are the draws: known before the scans) are binned in front of it.
// Returns (to every thread alike) whether what it waited for is there; false: the role gives up -- it marks its status
// word, the loss role also writes NaN, and NOTHING of the incomplete results is read.
struct FinalizeNoWait {
    __device__ __forceinline__ bool operator()() const { return true; }
};
// ITEMS: points per thread the REGS variant keeps in registers (per <= ITEMS * THREADS; beyond: the scratch variant).
// READY: the caller guarantees a.records_ready (the code that forms records is left out: it is what made the register
// variant spill inside the fused scan launch).
template <bool REGS, int THREADS, typename Wait = FinalizeNoWait, int ITEMS = FIN_REG_POINTS / THREADS, bool READY = false>
__device__ __forceinline__ void surface_finalize_body(const FinalizeArgs &a, int *ord_lds, const int block, Wait wait = Wait())
{
    constexpr int WAVES = THREADS / GEOM_WAVE;
    constexpr int V = FIN_VIRTUAL / THREADS;
    int *off = ord_lds;                       // [nf+1]: counts, then offsets (ordering only)
    int *seg = off + (a.want_order ? a.nf + 1 : 0); // [per]
    int *wave_total = seg + (a.want_order ? a.per : 0);
    float *fsum = reinterpret_cast<float *>(wave_total + WAVES + 4);
    const int mesh = a.want_order ? block : a.b, tid = threadIdx.x; // without ordering the grid is the loss workgroup alone
    const int lane = tid & (GEOM_WAVE - 1), wave = tid >> 6;

    // ---- the extra workgroup (block == b) reduces the loss while the others order their meshes: float4 loads, all
    //      of a thread's loads in flight together, then a fixed tree -- no cross-workgroup hand-off at all ----
    if (mesh == a.b) {
        if (!wait()) {
            if (tid == 0) {
                a.loss[0] = __builtin_nanf("");
                if (a.status) a.status[a.b] = 1;
            }
            return;
        }
        auto virtual_sum = [&](const float *x, int64_t n, int vt) {
            float acc = 0.f;
            const bool vec = (((uintptr_t)x) & 15) == 0;
            const int64_t n4 = vec ? n / 4 : 0;
            const float4 *x4 = reinterpret_cast<const float4 *>(x);
            for (int64_t base = 0; base < n4; base += (int64_t)8 * FIN_VIRTUAL) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int64_t i = base + vt + (int64_t)k * FIN_VIRTUAL;
                    v[k] = i < n4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            }
            for (int64_t i = 4 * n4 + vt; i < n; i += FIN_VIRTUAL) acc += x[i];
            return acc;
        };
        // per-mesh weights (several equal-size batches stacked into one call, each with its own factor): the same walk over
        // the flat array, every element times its mesh's weight -- the mesh of a thread's next element follows from the
        // previous one (stride FIN_VIRTUAL elements), no division per element
        auto weighted_sum = [&](const float *x, int per_mesh, int vt) {
            float acc = 0.f;
            if (per_mesh <= 0) return acc;
            const int64_t n = (int64_t)a.b * per_mesh;
            const bool vec = (((uintptr_t)x) & 15) == 0 && per_mesh % 4 == 0; // (a float4 then never straddles two meshes)
            const int width = vec ? 4 : 1;
            const int64_t units = n / width;
            const float4 *x4 = reinterpret_cast<const float4 *>(x);
            int mesh = (int)(((int64_t)vt * width) / per_mesh), rem = (int)(((int64_t)vt * width) % per_mesh);
            for (int64_t base = 0; base < units; base += (int64_t)8 * FIN_VIRTUAL) {
                float4 v[8];
                float w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int64_t i = base + vt + (int64_t)k * FIN_VIRTUAL;
                    const bool in = i < units;
                    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (in && vec) v[k] = x4[i];
                    if (in && !vec) v[k].x = x[i];
                    w[k] = in ? a.mesh_weight[mesh] : 0.f;
                    rem += FIN_VIRTUAL * width;
                    while (rem >= per_mesh) rem -= per_mesh, ++mesh;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += w[k] * ((v[k].x + v[k].y) + (v[k].z + v[k].w));
            }
            return acc;
        };
        float s1[V], s2[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (a.mesh_weight) {
                s1[j] = weighted_sum(a.sq_sample, a.num, tid + j * THREADS);
                s2[j] = weighted_sum(a.sq_other, a.n_gt, tid + j * THREADS);
            } else {
                s1[j] = virtual_sum(a.sq_sample, (int64_t)a.b * a.num, tid + j * THREADS);
                s2[j] = virtual_sum(a.sq_other, (int64_t)a.b * a.n_gt, tid + j * THREADS);
            }
        }
        const float t1 = block_sum_virtual<THREADS>(s1, fsum, tid);
        const float t2 = block_sum_virtual<THREADS>(s2, fsum + FIN_VWAVES, tid);
        if (tid == 0) {
            a.loss[0] = t1 * a.scale_sample + t2 * a.scale_other;
            if (a.status) a.status[a.b] = 0;
        }
        return;
    }

    if (a.want_order) {
        FIN_PHASE(0);
        for (int f = tid; f <= a.nf; f += THREADS) off[f] = 0;
        __syncthreads();
        FIN_PHASE(1);
        const int64_t p0 = (int64_t)mesh * a.per;
        // record of point `id` -> global, its face counted in LDS; returns the face (-1: none) and the arrival slot
        auto bin_point = [&](int id, int &fi, int &sl) {
            int64_t f, sp = -1;
            if (READY || a.records_ready) { // the fused scan already wrote the record: only the face is needed here
                if (id < a.num) f = a.choices[(int64_t)mesh * a.num + id];
                else {
                    const int64_t o = (int64_t)mesh * a.n_gt + (id - a.num);
                    f = a.other == OTHER_TRI ? (int64_t)a.index[o] : a.choices[(int64_t)mesh * a.num + a.idx_p[o]];
                }
            } else {
                V3 g;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                float skip_zero = 0.f;
                if (id < a.num) {
                    sp = (int64_t)mesh * a.num + id;
                    f = a.choices[sp];
                    g = (ld3(a.points + 3 * sp) - ld3(a.gt + 3 * ((int64_t)mesh * a.n_gt + a.idx_g[sp]))) * a.coef_sample;
                } else {
                    const int64_t o = (int64_t)mesh * a.n_gt + (id - a.num);
                    if (a.other == OTHER_TRI) {
                        f = a.index[o];
                        g = (ld3(a.closest + 3 * o) - ld3(a.gt + 3 * o)) * a.coef_other;
                        w = make_float4(a.weights[3 * o], a.weights[3 * o + 1], a.weights[3 * o + 2], 0.f);
                        skip_zero = 1.f; // the scatter does not touch a corner whose weight is exactly zero
                    } else {
                        sp = (int64_t)mesh * a.num + a.idx_p[o];
                        f = a.choices[sp];
                        g = (ld3(a.points + 3 * sp) - ld3(a.gt + 3 * o)) * a.coef_other;
                    }
                }
                if (sp >= 0) {
                    const float u = a.u[sp], v = a.v[sp];
                    w = make_float4(1.f - u, u * (1.f - v), u * v, 0.f);
                }
                a.rec[2 * (p0 + id) + 0] = make_float4(g.x, g.y, g.z, skip_zero);
                a.rec[2 * (p0 + id) + 1] = w;
            }
            const bool on_mesh = f >= 0 && f < a.nf; // else: contributes nowhere
            fi = on_mesh ? (int)f : -1;
            sl = on_mesh ? atomicAdd(&off[fi], 1) : 0; // LDS atomic: arrival slot inside the face
        };
        int my_f[ITEMS > ORD_CH ? ITEMS : ORD_CH], my_slot[ITEMS];
        constexpr bool NO_WAIT = std::is_same<Wait, FinalizeNoWait>::value;
        int *ungrouped = wave_total + WAVES + 1; // one of the 4 spare ints behind the wave totals (scratch variant of a role)
        if (REGS && NO_WAIT) {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int id = tid + it * THREADS;
                my_f[it] = -1, my_slot[it] = 0;
                if (id < a.per) bin_point(id, my_f[it], my_slot[it]);
            }
        } else if (REGS) {
            const int early = READY || a.records_ready ? a.num : 0; // ids below: nothing of the scans is read for them
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int id = tid + it * THREADS;
                my_f[it] = -1, my_slot[it] = 0;
                if (id < early) bin_point(id, my_f[it], my_slot[it]);
            }
            FIN_PHASE(2);
            if (!wait()) {
                if (tid == 0 && a.status) a.status[mesh] = 1;
                return;
            }
            FIN_PHASE(3);
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int id = tid + it * THREADS;
                if (id >= early && id < a.per) bin_point(id, my_f[it], my_slot[it]);
            }
        } else {
            const int early = READY || a.records_ready ? a.num : 0;
            auto bin_range = [&](int lo, int hi) {
                for (int base = lo + tid; base < hi; base += SCRATCH_CH * THREADS) {
                    int fi[SCRATCH_CH], sl[SCRATCH_CH];
#pragma unroll
                    for (int k = 0; k < SCRATCH_CH; ++k) {
                        fi[k] = -1, sl[k] = 0;
                        if (base + k * THREADS < hi) bin_point(base + k * THREADS, fi[k], sl[k]);
                    }
#pragma unroll
                    for (int k = 0; k < SCRATCH_CH; ++k)
                        if (base + k * THREADS < hi) a.pface[p0 + base + k * THREADS] = fi[k], a.slot[p0 + base + k * THREADS] = sl[k];
                }
            };
            bin_range(0, early);
            // A role workgroup has time in front of its wait.  When a face's sampled points have CONSECUTIVE ids (the
            // culled route generates them in face-visiting order), a sample's final place inside its face's segment is
            // its distance to the first id of its run: found here by looking back, packed into the upper half of its slot
            // word, so that behind the wait only the gt points have to be ranked (5.1 -> 2.x us).  Checked, not assumed:
            // the last id of every run must see run length == the face's count, else everything takes the ranking pass.
            if (!NO_WAIT && early > 0) {
                if (tid == 0) *ungrouped = 0;
                __syncthreads(); // every sample is counted
                const int64_t *face_of = a.choices + (int64_t)mesh * a.num;
                for (int id = tid; id < early; id += THREADS) {
                    const int64_t f = face_of[id];
                    if (f < 0 || f >= a.nf) continue;
                    int r = 0;
                    while (r < GROUP_LOOKBACK && id - r - 1 >= 0 && face_of[id - r - 1] == f) ++r;
                    const bool run_ends = id + 1 == early || face_of[id + 1] != f;
                    if (r == GROUP_LOOKBACK || (run_ends && r + 1 != off[(int)f])) *ungrouped = 1;
                    a.slot[p0 + id] |= r << 16; // arrival slot (< 65 536: per fits the launch's LDS) below, run position above
                }
            }
            FIN_PHASE(2);
            if (!wait()) {
                if (tid == 0 && a.status) a.status[mesh] = 1;
                return;
            }
            FIN_PHASE(3);
            bin_range(early, a.per);
        }
        __syncthreads();
        FIN_PHASE(4);
        // ---- exclusive scan of the counts (consecutive faces per thread) ----
        const int chunk = (a.nf + THREADS - 1) / THREADS;
        const int f0 = min(a.nf, tid * chunk), f1 = min(a.nf, f0 + chunk);
        int run = 0;
        for (int f = f0; f < f1; ++f) {
            const int c = off[f];
            off[f] = run;
            run += c;
        }
        int incl = run;
        for (int d = 1; d < GEOM_WAVE; d <<= 1) {
            const int t = __shfl_up(incl, d, GEOM_WAVE);
            if (lane >= d) incl += t;
        }
        if (lane == GEOM_WAVE - 1) wave_total[wave] = incl;
        __syncthreads();
        int base = incl - run;
        for (int w = 0; w < wave; ++w) base += wave_total[w];
        for (int f = f0; f < f1; ++f) off[f] += base;
        if (tid == THREADS - 1) off[a.nf] = base + run;
        __syncthreads();
        FIN_PHASE(5);
        int *g_off = a.off + (int64_t)mesh * (a.nf + 1);
        for (int f = tid; f <= a.nf; f += THREADS) g_off[f] = off[f];
        // ---- ids at offset + slot, then ranked into ascending order ----
        int placed_below = 0; // ids below: their final place was found in front of the wait (grouped samples of a role)
        if (REGS) {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it)
                if (my_f[it] >= 0) seg[off[my_f[it]] + my_slot[it]] = tid + it * THREADS;
        } else {
            // ORD_CH ids per thread and round: their faces / slots requested together (one L2 round trip per round); the
            // first round's faces stay in registers for the ranking pass below
            placed_below = !NO_WAIT && (READY || a.records_ready) && a.num > 0 && *ungrouped == 0 ? a.num : 0;
            int *g_place = a.seg + p0;
            for (int base = tid, round = 0; base < a.per; base += ORD_CH * THREADS, ++round) {
                int f[ORD_CH], sl[ORD_CH];
#pragma unroll
                for (int k = 0; k < ORD_CH; ++k) {
                    const int id = base + k * THREADS;
                    f[k] = id < a.per ? a.pface[p0 + id] : -1;
                    sl[k] = id < a.per ? a.slot[p0 + id] : 0;
                }
#pragma unroll
                for (int k = 0; k < ORD_CH; ++k) {
                    const int id = base + k * THREADS;
                    if (f[k] >= 0) {
                        const bool final_place = id < placed_below;
                        const int at = off[f[k]] + (final_place ? sl[k] >> 16 : sl[k] & 0xffff);
                        seg[at] = id; // samples too: the gt points of the face are ranked against the whole segment
                        if (final_place) g_place[at] = id, f[k] = -1;
                    }
                    if (round == 0) my_f[k] = f[k];
                }
            }
        }
        __syncthreads();
        FIN_PHASE(6);
        // ---- every id ranked inside its face's segment -> ascending order.  (Measured alternatives: a thread per FACE that
        //      orders its segment serially, 10 faces per thread: 18-24 us instead of 5 -- a crowded face is quadratic for one
        //      thread; the same with crowded faces handed to a wave each: 13.7 us.) ----
        int *g_seg = a.seg + p0;
        auto place = [&](int id, int f) {
            const int s0 = off[f], n = off[f + 1] - s0;
            int rank = 0;
            if (n > 1) // a face that holds one point (most do): nothing to rank
                for (int j = 0; j < n; ++j) rank += seg[s0 + j] < id ? 1 : 0;
            g_seg[s0 + rank] = id;
        };
        if (REGS) {
            // segment bounds of all of a thread's ids first (2 * ITEMS LDS reads in flight), then id by id.  (One entry of
            // EVERY id's segment per round, rounds = the longest segment: 7.5 us instead of 5 -- a crowded face makes all
            // of its thread's ids wait.)
            int s0[ITEMS], cnt[ITEMS];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int f = my_f[it] >= 0 ? my_f[it] : 0;
                s0[it] = off[f];
                cnt[it] = my_f[it] >= 0 ? off[f + 1] - s0[it] : 0;
            }
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                if (cnt[it] == 0) continue;
                const int id = tid + it * THREADS;
                int rank = 0;
                if (cnt[it] > 1) // a face that holds one point (most do): nothing to rank
                    for (int j = 0; j < cnt[it]; ++j) rank += seg[s0[it] + j] < id ? 1 : 0;
                g_seg[s0[it] + rank] = id;
            }
        } else {
#pragma unroll
            for (int k = 0; k < ORD_CH; ++k)
                if (tid + k * THREADS < a.per && my_f[k] >= 0) place(tid + k * THREADS, my_f[k]);
            for (int base = tid + ORD_CH * THREADS; base < a.per; base += ORD_CH * THREADS) {
                int f[ORD_CH];
#pragma unroll
                for (int k = 0; k < ORD_CH; ++k) f[k] = base + k * THREADS < a.per ? a.pface[p0 + base + k * THREADS] : -1;
#pragma unroll
                for (int k = 0; k < ORD_CH; ++k)
                    if (f[k] >= 0 && base + k * THREADS >= placed_below) place(base + k * THREADS, f[k]);
            }
        }
        FIN_PHASE(7);
        if (tid == 0 && a.status) a.status[mesh] = 0;
    }

}

// LDS ints the body needs (ordering: offsets + ids + per-wave totals; always: the two loss-sum scratch rows)
__host__ __device__ inline size_t finalize_lds_ints(int nf, int per, int threads, bool want_order)
{
    return (want_order ? (size_t)nf + 1 + per : 0) + threads / GEOM_WAVE + 4 + 2 * FIN_VWAVES;
}

} // namespace geom_finalize
