// Chamfer brute-force NN, scalar-broadcast scan: the workgroup body shared by the stand-alone launch
// (chamfer_nn.hip) and the fused surface scan (tri_distance.hip: NN jobs and point-to-triangle tiles in ONE launch).
#pragma once
#include "geom_common.h"

namespace {

constexpr int NN_QUERIES = GEOM_WAVE; // queries per workgroup

struct NNJob {
    const float *xyz1, *xyz2;
    float *dist1, *dist2;
    int *idx1, *idx2;
    int b, n, m;
};

// optional epilogue of the fused surface scan: the gradient record of every point the finalize / gather passes of the
// surface loss need (csrc/surface_gather.hip): {(point - partner) * coef, flag}, {corner weights}
struct NNRecords {
    float4 *rec;            // [b][per][2]; null: no records
    const float *u, *v;     // [b,num] draws of the sampled points (xyz2)
    float coef_sample, coef_other;
    int per, two_sided;     // two_sided: the gt points (xyz1) get a record through their nearest sampled point as well
};

__device__ __forceinline__ float min4(float a, float b, float c, float d)
{
    return fminf(fminf(a, b), fminf(c, d));
}

// Scalar-broadcast variant (no target staging in LDS): every lane of a wave needs the SAME target at the same time, so
// the wave reads its share of the target set with wave-uniform loads -- s_load_dwordx8 through the scalar cache into
// SGPRs, which the VALU takes as operands directly.  The LDS variant above is bound by LDS return bandwidth (a
// broadcast ds_read_b128 still moves 1 KiB into the register file: 302 of them per wave = 24 us per CU at the BASELINE
// shard, more than the 19 us of VALU issue); this one leaves only the VALU work: 34.9 -> 30.6 us for the 8-mesh
// shard, 15.3 -> 13.0 us for one mesh.  Groups of 16 targets share one running-minimum update.
constexpr int NNS_GROUP = 16;
constexpr int NNS_WAVES = 8;  // measured at the BASELINE shard: 4 waves 32.3 us, 8 waves 30.6 us, 16 waves 37.5 us
constexpr int NNS_THREADS = NNS_WAVES * GEOM_WAVE;

template <bool FMA>
__device__ __forceinline__ float nn_sqdist(float tx, float ty, float tz, float qx, float qy, float qz)
{
    return FMA ? geom::sqdist3_fma(tx, ty, tz, qx, qy, qz) : geom::sqdist3(tx, ty, tz, qx, qy, qz);
}

// FMA = the contracted arithmetic of GEOM_FLAG_NN_FMA (6 lane-ops per pair instead of 8)
template <bool FMA>
__device__ __forceinline__ void nn_scalar_body(const NNJob &job, int bid, const NNRecords &rr)
{
    __shared__ float part_d[NNS_WAVES][NN_QUERIES];
    __shared__ int part_i[NNS_WAVES][NN_QUERIES];

    const int longer = job.n > job.m ? job.n : job.m;
    int jobid, qtile;
    if (!geom::xcd_assign(bid, 2 * job.b, (longer + NN_QUERIES - 1) / NN_QUERIES, jobid, qtile)) return;
    const int dir = jobid / job.b;
    const int mesh = jobid - dir * job.b;
    const int nq = dir ? job.m : job.n;
    const int nt = dir ? job.n : job.m;
    const int q0 = qtile * NN_QUERIES;
    if (q0 >= nq) return;

    const float *Q = (dir ? job.xyz2 : job.xyz1) + (size_t)mesh * nq * 3;
    const float *__restrict__ T = (dir ? job.xyz1 : job.xyz2) + (size_t)mesh * nt * 3;
    float *out_d = (dir ? job.dist2 : job.dist1) + (size_t)mesh * nq;
    int *out_i = (dir ? job.idx2 : job.idx1) + (size_t)mesh * nq;

    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = q0 + lane;
    const bool live = q < nq;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        qx = Q[3 * q + 0];
        qy = Q[3 * q + 1];
        qz = Q[3 * q + 2];
    }

    // this wave's contiguous share of the targets, in whole groups; the last wave also takes the ragged tail
    const int groups = nt / NNS_GROUP;
    const int per_wave = (groups + NNS_WAVES - 1) / NNS_WAVES;
    const int g_begin = min(groups, wave * per_wave), g_end = min(groups, g_begin + per_wave);

    float best = INFINITY;
    int best_grp = -1;
#pragma unroll 2
    for (int g = g_begin; g < g_end; ++g) {
        const float *__restrict__ tp = T + (size_t)g * (3 * NNS_GROUP); // wave-uniform address: scalar loads
        float t[3 * NNS_GROUP];
#pragma unroll
        for (int i = 0; i < 3 * NNS_GROUP; ++i) t[i] = tp[i];
        float d[NNS_GROUP];
#pragma unroll
        for (int k = 0; k < NNS_GROUP; ++k) d[k] = nn_sqdist<FMA>(t[3 * k], t[3 * k + 1], t[3 * k + 2], qx, qy, qz);
        float m8 = INFINITY;
#pragma unroll
        for (int k = 0; k < NNS_GROUP; k += 4) m8 = fminf(m8, min4(d[k], d[k + 1], d[k + 2], d[k + 3])); // NaNs drop out
        if (m8 < best) {
            best = m8;
            best_grp = g;
        }
    }
    // exact first index inside the winning group (same arithmetic => exact equality)
    int best_idx = INT_MAX;
    if (best_grp >= 0) {
        const int k0 = best_grp * NNS_GROUP;
#pragma unroll
        for (int j = NNS_GROUP - 1; j >= 0; --j) {
            const int k = k0 + j;
            const float dd = nn_sqdist<FMA>(T[3 * k + 0], T[3 * k + 1], T[3 * k + 2], qx, qy, qz);
            if (dd == best) best_idx = k;
        }
    }
    if (wave == NNS_WAVES - 1) { // ragged tail (nt % 8 targets), after every group in index order
        for (int k = groups * NNS_GROUP; k < nt; ++k) {
            const float dd = nn_sqdist<FMA>(T[3 * k + 0], T[3 * k + 1], T[3 * k + 2], qx, qy, qz);
            if (dd < best) {
                best = dd;
                best_idx = k;
            }
        }
    }
    part_d[wave][lane] = best;
    part_i[wave][lane] = best_idx;
    __syncthreads();

    if (wave == 0 && live) {
        float acc_d = part_d[0][lane];
        int acc_i = part_i[0][lane];
#pragma unroll
        for (int w = 1; w < NNS_WAVES; ++w) {
            const float dd = part_d[w][lane];
            const int ii = part_i[w][lane];
            if (geom::lex_less(dd, ii, acc_d, acc_i)) {
                acc_d = dd;
                acc_i = ii;
            }
        }
        const float d_first = nn_sqdist<FMA>(T[0], T[1], T[2], qx, qy, qz);
        if (d_first != d_first || acc_i == INT_MAX) { // NaN seed sticks; nothing finite keeps the seed
            acc_d = d_first;
            acc_i = 0;
        }
        out_d[q] = acc_d;
        out_i[q] = acc_i;
        if (rr.rec) { // surface-loss record of this point: (sampled point - gt partner) * coef and the sample's corner weights
            const float tx = T[3 * acc_i + 0], ty = T[3 * acc_i + 1], tz = T[3 * acc_i + 2];
            if (dir) { // query = sampled point q of the mesh, partner = its nearest gt point
                const size_t sp = (size_t)mesh * nq + q;
                const float u = rr.u[sp], v = rr.v[sp];
                float4 *r = rr.rec + 2 * ((size_t)mesh * rr.per + q);
                r[0] = make_float4((qx - tx) * rr.coef_sample, (qy - ty) * rr.coef_sample, (qz - tz) * rr.coef_sample, 0.f);
                r[1] = make_float4(1.f - u, u * (1.f - v), u * v, 0.f);
            } else if (rr.two_sided) { // query = gt point, partner = its nearest sampled point (whose face it pulls on)
                const size_t sp = (size_t)mesh * nt + acc_i;
                const float u = rr.u[sp], v = rr.v[sp];
                float4 *r = rr.rec + 2 * ((size_t)mesh * rr.per + nt + q);
                r[0] = make_float4((tx - qx) * rr.coef_other, (ty - qy) * rr.coef_other, (tz - qz) * rr.coef_other, 0.f);
                r[1] = make_float4(1.f - u, u * (1.f - v), u * v, 0.f);
            }
        }
    }
}


} // namespace
