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

typedef float nn_f2 __attribute__((ext_vector_type(2)));

// FMA = the contracted arithmetic of GEOM_FLAG_NN_FMA (6 lane-ops per pair instead of 8).
// Measured and not taken (round 3): the subtractions and squares issued two components at a time from the SGPR pairs
// (v_pk_add_f32 / v_pk_mul_f32: same bits, 14.7 M instead of 21.4 M VALU instructions per 8-mesh launch) -- 32.1 us
// against 31.3: the packed instructions issue at about half the rate of the scalar ones here.
template <bool FMA>
__device__ __forceinline__ void nn_scalar_body(const NNJob &job, int bid, const NNRecords &rr)
{
    __shared__ float part_d[NNS_WAVES][NN_QUERIES];
    __shared__ int part_i[NNS_WAVES][NN_QUERIES];

    const int longer = job.n > job.m ? job.n : job.m;
    int jobid, qtile;
    if (!geom::xcd_assign(bid, 2 * job.b, (longer + NN_QUERIES - 1) / NN_QUERIES, jobid, qtile)) return;
    const int dir = geom::nn_job_dir(jobid, job.b);
    const int mesh = jobid % job.b;
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
        geom::store_agent(out_d + q, acc_d); // read by the loss role of the fused scan's finalize tail
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


// ---------------------------------------------------------------------------------------------------------------------
// Culled scan.  Both clouds are handed over in a VISITING ORDER that keeps neighbours in space neighbours in the list
// (perm: visiting position -> original index), as a copy in that order -- whole runs of NNS_GROUP consecutive points stored
// x[16] y[16] z[16], the ragged tail as xyz triples -- with a bounding sphere per run.  A tile is 64 consecutive queries of
// the order, i.e. a compact patch, and the workgroup's eight waves hold the SAME 64 queries (lane <-> query):
//   1. every load that depends on nothing goes out first (queries, the target cloud's run spheres -> LDS, the patch's own
//      run spheres, the first target); the run nearest to the patch centre and its seven neighbours in the order are the
//      SEEDS: wave w evaluates one, the lanes share the best of the eight as their bound;
//   2. patch-level test, lane <-> run, 64 runs per step:  |c_run - c| > r_patch + max_q s_q + R_run  drops a run for every
//      query at once (triangle inequality); the kept runs are dealt to the waves round-robin into per-wave pending lists;
//   3. a wave's pending runs are fetched global -> LDS without passing through registers (LDS-DMA, five runs per
//      instruction, all in flight), then each takes the lanes' own test
//          |q - c_run|^2 > (s_q + R_run)^2  for every lane  =>  no target of the run can beat or tie any lane's best
//      (s_q = sqrt(bound_q) and R_run with the margins of the culled triangle scan, LAB_NOTES.md 5b) and, if some lane admits
//      it, the brute-force scan's arithmetic -- two targets per packed instruction (v_pk_add/mul/fma_f32 are IEEE per
//      component: the same bits) -- group minimum first;
//   4. closing phase, thread <-> (query, member of a run): the minimum over the waves' partial results and the exact
//      original index inside the winning run(s).
// Candidates are ordered lexicographically on (distance, ORIGINAL index) -- an exact tie between two runs takes a rare
// slow path -- so the result is the brute-force scan's bit for bit for ANY visiting order; a bad order only costs speed.
// NaN / inf: every comparison is written so that a NaN falls on the "evaluate" side; a run with a non-finite point has
// R = +inf and is never skipped; a query with a non-finite coordinate ends on the seed rule whatever is evaluated.
struct NNCull {
    const float *xs1, *xs2;    // [b][nn_cull_stride(n)] / [b][nn_cull_stride(m)]: the clouds in visiting order
    const int *perm1, *perm2;  // [b,n] / [b,m] visiting position -> original index; null: identity
    const float4 *sph1, *sph2; // [b, n/16] / [b, m/16]: {centre, effective radius} of the runs of xs1 / xs2
#ifdef NN_CULL_STATS
    unsigned dbg;
#endif
};

#ifdef NN_CULL_STATS // tools/probe only: counters and ablation knobs
__device__ unsigned long long nn_cull_stats[8];
#define NN_STAT(k, v) do { if ((cu.dbg & 8) && lane == 0) atomicAdd(&nn_cull_stats[k], (unsigned long long)(v)); } while (0)
#define NN_DBG(bit) (cu.dbg & (bit))
#else
#define NN_STAT(k, v) do { } while (0)
#define NN_DBG(bit) false
#endif

constexpr int NN_UNKNOWN = INT_MAX - 1; // "original index of the current best not resolved yet"
constexpr int NNC_SPH = 512;            // run spheres staged per pass (8 KiB)
constexpr int NNC_SLOTS = 15;           // runs a wave fetches per batch (192 B each; 5 per LDS-DMA instruction)
constexpr int NNC_RUN_FLOATS = 3 * NNS_GROUP;

// floats per mesh of a visiting-order copy: rows of runs, 16-byte aligned
__host__ __device__ __forceinline__ size_t nn_cull_stride(int n) { return ((size_t)3 * n + 3) & ~(size_t)3; }

// component c of point p in a visiting-order copy of n points
__device__ __forceinline__ size_t nn_cull_at(int n, int p, int c)
{
    const int full = (n / NNS_GROUP) * NNS_GROUP;
    return p < full ? (size_t)(p / NNS_GROUP) * NNC_RUN_FLOATS + c * NNS_GROUP + (p % NNS_GROUP)
                    : (size_t)full * 3 + (size_t)(p - full) * 3 + c;
}

template <bool FMA>
__device__ __forceinline__ nn_f2 nn_sqdist2(nn_f2 tx, nn_f2 ty, nn_f2 tz, float qx, float qy, float qz)
{ // geom::sqdist3 / sqdist3_fma on two targets at once
    const nn_f2 dx = tx - qx, dy = ty - qy, dz = tz - qz;
    if (FMA) return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
    const nn_f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const nn_f2 sum = xx + yy;
    return sum + zz;
}

// the 16 squared distances of one run whose 48 floats (x[16] y[16] z[16]) sit at t48 (LDS or global; 16-byte aligned)
template <bool FMA>
__device__ __forceinline__ void nn_run_distances(const float *t48, float qx, float qy, float qz, float (&d)[NNS_GROUP])
{
    const float4 *t4 = reinterpret_cast<const float4 *>(t48);
    float4 w[NNC_RUN_FLOATS / 4];
#pragma unroll
    for (int i = 0; i < NNC_RUN_FLOATS / 4; ++i) w[i] = t4[i];
#pragma unroll
    for (int i = 0; i < NNS_GROUP / 4; ++i) {
        const nn_f2 lo = nn_sqdist2<FMA>(nn_f2{w[i].x, w[i].y}, nn_f2{w[4 + i].x, w[4 + i].y}, nn_f2{w[8 + i].x, w[8 + i].y}, qx, qy, qz);
        const nn_f2 hi = nn_sqdist2<FMA>(nn_f2{w[i].z, w[i].w}, nn_f2{w[4 + i].z, w[4 + i].w}, nn_f2{w[8 + i].z, w[8 + i].w}, qx, qy, qz);
        d[4 * i + 0] = lo.x, d[4 * i + 1] = lo.y, d[4 * i + 2] = hi.x, d[4 * i + 3] = hi.y;
    }
}

template <bool FMA>
__device__ __forceinline__ int nn_resolve(const float *__restrict__ Ts, const int *__restrict__ permT, int run, float val,
                                          float qx, float qy, float qz)
{ // lowest ORIGINAL index among the run's targets whose distance equals val exactly.  Per lane (runs differ) and only on
  // the rare tie path: deliberately rolled, a handful of registers
    int lowest = INT_MAX;
    if (run < 0) return lowest;
    const float *t = Ts + (size_t)run * NNC_RUN_FLOATS;
#pragma unroll 1
    for (int j = 0; j < NNS_GROUP; ++j) {
        const float dd = nn_sqdist<FMA>(t[j], t[NNS_GROUP + j], t[2 * NNS_GROUP + j], qx, qy, qz);
        const int o = permT ? permT[run * NNS_GROUP + j] : run * NNS_GROUP + j;
        if (dd == val && o < lowest) lowest = o;
    }
    return lowest;
}

// the run's 48 floats, global -> LDS without passing through registers: lanes 0..11 move 16 B each (LDS-DMA: the
// destination is the wave-uniform base + lane * 16).  The caller waits on vmcnt before reading.
__device__ __forceinline__ void nn_fetch_run(const float *__restrict__ run, float *slot, int lane)
{
    if (lane < NNC_RUN_FLOATS / 4)
        __builtin_amdgcn_global_load_lds(run + 4 * lane, (__attribute__((address_space(3))) void *)slot, 16, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ float nn_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// minimum over the wave, wave-uniform: butterflies inside each row of 16 lanes on the DPP path, then the four rows
__device__ __forceinline__ float nn_wave_min(float v)
{
    v = fminf(v, nn_dpp<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fminf(v, nn_dpp<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fminf(v, nn_dpp<0x141>(v)); // row_half_mirror
    v = fminf(v, nn_dpp<0x140>(v)); // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}

// The culled tile's LDS as ONE object, so that a kernel that runs this body beside another one (the fused surface scan: a
// workgroup is EITHER a triangle tile OR a Chamfer tile) can overlay the two in a union -- separate __shared__ arrays of two
// inlined bodies are added up by the compiler (70 KB for the pair = two workgroups per CU; 43 KB overlaid = three).
struct NNCullLds {
    float4 sph[NNC_SPH];
    __attribute__((aligned(16))) float stage[NNS_WAVES][NNC_SLOTS][NNC_RUN_FLOATS];
    float4 pend_sph[NNS_WAVES][NNC_SLOTS + 1]; // the wave's pending runs: sphere, run
    int pend_run[NNS_WAVES][NNC_SLOTS + 1];
    float part_d[NNS_WAVES + 1][NN_QUERIES], seed_d[NNS_WAVES][NN_QUERIES], qp[3][NN_QUERIES]; // + 1: the ragged tail's partial result
    int part_i[NNS_WAVES + 1][NN_QUERIES], part_g[NNS_WAVES][NN_QUERIES];
};

template <bool FMA>
__device__ __forceinline__ void nn_culled_body(const NNJob &job, const NNCull &cu, int bid, const NNRecords &rr, NNCullLds &L)
{
    auto &sph = L.sph;
    auto &stage = L.stage;
    auto &pend_sph = L.pend_sph;
    auto &pend_run = L.pend_run;
    constexpr int PARTS = NNS_WAVES + 1;                  // one partial result per wave + the ragged tail's
    auto &part_d = L.part_d;
    auto &seed_d = L.seed_d;
    auto &qp = L.qp;
    auto &part_i = L.part_i;
    auto &part_g = L.part_g;

    const int longer = job.n > job.m ? job.n : job.m;
    int jobid, qtile;
    if (!geom::xcd_assign(bid, 2 * job.b, (longer + NN_QUERIES - 1) / NN_QUERIES, jobid, qtile)) return;
    const int dir = geom::nn_job_dir(jobid, job.b);
    const int mesh = jobid % job.b;
    const int nq = dir ? job.m : job.n;
    const int nt = dir ? job.n : job.m;
    const int q0 = qtile * NN_QUERIES;
    if (q0 >= nq) return;

    const float *Qs = (dir ? cu.xs2 : cu.xs1) + (size_t)mesh * nn_cull_stride(nq);
    const int *permQ = dir ? cu.perm2 : cu.perm1;
    const float *__restrict__ Ts = (dir ? cu.xs1 : cu.xs2) + (size_t)mesh * nn_cull_stride(nt);
    const int *__restrict__ permT = dir ? cu.perm1 : cu.perm2;
    if (permQ) permQ += (size_t)mesh * nq;
    if (permT) permT += (size_t)mesh * nt;
    const int runs = nt / NNS_GROUP, qruns = nq / NNS_GROUP;
    const float4 *__restrict__ S = (dir ? cu.sph1 : cu.sph2) + (size_t)mesh * runs;
    const float4 *__restrict__ SQ = (dir ? cu.sph2 : cu.sph1) + (size_t)mesh * qruns;
    const float *T0 = (dir ? job.xyz1 : job.xyz2) + (size_t)mesh * nt * 3; // ORIGINAL order: seed rule, records
    float *out_d = (dir ? job.dist2 : job.dist1) + (size_t)mesh * nq;
    int *out_i = (dir ? job.idx2 : job.idx1) + (size_t)mesh * nq;

    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = q0 + lane;
    const bool live = q < nq;
    // every load that depends on nothing goes out first: one round trip for the queries, the run spheres, the patch's
    // own spheres and the first target
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        qx = Qs[nn_cull_at(nq, q, 0)];
        qy = Qs[nn_cull_at(nq, q, 1)];
        qz = Qs[nn_cull_at(nq, q, 2)];
    }
    // the closing phase maps thread t to (query t / 16 [+ 32], member t % 16); results live at the ORIGINAL positions
    int qo[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int qh = q0 + h * (NN_QUERIES / 2) + (int)(threadIdx.x >> 4);
        qo[h] = (qh < nq && permQ) ? permQ[qh] : qh;
    }
    const float f0x = T0[0], f0y = T0[1], f0z = T0[2];
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f); // lanes 0..3: the spheres of the patch's own runs
    const int own_cnt = min(NN_QUERIES / NNS_GROUP, qruns - q0 / NNS_GROUP);
    if (lane < own_cnt) own = SQ[q0 / NNS_GROUP + lane];
    for (int t = threadIdx.x; t < min(NNC_SPH, runs); t += NNS_THREADS) sph[t] = S[t];
    if (wave == 0) qp[0][lane] = qx, qp[1][lane] = qy, qp[2][lane] = qz;
    __syncthreads();
    if (NN_DBG(32)) { out_d[q0] = qx + f0x + own.x + sph[lane].x + qo[0] + qo[1]; return; }

    float best = INFINITY;
    int best_grp = -1, best_orig = INT_MAX;

    // one run (48 floats in LDS, read as wave-uniform broadcasts) through the brute-force arithmetic; candidates are
    // ordered on (distance, original index)
    auto evaluate = [&](const float *t48, int g) {
        float d[NNS_GROUP];
        nn_run_distances<FMA>(t48, qx, qy, qz, d);
        float m8 = INFINITY;
#pragma unroll
        for (int k = 0; k < NNS_GROUP; k += 4) m8 = fminf(m8, min4(d[k], d[k + 1], d[k + 2], d[k + 3])); // NaNs drop out
        if (m8 < best) {
            best = m8;
            best_grp = g;
            best_orig = NN_UNKNOWN;
        } else if (__builtin_amdgcn_ballot_w64(m8 == best && m8 < INFINITY && g != best_grp) != 0ull) {
            if (m8 == best && m8 < INFINITY && g != best_grp) { // an exact tie between two runs (rare): lowest original index
                const int cand = nn_resolve<FMA>(Ts, permT, g, best, qx, qy, qz);
                if (best_orig == NN_UNKNOWN) best_orig = nn_resolve<FMA>(Ts, permT, best_grp, best, qx, qy, qz);
                if (cand < best_orig) {
                    best_orig = cand;
                    best_grp = g;
                }
            }
        }
    };

    if (runs > 0) {
        // patch centre: the mean of its own runs' centres (any point will do: it only picks the seeds); a patch made of
        // the ragged tail alone takes its first query
        float cx = own.x, cy = own.y, cz = own.z;
        cx += nn_dpp<0xB1>(cx), cy += nn_dpp<0xB1>(cy), cz += nn_dpp<0xB1>(cz);
        cx += nn_dpp<0x4E>(cx), cy += nn_dpp<0x4E>(cy), cz += nn_dpp<0x4E>(cz);
        const float inv = own_cnt > 0 ? 1.f / (float)own_cnt : 1.f;
        cx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(own_cnt > 0 ? cx : qx)));
        cy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(own_cnt > 0 ? cy : qy)));
        cz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(own_cnt > 0 ? cz : qz)));
        cx *= inv, cy *= inv, cz *= inv;
        // the run whose centre is nearest to it (one sphere per lane and step)
        float key = INFINITY;
        int kr = 0;
        for (int r = lane; r < runs; r += GEOM_WAVE) {
            const float4 sp = r < NNC_SPH ? sph[r] : S[r];
            const float dx = sp.x - cx, dy = sp.y - cy, dz = sp.z - cz;
            const float k2 = dx * dx + dy * dy + dz * dz;
            if (k2 < key) key = k2, kr = r; // a NaN centre never wins
        }
        const float kmin = nn_wave_min(key);
        const unsigned long long at = __builtin_amdgcn_ballot_w64(key == kmin);
        const int rstar = __builtin_amdgcn_readlane(kr, at ? __builtin_ctzll(at) : 0);
        // seeds: eight neighbouring runs of the visiting order around it, one per wave; the lanes share the best of the eight
        const int lo = max(0, min(rstar - 3, runs - NNS_WAVES));
        if (NN_DBG(64)) { out_d[q0] = qx + f0x + lo + qo[0] + qo[1]; return; }
        if (lo + wave < runs) {
            nn_fetch_run(Ts + (size_t)(lo + wave) * NNC_RUN_FLOATS, &stage[wave][0][0], lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            evaluate(&stage[wave][0][0], lo + wave);
        }
        seed_d[wave][lane] = best;
        __syncthreads();
        float bound = seed_d[0][lane];
#pragma unroll
        for (int w = 1; w < NNS_WAVES; ++w) bound = fminf(bound, seed_d[w][lane]);
        if (NN_DBG(128)) { out_d[q0] = qx + f0x + bound + qo[0] + qo[1]; return; }
        const bool geo = live && qx - qx == 0.f && qy - qy == 0.f && qz - qz == 0.f;
        const unsigned long long geo_mask = __builtin_amdgcn_ballot_w64(geo);
        // s_q >= sqrt of the lane's bound: v_sqrt_f32 is good to 1 ulp, the margins are 2^-10 relative + 2^-12 |q|_1
        const float q1 = (fabsf(qx) + fabsf(qy) + fabsf(qz)) * 0x1p-12f;
        auto slack = [&](float b2) { return __builtin_amdgcn_sqrtf(b2) * (1.f + 0x1p-10f) + q1; };
        float s = slack(fminf(best, bound));
        auto wanted = [&](const float4 sp) { // does ANY query of the patch admit the run?  (NaN anywhere: yes)
            const float dx = sp.x - qx, dy = sp.y - qy, dz = sp.z - qz;
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const float reach = s + sp.w;
            return (__builtin_amdgcn_ballot_w64(!(d2 > reach * reach)) & geo_mask) != 0ull;
        };

        // patch radius and the widest bound of the patch: a run with |c_run - c| > (rq + s_max) + R_run is out of reach of
        // EVERY query of the patch (triangle inequality) -- tested with lane <-> run, 64 runs per step
        float rq = 0.f;
        if (geo) {
            const float dx = qx - cx, dy = qy - cy, dz = qz - cz;
            rq = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
        }
        rq = -nn_wave_min(-rq);
        const float smax = -nn_wave_min(geo ? -s : 0.f);
        const float reach_patch = rq * (1.f + 0x1p-10f) + (fabsf(cx) + fabsf(cy) + fabsf(cz)) * 0x1p-12f + smax; // NaN: keeps all
        int dealt = 0;   // runs kept so far: they are dealt to the waves round-robin
        int pending = 0; // this wave's kept runs not yet evaluated (pend_sph / pend_run)
        // fetch the pending runs global -> LDS (five runs = 60 lanes x 16 B per LDS-DMA instruction, all in flight
        // together), then each goes through the lanes' own, tighter test and, if any lane admits it, the evaluation
        auto flush = [&]() {
            float *slots = &stage[wave][0][0];
            const int piece = lane % (NNC_RUN_FLOATS / 4), which = lane / (NNC_RUN_FLOATS / 4); // which < 5 for lanes < 60
#pragma unroll
            for (int f = 0; f < NNC_SLOTS / 5; ++f) {
                const int k = 5 * f + which;
                if (5 * f < pending && lane < 60 && k < pending)
                    __builtin_amdgcn_global_load_lds(Ts + (size_t)pend_run[wave][k] * NNC_RUN_FLOATS + 4 * piece,
                                                     (__attribute__((address_space(3))) void *)(slots + 5 * f * NNC_RUN_FLOATS), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int k = 0; k < pending; ++k) {
                if (!wanted(pend_sph[wave][k])) continue;
                NN_STAT(2, 1);
                if (NN_DBG(4)) continue;
                evaluate(slots + k * NNC_RUN_FLOATS, __builtin_amdgcn_readfirstlane(pend_run[wave][k]));
                s = slack(fminf(best, bound));
            }
            pending = 0;
        };
        for (int c0 = 0; c0 < runs; c0 += NNC_SPH) {
            if (NN_DBG(1)) break;
            const int len = min(NNC_SPH, runs - c0);
            if (c0) { // the first chunk is staged already
                __syncthreads();
                for (int t = threadIdx.x; t < len; t += NNS_THREADS) sph[t] = S[c0 + t];
                __syncthreads();
            }
            for (int t0 = 0; t0 < len; t0 += GEOM_WAVE) {
                const int t = t0 + lane, g = c0 + t;
                bool keep = false;
                float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < len && !(g >= lo && g < lo + NNS_WAVES)) { // the seeds are done
                    sp = sph[t];
                    const float dx = sp.x - cx, dy = sp.y - cy, dz = sp.z - cz;
                    const float d2 = dx * dx + dy * dy + dz * dz;
                    const float reach = reach_patch + sp.w;
                    keep = !(d2 > reach * reach);
                }
                const unsigned long long kept = __builtin_amdgcn_ballot_w64(keep);
                if (kept == 0ull) continue;
                const int ord = dealt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(kept >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kept, 0u));
                dealt += __builtin_popcountll(kept);
                if (wave == 0) NN_STAT(1, __builtin_popcountll(kept));
                if (NN_DBG(2)) continue;
                // this wave's share: every 8th kept run (at most 8 of a step)
                const bool is_mine = keep && (ord & (NNS_WAVES - 1)) == wave;
                const unsigned long long mine = __builtin_amdgcn_ballot_w64(is_mine);
                const int more = __builtin_popcountll(mine);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0u));
                for (int queued = 0; queued < more;) { // into the pending list, as many as there is room for; flush when full
                    if (pending == NNC_SLOTS) flush();
                    const int take = min(NNC_SLOTS - pending, more - queued);
                    const int k = rank - queued;
                    if (is_mine && k >= 0 && k < take) {
                        pend_sph[wave][pending + k] = sp;
                        pend_run[wave][pending + k] = g;
                    }
                    pending += take;
                    queued += take;
                }
            }
            if (pending) flush(); // before the spheres' chunk... (pend_sph holds copies: only for simplicity)
        }
        if (wave == 0) NN_STAT(0, 1);
    }
    part_d[wave][lane] = best;
    part_i[wave][lane] = best_orig; // INT_MAX: no candidate; NN_UNKNOWN: somewhere in run part_g
    part_g[wave][lane] = best_grp;
    if (wave == NNS_WAVES - 1) { // ragged tail (nt % 16 targets, always evaluated): a partial result of its own
        float tail_d = INFINITY;
        int tail_o = INT_MAX;
        const float *tail = Ts + (size_t)runs * NNC_RUN_FLOATS;
        for (int k = runs * NNS_GROUP; k < nt; ++k, tail += 3) {
            const float dd = nn_sqdist<FMA>(tail[0], tail[1], tail[2], qx, qy, qz);
            const int o = permT ? permT[k] : k;
            if (dd < tail_d || (dd == tail_d && dd < INFINITY && o < tail_o)) tail_d = dd, tail_o = o;
        }
        part_d[NNS_WAVES][lane] = tail_d;
        part_i[NNS_WAVES][lane] = tail_o;
    }
    __syncthreads();
    if (NN_DBG(256)) { out_d[q0] = part_d[0][lane] + f0x + qo[0] + qo[1]; return; }

    // closing phase, thread <-> (query, member of a run): the smallest distance of the eight waves, the lowest original
    // index among the targets that attain it (the members of the winning run(s), one per thread; same arithmetic => exact
    // equality), the seed rule, the outputs
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ql = h * (NN_QUERIES / 2) + (int)(threadIdx.x >> 4), mem = threadIdx.x & (NNS_GROUP - 1);
        const float px = qp[0][ql], py = qp[1][ql], pz = qp[2][ql];
        float pd[PARTS];
#pragma unroll
        for (int w = 0; w < PARTS; ++w) pd[w] = part_d[w][ql];
        float acc_d = pd[0];
#pragma unroll
        for (int w = 1; w < PARTS; ++w) acc_d = fminf(acc_d, pd[w]);
        int first = PARTS, count = 0;
#pragma unroll
        for (int w = PARTS - 1; w >= 0; --w)
            if (pd[w] == acc_d) first = w, ++count;
        int cand = INT_MAX;
        auto member = [&](int w) {
            const int io = part_i[w][ql];
            if (io != NN_UNKNOWN) return io; // (always the case for the tail's partial)
            const int g = part_g[w < NNS_WAVES ? w : 0][ql];
            const float *t = Ts + (size_t)g * NNC_RUN_FLOATS + mem;
            const float dd = nn_sqdist<FMA>(t[0], t[NNS_GROUP], t[2 * NNS_GROUP], px, py, pz);
            const int o = permT ? permT[g * NNS_GROUP + mem] : g * NNS_GROUP + mem;
            return dd == acc_d ? o : INT_MAX;
        };
        if (first < PARTS && acc_d < INFINITY) {
            cand = member(first);
            if (count > 1) { // two partial results hold the same distance (rare)
                for (int w = first + 1; w < PARTS; ++w)
                    if (part_d[w][ql] == acc_d) cand = min(cand, member(w));
            }
        }
        cand = min(cand, __builtin_amdgcn_update_dpp(0, cand, 0xB1, 0xf, 0xf, false));
        cand = min(cand, __builtin_amdgcn_update_dpp(0, cand, 0x4E, 0xf, 0xf, false));
        cand = min(cand, __builtin_amdgcn_update_dpp(0, cand, 0x141, 0xf, 0xf, false));
        cand = min(cand, __builtin_amdgcn_update_dpp(0, cand, 0x140, 0xf, 0xf, false));
        if (mem == 0 && q0 + ql < nq) {
            int acc_i = cand;
            const float d_first = nn_sqdist<FMA>(f0x, f0y, f0z, px, py, pz);
            if (d_first != d_first || acc_i == INT_MAX) { // NaN seed sticks; nothing finite keeps the seed
                acc_d = d_first;
                acc_i = 0;
            }
            const int qq = qo[h];
            geom::store_agent(out_d + qq, acc_d); // read by the loss role of the fused scan's finalize tail
            out_i[qq] = acc_i;
            if (rr.rec) { // surface-loss record of this point (see nn_scalar_body)
                const float tx = T0[3 * acc_i + 0], ty = T0[3 * acc_i + 1], tz = T0[3 * acc_i + 2];
                if (dir) {
                    const size_t sp = (size_t)mesh * nq + qq;
                    const float u = rr.u[sp], v = rr.v[sp];
                    float4 *r = rr.rec + 2 * ((size_t)mesh * rr.per + qq);
                    r[0] = make_float4((px - tx) * rr.coef_sample, (py - ty) * rr.coef_sample, (pz - tz) * rr.coef_sample, 0.f);
                    r[1] = make_float4(1.f - u, u * (1.f - v), u * v, 0.f);
                } else if (rr.two_sided) {
                    const size_t sp = (size_t)mesh * nt + acc_i;
                    const float u = rr.u[sp], v = rr.v[sp];
                    float4 *r = rr.rec + 2 * ((size_t)mesh * rr.per + nt + qq);
                    r[0] = make_float4((tx - px) * rr.coef_other, (ty - py) * rr.coef_other, (tz - pz) * rr.coef_other, 0.f);
                    r[1] = make_float4(1.f - u, u * (1.f - v), u * v, 0.f);
                }
            }
        }
    }
}

// visiting-order copy of a cloud + the bounding sphere of every run of NNS_GROUP consecutive points: the thread holds
// point p of the order (p >= n: none); the 16 threads of a run are consecutive lanes of one wave, all present
__device__ __forceinline__ void nn_cull_emit(float px, float py, float pz, int n, int p, float *xs, float4 *sph)
{
    if (p < n) xs[nn_cull_at(n, p, 0)] = px, xs[nn_cull_at(n, p, 1)] = py, xs[nn_cull_at(n, p, 2)] = pz;
    const int runs = n / NNS_GROUP;
    float cx = px, cy = py, cz = pz;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) cx += __shfl_xor(cx, off), cy += __shfl_xor(cy, off), cz += __shfl_xor(cz, off);
    cx *= 1.f / NNS_GROUP, cy *= 1.f / NNS_GROUP, cz *= 1.f / NNS_GROUP;
    const float dx = px - cx, dy = py - cy, dz = pz - cz;
    float r2 = dx * dx + dy * dy + dz * dz;
    const bool finite = px - px == 0.f && py - py == 0.f && pz - pz == 0.f;
    int bad = finite ? 0 : 1;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, off)), bad |= __shfl_xor(bad, off);
    const int run = p / NNS_GROUP;
    if ((p % NNS_GROUP) == 0 && run < runs) {
        const float r = sqrtf(r2);
        float reff = r * (1.f + 0x1p-10f) + 0x1p-12f * (fabsf(cx) + fabsf(cy) + fabsf(cz) + r);
        if (bad || !(reff == reff)) reff = INFINITY; // a non-finite member: the run is never skipped
        sph[run] = make_float4(cx, cy, cz, reff);
    }
}

__device__ __forceinline__ void nn_cull_prep_point(const float *x, const int *perm, int n, int p, float *xs, float4 *sph)
{
    float px = 0.f, py = 0.f, pz = 0.f;
    if (p < n) {
        const int k = perm ? perm[p] : p;
        px = x[3 * k + 0], py = x[3 * k + 1], pz = x[3 * k + 2];
    }
    nn_cull_emit(px, py, pz, n, p, xs, sph);
}

} // namespace
