// Chamfer brute-force nearest neighbour for gfx950, both directions in ONE launch.
//
// Two kernels: chamfer_nn_scalar_kernel (the default: targets arrive as wave-uniform scalar loads, no LDS staging --
// see its comment further down) and chamfer_nn_kernel<TRUNC> (targets staged in LDS; used for the reference-tail-
// truncation mode, whose per-target skip rule wants the staging pass).  The design notes below describe the LDS
// kernel; job mapping, arithmetic, merge and seed rule are shared.
//
// Replaces ChamferDistanceKernel + launcher of the reference
// (chamfer_distance/chamfer_distance.cu:6-55, 57-73: two <<<(32,16),512>>> launches where
// only ceil(n/512) blocks per mesh do any work, a scalar 3-float LDS read per pair and
// a global read-modify-write per 512-target tile).
//
// Design (written for CDNA4, not translated):
//   * one workgroup = 8 waves = 64 query points of one (direction, mesh) job; lane <-> query,
//     so every lane of a wave looks at the SAME target at the same time and the target
//     tile is read from LDS as wave-uniform (broadcast) ds_read_b128s: 3 reads serve
//     4 targets x 64 queries.
//   * the target set is staged in LDS in sweeps of up to 4096 points (the whole set at the
//     BASELINE size: one coalesced sweep, one barrier) and the 8 waves split it eight ways, which
//     gives b * (ceil(n/64)+ceil(m/64)) * 8 waves (6016 at the BASELINE shard of 8 meshes)
//     instead of the reference's 6 busy blocks per mesh.
//   * the arg-min keeps only a running minimum per GROUP of 4 targets in the hot loop
//     (v_min3/v_min + one compare + two selects per 4 pairs instead of a compare and two
//     selects per pair); the winning group's 4 distances are recomputed once at the end
//     to recover the exact first-minimum index.
//   * partial results of the 8 waves are merged lexicographically on (distance, index),
//     which equals the reference's sequential strict-'<' scan; the "first target seeds"
//     rule (NaN seed sticks, all-inf keeps index 0) is applied explicitly.
//   * arithmetic: (dx*dx + dy*dy) + dz*dz with dx = target - query, un-fused
//     (-ffp-contract=off) -- bit-identical to the reference CPU nnsearch (my_lib.c:13-16).
//
// The pair scan is FP32-VALU bound (~9.3 lane-ops per pair); HBM traffic is the
// compulsory 20 B per point.  See LAB_NOTES.md section 4.  (Measured and rejected: packed v_pk_add_f32 /
// v_pk_mul_f32 on two targets per lane with an x|y|z-per-group LDS layout -- bit-identical results, 96 packed
// instructions in the loop, but 38.6 us instead of 34.8: on gfx950 the packed ops do not issue at twice the
// scalar rate for this instruction mix.)
#include "geom_common.h"
#include "nn_scan.h"

namespace {

constexpr int NN_THREADS = 512;
constexpr int NN_WAVES = NN_THREADS / GEOM_WAVE; // 8
constexpr int NN_CHUNK = 4096;                   // targets staged in LDS per sweep (48 KiB; multiple of 4 * NN_WAVES)



template <bool TRUNC>
__global__ __launch_bounds__(NN_THREADS) void chamfer_nn_kernel(NNJob job)
{
    __shared__ float4 tile[NN_CHUNK * 3 / 4]; // 48 KiB: [target][xyz] packed, read 4 targets per 3 float4
    __shared__ float part_d[NN_WAVES][NN_QUERIES];
    __shared__ int part_i[NN_WAVES][NN_QUERIES];

    // (direction, mesh) jobs are pinned to XCDs so a job's target set is fetched into one L2 only
    const int longer = job.n > job.m ? job.n : job.m;
    int jobid, qtile;
    if (!geom::xcd_assign(blockIdx.x, 2 * job.b, (longer + NN_QUERIES - 1) / NN_QUERIES, jobid, qtile)) return;
    const int dir = geom::nn_job_dir(jobid, job.b);
    const int mesh = jobid % job.b;
    const int nq = dir ? job.m : job.n;
    const int nt = dir ? job.n : job.m;
    const int q0 = qtile * NN_QUERIES;
    if (q0 >= nq) return; // tiles are sized for the longer direction

    const float *Q = (dir ? job.xyz2 : job.xyz1) + (size_t)mesh * nq * 3;
    const float *T = (dir ? job.xyz1 : job.xyz2) + (size_t)mesh * nt * 3;
    float *out_d = (dir ? job.dist2 : job.dist1) + (size_t)mesh * nq;
    int *out_i = (dir ? job.idx2 : job.idx1) + (size_t)mesh * nq;

    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int q = q0 + lane;
    const bool live = q < nq;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        qx = Q[3 * q + 0];
        qy = Q[3 * q + 1];
        qz = Q[3 * q + 2];
    }

    float best = INFINITY; // running minimum over this wave's share of the targets
    int best_grp = -1;     // global group id (= first target index / 4) that produced it

    float *tile_f = reinterpret_cast<float *>(tile);
    for (int c0 = 0; c0 < nt; c0 += NN_CHUNK) {
        const int len = min(NN_CHUNK, nt - c0);
        const int groups = (len + 3) >> 2;
        // cooperative, coalesced stage; pad the last group with +inf (never '<' anything)
        for (int i = threadIdx.x; i < groups * 12; i += NN_THREADS) {
            const int k = i / 3;
            float v = INFINITY;
            if (k < len && !(TRUNC && geom::ref_tail_skipped(c0 + k, nt))) v = T[(size_t)c0 * 3 + i];
            tile_f[i] = v;
        }
        __syncthreads();

        const int per_wave = (groups + NN_WAVES - 1) / NN_WAVES;
        const int g_begin = wave * per_wave;
        const int g_end = min(groups, g_begin + per_wave);
        const int grp_base = c0 >> 2;
#pragma unroll 2
        for (int g = g_begin; g < g_end; ++g) {
            const float4 a = tile[3 * g + 0];
            const float4 b = tile[3 * g + 1];
            const float4 c = tile[3 * g + 2];
            const float d0 = geom::sqdist3(a.x, a.y, a.z, qx, qy, qz);
            const float d1 = geom::sqdist3(a.w, b.x, b.y, qx, qy, qz);
            const float d2 = geom::sqdist3(b.z, b.w, c.x, qx, qy, qz);
            const float d3 = geom::sqdist3(c.y, c.z, c.w, qx, qy, qz);
            const float m4 = min4(d0, d1, d2, d3); // NaNs drop out, like a false '<'
            if (m4 < best) {
                best = m4;
                best_grp = grp_base + g;
            }
        }
        __syncthreads();
    }

    // recover the exact first index inside the winning group (same arithmetic => exact equality)
    int best_idx = INT_MAX;
    if (best_grp >= 0) {
        const int k0 = best_grp << 2;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int k = k0 + j;
            if (k < nt && !(TRUNC && geom::ref_tail_skipped(k, nt))) {
                const float d = geom::sqdist3(T[3 * k + 0], T[3 * k + 1], T[3 * k + 2], qx, qy, qz);
                if (d == best) best_idx = k;
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
        for (int w = 1; w < NN_WAVES; ++w) {
            const float d = part_d[w][lane];
            const int i = part_i[w][lane];
            if (geom::lex_less(d, i, acc_d, acc_i)) {
                acc_d = d;
                acc_i = i;
            }
        }
        // "k == 0 ||" seed of the sequential scan (my_lib.c:17, chamfer_distance.cu:39)
        const float d_first = geom::sqdist3(T[0], T[1], T[2], qx, qy, qz);
        if (d_first != d_first || acc_i == INT_MAX) { // NaN seed sticks; nothing finite keeps the seed
            acc_d = d_first;
            acc_i = 0;
        }
        if (TRUNC) {
            // Q3: a final reference tile shorter than 4 targets scans nothing and merges (0.0, 0)
            const int last0 = ((nt - 1) / geom::REF_TILE) * geom::REF_TILE;
            if (nt - last0 < 4 && (last0 == 0 || acc_d > 0.f)) {
                acc_d = 0.f;
                acc_i = 0;
            }
        }
        out_d[q] = acc_d;
        out_i[q] = acc_i;
    }
}


template <bool FMA>
__global__ __launch_bounds__(NNS_THREADS) void chamfer_nn_scalar_kernel(NNJob job)
{
    nn_scalar_body<FMA>(job, blockIdx.x, NNRecords{nullptr, nullptr, nullptr, 0.f, 0.f, 0, 0});
}

template <bool FMA>
__global__ __launch_bounds__(NNS_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) void chamfer_nn_culled_kernel(NNJob job, NNCull cu)
{
    __shared__ NNCullLds lds;
    nn_culled_body<FMA>(job, cu, blockIdx.x, NNRecords{nullptr, nullptr, nullptr, 0.f, 0.f, 0, 0}, lds);
}

__global__ __launch_bounds__(256) void nn_cull_index_kernel(int n, const float *xyz, const int *order, float *xs, float4 *sph)
{
    const int mesh = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x; // whole waves run: the run spheres are 16-lane reductions
    nn_cull_prep_point(xyz + (size_t)mesh * n * 3, order ? order + (size_t)mesh * n : nullptr, n, p, xs + (size_t)mesh * nn_cull_stride(n),
                       sph + (size_t)mesh * (n / NNS_GROUP));
}

} // namespace

extern "C" int geom_chamfer_nn_f32(int b, int n, const float *xyz, int m, const float *xyz2,
                                   float *result, int *result_i, float *result2, int *result2_i,
                                   unsigned flags, void *stream)
{
    if (b < 0 || n < 0 || m < 0) return GEOM_EINVAL;
    if (b == 0 || (n == 0 && m == 0)) return 0;
    if (n == 0 || m == 0) return GEOM_EINVAL; // a direction with no targets has no arg-min
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i) return GEOM_EINVAL;
    NNJob job{xyz, xyz2, result, result2, result_i, result2_i, b, n, m};
    const int longer = n > m ? n : m;
    const int64_t blocks = (int64_t)geom::NUM_XCD * ((longer + NN_QUERIES - 1) / NN_QUERIES) * ((2 * (int64_t)b + 7) / 8);
    if (blocks > 0x7fffffffLL) return GEOM_ETOOBIG;
    dim3 grid(geom::xcd_grid(2 * b, (longer + NN_QUERIES - 1) / NN_QUERIES), 1, 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((flags & GEOM_FLAG_REF_TAIL_TRUNC) && (flags & GEOM_FLAG_NN_FMA)) return GEOM_EINVAL; // one arithmetic per quirk mode
    if (flags & GEOM_FLAG_REF_TAIL_TRUNC)
        hipLaunchKernelGGL(chamfer_nn_kernel<true>, grid, dim3(NN_THREADS), 0, s, job);
    else if (flags & GEOM_FLAG_NN_FMA)
        hipLaunchKernelGGL(chamfer_nn_scalar_kernel<true>, grid, dim3(NNS_THREADS), 0, s, job);
    else
        hipLaunchKernelGGL(chamfer_nn_scalar_kernel<false>, grid, dim3(NNS_THREADS), 0, s, job);
    return geom::launch_status();
}

/* the index of one cloud for the culled scan: [b][n/16] run spheres (float4), then [b][nn_cull_stride(n)] floats of the
 * cloud in visiting order */
extern "C" int64_t geom_nn_cull_index_floats(int b, int n)
{
    if (b <= 0 || n <= 0) return 0;
    return (int64_t)b * (4 * ((int64_t)n / NNS_GROUP) + (int64_t)nn_cull_stride(n));
}

extern "C" int geom_nn_cull_index_f32(int b, int n, const float *xyz, const int *order, float *index, void *stream)
{
    if (b < 0 || n < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (!xyz || !index || ((uintptr_t)index & 15)) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    float4 *sph = reinterpret_cast<float4 *>(index);
    hipLaunchKernelGGL(nn_cull_index_kernel, dim3((n + 255) / 256, b), dim3(256), 0, static_cast<hipStream_t>(stream), n, xyz, order,
                       reinterpret_cast<float *>(sph + (size_t)b * (n / NNS_GROUP)), sph);
    return geom::launch_status();
}

/* workspace floats of geom_chamfer_nn_culled_f32: the indices of both clouds */
extern "C" int64_t geom_chamfer_nn_culled_workspace_floats(int b, int n, int m)
{
    return geom_nn_cull_index_floats(b, n) + geom_nn_cull_index_floats(b, m);
}

extern "C" int geom_chamfer_nn_culled_f32(int b, int n, const float *xyz, int m, const float *xyz2, const int *order1,
                                          const int *order2, float *result, int *result_i, float *result2, int *result2_i,
                                          unsigned flags, float *workspace, void *stream)
{
    if (b < 0 || n < 0 || m < 0) return GEOM_EINVAL;
    if (b == 0 || (n == 0 && m == 0)) return 0;
    if (n == 0 || m == 0) return GEOM_EINVAL;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i || !workspace || ((uintptr_t)workspace & 15)) return GEOM_EINVAL;
#ifndef NN_CULL_STATS
    if (flags & ~GEOM_FLAG_NN_FMA) return GEOM_EUNSUPPORTED;
#endif
    if (b > 65535) return GEOM_ETOOBIG;
    NNJob job{xyz, xyz2, result, result2, result_i, result2_i, b, n, m};
    float *index1 = workspace, *index2 = workspace + geom_nn_cull_index_floats(b, n);
    int rc = geom_nn_cull_index_f32(b, n, xyz, order1, index1, stream);
    if (rc == 0) rc = geom_nn_cull_index_f32(b, m, xyz2, order2, index2, stream);
    if (rc != 0) return rc;
    const float4 *sph1 = reinterpret_cast<const float4 *>(index1), *sph2 = reinterpret_cast<const float4 *>(index2);
    const float *xs1 = reinterpret_cast<const float *>(sph1 + (size_t)b * (n / NNS_GROUP));
    const float *xs2 = reinterpret_cast<const float *>(sph2 + (size_t)b * (m / NNS_GROUP));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int longer = n > m ? n : m;
#ifdef NN_CULL_STATS
    NNCull cu{xs1, xs2, order1, order2, sph1, sph2, flags >> 16};
    flags &= 0xffff;
#else
    NNCull cu{xs1, xs2, order1, order2, sph1, sph2};
#endif
    const int64_t blocks = (int64_t)geom::NUM_XCD * ((longer + NN_QUERIES - 1) / NN_QUERIES) * ((2 * (int64_t)b + 7) / 8);
    if (blocks > 0x7fffffffLL) return GEOM_ETOOBIG;
    dim3 grid(geom::xcd_grid(2 * b, (longer + NN_QUERIES - 1) / NN_QUERIES), 1, 1);
    if (flags & GEOM_FLAG_NN_FMA) hipLaunchKernelGGL(chamfer_nn_culled_kernel<true>, grid, dim3(NNS_THREADS), 0, s, job, cu);
    else hipLaunchKernelGGL(chamfer_nn_culled_kernel<false>, grid, dim3(NNS_THREADS), 0, s, job, cu);
    return geom::launch_status();
}

#ifdef NN_CULL_STATS
extern "C" void geom_cull_stats(unsigned long long *out, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(nn_cull_stats), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(nn_cull_stats), z, sizeof(z));
    }
}
#endif
