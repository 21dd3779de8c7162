// Point-to-triangle arg-min scan for gfx950.
//
// Replaces TriDistanceKernel + launcher of the reference (tri_distance/tri_distance.cu:94-211,
// 213-228).  Same workgroup shape as the Chamfer scan (chamfer_nn.hip): 64 query points per
// workgroup, lane <-> query, the 4 waves split each LDS chunk of triangles four ways, and
// the partial (distance, triangle, region) results are merged lexicographically -- equal to
// the reference's sequential strict-'<' scan, with the "first triangle seeds" rule applied
// explicitly.  Triangle corners are staged in LDS once per chunk as 3 x float4 records and
// read wave-uniformly; in the INDEXED variant the corners are gathered from verts through
// faces while staging, so the three [b,F,3] corner arrays the reference materialises
// (utils.py:467-469) never exist.
//
// Arithmetic: tri_math.h (literal operation order of tri_distance.cu:140-191).
#include "geom_common.h"
#include "tri_math.h"

namespace {

using geom::V3;

constexpr int TRI_THREADS = 256;
constexpr int TRI_WAVES = TRI_THREADS / GEOM_WAVE;
constexpr int TRI_QUERIES = GEOM_WAVE;
constexpr int TRI_CHUNK = 512; // triangles staged per pass: 512 * 48 B = 24 KiB

struct TriJob {
    const float *xyz;                 // [b,n,3]
    const float *tri1, *tri2, *tri3;  // [b,m,3] (direct variant)
    const float *verts;               // [b,nv,3] (indexed variant)
    const int64_t *faces;             // [m,3]
    float *dist;
    int *point, *index;
    int b, n, m, nv;
};

__device__ __forceinline__ V3 load3(const float *p) { return geom::mk(p[0], p[1], p[2]); }

template <bool INDEXED>
__device__ __forceinline__ void fetch_triangle(const TriJob &job, int mesh, int k, V3 &A, V3 &B, V3 &C)
{
    if (INDEXED) {
        const float *V = job.verts + (size_t)mesh * job.nv * 3;
        A = load3(V + 3 * job.faces[3 * (size_t)k + 0]);
        B = load3(V + 3 * job.faces[3 * (size_t)k + 1]);
        C = load3(V + 3 * job.faces[3 * (size_t)k + 2]);
    } else {
        const size_t o = ((size_t)mesh * job.m + k) * 3;
        A = load3(job.tri1 + o);
        B = load3(job.tri2 + o);
        C = load3(job.tri3 + o);
    }
}

template <bool INDEXED, bool TRUNC, bool FIX6>
__global__ __launch_bounds__(TRI_THREADS) void tri_distance_kernel(TriJob job)
{
    // record = {A.xyz, skip}, {B.xyz, -}, {C.xyz, -}
    __shared__ float4 tile[TRI_CHUNK * 3];
    __shared__ float part_d[TRI_WAVES][TRI_QUERIES];
    __shared__ int part_k[TRI_WAVES][TRI_QUERIES];

    const int mesh = blockIdx.y;
    const int q0 = blockIdx.x * TRI_QUERIES;
    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int q = q0 + lane;
    const bool live = q < job.n;
    V3 p = geom::mk(0.f, 0.f, 0.f);
    if (live) p = load3(job.xyz + ((size_t)mesh * job.n + q) * 3);

    float best = INFINITY;
    int best_key = INT_MAX; // (triangle << 3) | region

    for (int c0 = 0; c0 < job.m; c0 += TRI_CHUNK) {
        const int len = min(TRI_CHUNK, job.m - c0);
        for (int t = threadIdx.x; t < len; t += TRI_THREADS) {
            V3 A, B, C;
            fetch_triangle<INDEXED>(job, mesh, c0 + t, A, B, C);
            const float skip = (TRUNC && geom::ref_tail_skipped(c0 + t, job.m)) ? 1.f : 0.f;
            tile[3 * t + 0] = make_float4(A.x, A.y, A.z, skip);
            tile[3 * t + 1] = make_float4(B.x, B.y, B.z, 0.f);
            tile[3 * t + 2] = make_float4(C.x, C.y, C.z, 0.f);
        }
        __syncthreads();

        const int per_wave = (len + TRI_WAVES - 1) / TRI_WAVES;
        const int t_begin = wave * per_wave;
        const int t_end = min(len, t_begin + per_wave);
        for (int t = t_begin; t < t_end; ++t) {
            const float4 ra = tile[3 * t + 0];
            const float4 rb = tile[3 * t + 1];
            const float4 rc = tile[3 * t + 2];
            if (TRUNC && ra.w != 0.f) continue; // wave-uniform
            int opt;
            const float d = geom::tri_pair_literal<FIX6>(p, geom::mk(ra.x, ra.y, ra.z), geom::mk(rb.x, rb.y, rb.z),
                                                         geom::mk(rc.x, rc.y, rc.z), opt);
            if (d < best) {
                best = d;
                best_key = ((c0 + t) << 3) | opt;
            }
        }
        __syncthreads();
    }

    part_d[wave][lane] = best;
    part_k[wave][lane] = best_key;
    __syncthreads();

    if (wave == 0 && live) {
        float acc_d = part_d[0][lane];
        int acc_k = part_k[0][lane];
#pragma unroll
        for (int w = 1; w < TRI_WAVES; ++w) {
            const float d = part_d[w][lane];
            const int k = part_k[w][lane];
            if (geom::lex_less(d, k, acc_d, acc_k)) {
                acc_d = d;
                acc_k = k;
            }
        }
        // "k == 0 ||" seed (tri_distance.cu:194): a NaN first distance sticks, and when
        // nothing compares below +inf the first triangle's result stands.
        V3 A, B, C;
        fetch_triangle<INDEXED>(job, mesh, 0, A, B, C);
        int opt0;
        const float d0 = geom::tri_pair_literal<FIX6>(p, A, B, C, opt0);
        if (d0 != d0 || acc_k == INT_MAX) {
            acc_d = d0;
            acc_k = opt0;
        }
        if (TRUNC) {
            // Q3: a final reference tile shorter than 4 triangles scans nothing and merges its
            // initial (10000, region 0, index 0)  (tri_distance.cu:127-131, 202-206)
            const int last0 = ((job.m - 1) / geom::REF_TILE) * geom::REF_TILE;
            if (job.m - last0 < 4 && (last0 == 0 || acc_d > 10000.f)) {
                acc_d = 10000.f;
                acc_k = 0;
            }
        }
        const size_t o = (size_t)mesh * job.n + q;
        job.dist[o] = acc_d;
        job.point[o] = acc_k & 7;
        job.index[o] = acc_k >> 3;
    }
}

template <bool INDEXED>
int launch_tri(const TriJob &job, unsigned flags, void *stream)
{
    dim3 grid((job.n + TRI_QUERIES - 1) / TRI_QUERIES, job.b, 1);
    dim3 block(TRI_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool trunc = flags & GEOM_FLAG_REF_TAIL_TRUNC, fix6 = flags & GEOM_FLAG_FIX_REGION6;
    if (trunc && fix6)
        hipLaunchKernelGGL((tri_distance_kernel<INDEXED, true, true>), grid, block, 0, s, job);
    else if (trunc)
        hipLaunchKernelGGL((tri_distance_kernel<INDEXED, true, false>), grid, block, 0, s, job);
    else if (fix6)
        hipLaunchKernelGGL((tri_distance_kernel<INDEXED, false, true>), grid, block, 0, s, job);
    else
        hipLaunchKernelGGL((tri_distance_kernel<INDEXED, false, false>), grid, block, 0, s, job);
    return geom::launch_status();
}

} // namespace

extern "C" int geom_tri_distance_f32(int b, int n, const float *xyz, int m,
                                     const float *tri1, const float *tri2, const float *tri3,
                                     float *dist, int *point, int *index, unsigned flags, void *stream)
{
    if (b < 0 || n < 0 || m < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (m == 0) return GEOM_EINVAL;
    if (!xyz || !tri1 || !tri2 || !tri3 || !dist || !point || !index) return GEOM_EINVAL;
    if (b > 65535 || m >= (1 << 28)) return GEOM_ETOOBIG;
    TriJob job{xyz, tri1, tri2, tri3, nullptr, nullptr, dist, point, index, b, n, m, 0};
    return launch_tri<false>(job, flags, stream);
}

extern "C" int geom_tri_distance_indexed_f32(int b, int n, const float *xyz, int nv, const float *verts,
                                             int nf, const int64_t *faces,
                                             float *dist, int *point, int *index, unsigned flags, void *stream)
{
    if (b < 0 || n < 0 || nf < 0 || nv < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (nf == 0 || nv == 0) return GEOM_EINVAL;
    if (!xyz || !verts || !faces || !dist || !point || !index) return GEOM_EINVAL;
    if (b > 65535 || nf >= (1 << 28)) return GEOM_ETOOBIG;
    TriJob job{xyz, nullptr, nullptr, nullptr, verts, faces, dist, point, index, b, n, nf, nv};
    return launch_tri<true>(job, flags, stream);
}
