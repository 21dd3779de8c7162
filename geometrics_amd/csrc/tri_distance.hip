// Point-to-triangle arg-min scan for gfx950.
//
// Replaces TriDistanceKernel + launcher of the reference (tri_distance/tri_distance.cu:94-211,
// 213-228).  Four scans live here, all with 64 query points per workgroup,
// all producing the reference's sequential strict-'<' arg-min bit for bit (partial results are merged
// lexicographically on (distance, triangle) and the "first triangle seeds" rule is applied explicitly):
//
//   1. tri_distance_kernel            brute force: every (point, triangle) pair through the full decision
//                                     tree, corners staged in LDS.  Selected by GEOM_FLAG_TRI_BRUTE_FORCE;
//                                     it is the in-library cross-check of the other two.
//   2. tri_distance_culled_kernel     culled scan with in-kernel staging from (verts, faces) or the three
//                                     corner arrays -- what the workspace-free entry points (the exact
//                                     reference prototypes) run.
//   3. tri_prep_kernel + tri_scan_ws_kernel (+ tri_finalize_kernel)
//                                     culled scan over per-triangle records in a caller-provided workspace
//                                     (flat: every query tests every triangle sphere).
//   4. tri_prep_grouped_kernel + tri_scan_grouped_kernel
//                                     two-level scan over groups of 16 triangles (group sphere -> member
//                                     spheres -> literal evaluation) when the caller supplies a spatially
//                                     coherent visiting order: the fast path the python operators use.
//
// In the INDEXED variants the corners are gathered from verts through faces, so the three [b,F,3] corner
// arrays the reference materialises (utils.py:467-469) never exist.
// Arithmetic: tri_math.h (literal operation order of tri_distance.cu:140-191).
#include "geom_common.h"
#include "tri_math.h"
#include "nn_scan.h"
#include "surface_layout.h"
#include "draw_body.h"
#include "finalize_body.h"

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


// ---------------------------------------------------------------------------------------
// Culled scan (default).  Same result as the brute-force kernel above, bit for bit, at a
// fraction of the work:
//
//   1. every triangle gets a conservative bounding sphere that contains every point the
//      decision tree can return for it.  With the reference's region-6 quirk the candidate set
//      is the parallelogram A,B,C,D=B+C-A (region 6 returns C + t*(B-A), t in [0,1], the edge
//      C->D), so the sphere is centred on the midpoint of BC with radius max(|B-m|,|A-m|);
//      with GEOM_FLAG_FIX_REGION6 it is the centroid sphere of the triangle.  The radius is
//      inflated by 2^-10 relative and 2^-12 * coordinate-magnitude absolute margins (>= 100x
//      the fp32 error of any computed closest point).
//   2. per query point the workgroup keeps ONE 64-bit word in LDS: (distance bits << 32) |
//      (triangle << 3 | region), updated with ds_min_u64.  Distances are non-negative, so
//      unsigned order == lexicographic (distance, triangle) order == the reference's
//      sequential strict-'<' scan.  It is seeded by a HINT -- each wave evaluates, with the
//      literal arithmetic, the triangle whose sphere centre is nearest among every
//      HINT_STRIDE-th triangle -- so from the start it holds the distance of a REAL candidate,
//      an upper bound on the final minimum shared by all four waves.
//   3. the scan costs ~11 lane-ops per (point, triangle) pair: |p-c|^2 > (r_eff + s)^2 with
//      s = sqrt(bound) (+ margins) proves the triangle's literal distance is strictly above
//      the bound, so it can neither win nor tie.  Survivors (tens per point) are compacted
//      across the wave (ballot + mbcnt) into an LDS queue of (lane, triangle) items and
//      evaluated 64 at a time with every lane busy, corners read from the LDS chunk; each
//      result goes into the query's word with one LDS atomic, which also tightens the bound
//      every wave culls against.
//   4. the "first triangle seeds" rule (NaN seed sticks) is applied explicitly at the end.
//
// A triangle whose sphere is not trustworthy (non-finite, or nearly degenerate so that its
// computed plane normal could be garbage) gets r_eff = +inf and is never culled; a query
// whose bound is still unknown (NaN) culls nothing.
constexpr int CULL_CHUNK = 512;   // triangles staged per pass: 512 * (16 + 48) B = 32 KiB
constexpr int CULL_QCAP = 64 + 4 * GEOM_WAVE; // queue slots per wave
constexpr int HINT_STRIDE = 8;
constexpr unsigned long long KEY_NONE = ~0ull;

template <bool FIX6>
__device__ __forceinline__ float4 bounding_sphere(V3 A, V3 B, V3 C)
{
    V3 c;
    float r;
    if (FIX6) {
        const float third = 1.f / 3.f;
        c = geom::mk((A.x + B.x + C.x) * third, (A.y + B.y + C.y) * third, (A.z + B.z + C.z) * third);
        r = sqrtf(fmaxf(fmaxf(geom::dot3(A - c, A - c), geom::dot3(B - c, B - c)), geom::dot3(C - c, C - c)));
    } else {
        c = geom::mk((B.x + C.x) * 0.5f, (B.y + C.y) * 0.5f, (B.z + C.z) * 0.5f);
        r = sqrtf(fmaxf(fmaxf(geom::dot3(A - c, A - c), geom::dot3(B - c, B - c)), geom::dot3(C - c, C - c)));
    }
    const float mag = fabsf(c.x) + fabsf(c.y) + fabsf(c.z) + r;
    float r_eff = r * (1.f + 0x1p-10f) + 0x1p-12f * mag;
    // trust test: finite, and sin(angle at A) >= 1/64 so the normal (hence the plane foot) is accurate
    const V3 ab = B - A, ac = C - A;
    const V3 n = geom::cross3(ab, ac);
    const float nn = geom::dot3(n, n);
    const bool ok = (nn >= 0x1p-12f * geom::dot3(ab, ab) * geom::dot3(ac, ac)) && (nn > 0.f) && (mag < 0x1p60f);
    if (!ok) r_eff = INFINITY; // also catches NaN (comparisons false)
    return make_float4(c.x, c.y, c.z, r_eff);
}

__device__ __forceinline__ unsigned long long pack_key(float d, int k, int opt)
{
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)((k << 3) | opt);
}

template <bool INDEXED, bool TRUNC, bool FIX6>
__global__ __launch_bounds__(TRI_THREADS) void tri_distance_culled_kernel(TriJob job)
{
    __shared__ float4 sph[CULL_CHUNK];        // {centre, r_eff}
    __shared__ float4 cor[CULL_CHUNK][3];     // corners A, B, C of the staged chunk
    __shared__ unsigned long long qbest[TRI_QUERIES];
    __shared__ float qp[3][TRI_QUERIES];
    __shared__ unsigned queue[TRI_WAVES][CULL_QCAP];

    const int mesh = blockIdx.y;
    const int q0 = blockIdx.x * TRI_QUERIES;
    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int q = q0 + lane;
    const bool live = q < job.n;
    V3 p = geom::mk(0.f, 0.f, 0.f);
    if (live) p = load3(job.xyz + ((size_t)mesh * job.n + q) * 3);
    const float p_mag = fabsf(p.x) + fabsf(p.y) + fabsf(p.z);
    if (wave == 0) {
        qbest[lane] = KEY_NONE;
        qp[0][lane] = p.x;
        qp[1][lane] = p.y;
        qp[2][lane] = p.z;
    }

    // ---- hint: nearest of every HINT_STRIDE-th sphere centre, per wave --------------------
    {
        float near_c2 = INFINITY;
        int near_k = wave < job.m ? wave : 0; // any valid triangle is a valid hint
        const int hints = (job.m + HINT_STRIDE - 1) / HINT_STRIDE;
        for (int c0 = 0; c0 < hints; c0 += CULL_CHUNK) {
            const int len = min(CULL_CHUNK, hints - c0);
            for (int t = threadIdx.x; t < len; t += TRI_THREADS) {
                V3 A, B, C;
                fetch_triangle<INDEXED>(job, mesh, (c0 + t) * HINT_STRIDE, A, B, C);
                sph[t] = bounding_sphere<FIX6>(A, B, C);
            }
            __syncthreads();
            const int per_wave = (len + TRI_WAVES - 1) / TRI_WAVES;
            const int t_end = min(len, (wave + 1) * per_wave);
            for (int t = wave * per_wave; t < t_end; ++t) {
                const float4 rec = sph[t];
                const float c2 = geom::sqdist3(rec.x, rec.y, rec.z, p.x, p.y, p.z);
                if (c2 < near_c2) {
                    near_c2 = c2;
                    near_k = (c0 + t) * HINT_STRIDE;
                }
            }
            __syncthreads();
        }
        if (!(TRUNC && geom::ref_tail_skipped(near_k, job.m))) {
            V3 A, B, C;
            fetch_triangle<INDEXED>(job, mesh, near_k, A, B, C);
            int opt;
            const float d = geom::tri_pair_literal<FIX6>(p, A, B, C, opt);
            if (d == d) atomicMin(&qbest[lane], pack_key(d, near_k, opt));
        }
        __syncthreads();
    }

    // culling slack from the current bound (NaN while the query has no candidate: culls nothing)
    auto slack = [&]() {
        const float bound = __uint_as_float((unsigned)(qbest[lane] >> 32));
        return sqrtf(bound) * (1.f + 0x1p-10f) + 0x1p-12f * p_mag;
    };
    float s = slack();

    unsigned *my_queue = queue[wave];
    int qn = 0; // wave-uniform queue length

    // evaluate `count` (<= 64) queued items, one per lane, with the literal arithmetic
    auto process = [&](int first, int count, int c0) {
        if (lane < count) {
            const unsigned item = my_queue[first + lane];
            const int ql = item >> 26;
            const int t = item & 0x3ffffff;
            if (!(TRUNC && geom::ref_tail_skipped(c0 + t, job.m))) {
                const float4 a = cor[t][0], b = cor[t][1], c = cor[t][2];
                const V3 pq = geom::mk(qp[0][ql], qp[1][ql], qp[2][ql]);
                int opt;
                const float d = geom::tri_pair_literal<FIX6>(pq, geom::mk(a.x, a.y, a.z), geom::mk(b.x, b.y, b.z),
                                                             geom::mk(c.x, c.y, c.z), opt);
                if (d == d) atomicMin(&qbest[ql], pack_key(d, c0 + t, opt));
            }
        }
    };

    // ---- culled scan --------------------------------------------------------------------
    for (int c0 = 0; c0 < job.m; c0 += CULL_CHUNK) {
        const int len = min(CULL_CHUNK, job.m - c0);
        for (int t = threadIdx.x; t < len; t += TRI_THREADS) {
            V3 A, B, C;
            fetch_triangle<INDEXED>(job, mesh, c0 + t, A, B, C);
            float4 rec = bounding_sphere<FIX6>(A, B, C);
            if (TRUNC && geom::ref_tail_skipped(c0 + t, job.m)) rec.x = INFINITY; // culled unless s is inf/NaN
            sph[t] = rec;
            cor[t][0] = make_float4(A.x, A.y, A.z, 0.f);
            cor[t][1] = make_float4(B.x, B.y, B.z, 0.f);
            cor[t][2] = make_float4(C.x, C.y, C.z, 0.f);
        }
        __syncthreads();
        const int per_wave = (((len + TRI_WAVES - 1) / TRI_WAVES) + 3) & ~3; // multiple of 4
        const int t_begin = wave * per_wave;
        const int t_end = min(len, t_begin + per_wave);
        for (int t = t_begin; t < t_end; t += 4) {
            bool keep[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 rec = sph[min(t + j, len - 1)];
                const float c2 = geom::sqdist3(rec.x, rec.y, rec.z, p.x, p.y, p.z);
                const float reach = rec.w + s;
                keep[j] = live && (t + j < t_end) && !(c2 > reach * reach);
            }
            if (__ballot(keep[0] || keep[1] || keep[2] || keep[3]) != 0ull) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long m = __ballot(keep[j]);
                    if (m != 0ull) {
                        const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (keep[j]) my_queue[pos] = ((unsigned)lane << 26) | (unsigned)(t + j);
                        qn += __popcll(m);
                    }
                }
                if (qn >= GEOM_WAVE) {
                    do {
                        qn -= GEOM_WAVE;
                        process(qn, GEOM_WAVE, c0);
                    } while (qn >= GEOM_WAVE);
                    s = slack();
                }
            }
        }
        if (qn > 0) { // the corners of this chunk are about to be overwritten: drain
            process(0, qn, c0);
            qn = 0;
            s = slack();
        }
        __syncthreads();
    }

    if (wave == 0 && live) {
        const unsigned long long word = qbest[lane];
        float acc_d = __uint_as_float((unsigned)(word >> 32));
        int acc_k = (int)(unsigned)word;
        // "k == 0 ||" seed (tri_distance.cu:194): a NaN first distance sticks, and when nothing
        // finite was found the first triangle's result stands.
        V3 A, B, C;
        fetch_triangle<INDEXED>(job, mesh, 0, A, B, C);
        int opt0;
        const float d0 = geom::tri_pair_literal<FIX6>(p, A, B, C, opt0);
        if (d0 != d0 || word == KEY_NONE) {
            acc_d = d0;
            acc_k = opt0;
        }
        if (TRUNC) {
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

// ---------------------------------------------------------------------------------------
// Workspace variant of the culled scan (what the python operators use).
//
// A prep kernel writes, once per (mesh, triangle), the bounding sphere and the three corners
// into a caller-provided workspace (64 B per triangle).  The scan kernel then never touches
// faces/verts again and needs no LDS staging and no chunk barriers:
//   * the sphere records (16 B per triangle) are staged into LDS in chunks of 3072 with ONE
//     coalesced sweep per chunk -- every load of the chunk is in flight together, so a workgroup
//     pays one memory round trip per chunk -- and read back as wave-uniform ds_read_b128
//     (LDS broadcast).  (A scalar-load / SGPR-broadcast variant was measured first: with only
//     1-3 waves per SIMD at the 8-mesh shard its per-record s_load round trips, ~0.3 us each
//     and barely overlappable, made a workgroup take ~100 us regardless of batch size.)
//   * 16 waves share 64 query points and split every chunk 16 ways (6016 waves at the BASELINE
//     shard of 8 meshes), so one block's serial chain is 1/16 of the range (8 waves measured 16 % slower);
//   * survivors are compacted into per-wave LDS queues exactly as above, their corners come
//     from the workspace with one 48-byte gather per item, and the queue only drains at the end.
constexpr int WS_THREADS = 1024;
constexpr int WS_WAVES = WS_THREADS / GEOM_WAVE; // 16
constexpr int WS_PAD = 4 * WS_WAVES;             // triangle count is padded to a multiple of this

struct TriWs {
    float4 *sph; // [b][m_pad]     {centre, r_eff}
    float4 *cor; // [b][m_pad][3]  corners A, B, C
    unsigned long long *keys; // [b][n] merged (distance, triangle, region) words, used when split > 1
    int m_pad;
    int split;   // workgroups per query tile (each scans 1/split of the triangles)
};

__host__ __device__ inline int ws_pad(int m) { return (m + WS_PAD - 1) / WS_PAD * WS_PAD; }

template <bool INDEXED, bool TRUNC, bool FIX6>
__global__ __launch_bounds__(256) void tri_prep_kernel(TriJob job, TriWs ws)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int mesh = blockIdx.y;
    if (ws.split > 1 && k < job.n) ws.keys[(size_t)mesh * job.n + k] = KEY_NONE; // merged across workgroups later
    if (k >= ws.m_pad) return;
    float4 rec = make_float4(INFINITY, 0.f, 0.f, 0.f); // padding: culled (or dropped by the k < m test)
    V3 A = geom::mk(0.f, 0.f, 0.f), B = A, C = A;
    if (k < job.m) {
        fetch_triangle<INDEXED>(job, mesh, k, A, B, C);
        rec = bounding_sphere<FIX6>(A, B, C);
        if (TRUNC && geom::ref_tail_skipped(k, job.m)) rec.x = INFINITY;
    }
    const size_t o = (size_t)mesh * ws.m_pad + k;
    ws.sph[o] = rec;
    ws.cor[3 * o + 0] = make_float4(A.x, A.y, A.z, 0.f);
    ws.cor[3 * o + 1] = make_float4(B.x, B.y, B.z, 0.f);
    ws.cor[3 * o + 2] = make_float4(C.x, C.y, C.z, 0.f);
}

constexpr int WS_CHUNK = 3072; // sphere records staged in LDS per pass (48 KiB); multiple of WS_PAD

template <bool TRUNC, bool FIX6>
__global__ __launch_bounds__(WS_THREADS) void tri_scan_ws_kernel(const float *__restrict__ xyz, int b, int n, int m, int m_pad,
                                                                  const float4 *__restrict__ sph_all,
                                                                  const float4 *__restrict__ cor_all,
                                                                  float *__restrict__ dist, int *__restrict__ point,
                                                                  int *__restrict__ index, int split,
                                                                  unsigned long long *__restrict__ keys)
{
    __shared__ float4 tile[WS_CHUNK];
    __shared__ unsigned long long qbest[TRI_QUERIES];
    __shared__ float qp[3][TRI_QUERIES];
    __shared__ unsigned queue[WS_WAVES][CULL_QCAP];

    int mesh, task; // all workgroups of a mesh share one XCD (its records stay in that L2)
    if (!geom::xcd_assign(blockIdx.x, b, ((n + TRI_QUERIES - 1) / TRI_QUERIES) * split, mesh, task)) return;
    const int qtile = task / split, part = task - qtile * split; // `split` workgroups share a query tile
    const int q0 = qtile * TRI_QUERIES;
    // this workgroup's triangle range [r_begin, r_end): 1/split of the padded range, in WS_PAD units
    const int r_len = ((m_pad / WS_PAD + split - 1) / split) * WS_PAD;
    const int r_begin = part * r_len, r_end = min(m_pad, r_begin + r_len);
    if (r_begin >= r_end) return;
    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform => scalar addressing
    const int q = q0 + lane;
    const bool live = q < n;
    const float4 *__restrict__ sph = sph_all + (size_t)mesh * m_pad;
    const float4 *__restrict__ cor = cor_all + (size_t)mesh * m_pad * 3;
    V3 p = geom::mk(0.f, 0.f, 0.f);
    if (live) p = load3(xyz + ((size_t)mesh * n + q) * 3);
    const float p_mag = fabsf(p.x) + fabsf(p.y) + fabsf(p.z);
    if (wave == 0) {
        qbest[lane] = KEY_NONE;
        qp[0][lane] = p.x;
        qp[1][lane] = p.y;
        qp[2][lane] = p.z;
    }

    // literal evaluation of triangle k for the query held by lane `ql`, merged with one LDS atomic
    auto evaluate = [&](int ql, int k) {
        if (k >= m || (TRUNC && geom::ref_tail_skipped(k, m))) return;
        const float4 a = cor[3 * (size_t)k + 0], bq = cor[3 * (size_t)k + 1], c = cor[3 * (size_t)k + 2];
        const V3 pq = geom::mk(qp[0][ql], qp[1][ql], qp[2][ql]);
        int opt;
        const float d = geom::tri_pair_literal<FIX6>(pq, geom::mk(a.x, a.y, a.z), geom::mk(bq.x, bq.y, bq.z),
                                                     geom::mk(c.x, c.y, c.z), opt);
        if (d == d) atomicMin(&qbest[ql], pack_key(d, k, opt));
    };

    // ---- hint: every HINT_STRIDE-th sphere record of the whole mesh is staged once (all loads in
    //      flight together), each wave takes the nearest centre of its share and evaluates it ----
    {
        const int hints = (m + HINT_STRIDE - 1) / HINT_STRIDE;
        float near_c2 = INFINITY;
        int near_k = wave < m ? wave : 0;
        for (int h0 = 0; h0 < hints; h0 += WS_CHUNK) {
            const int len = min(WS_CHUNK, hints - h0);
            for (int t = threadIdx.x; t < len; t += WS_THREADS) tile[t] = sph[(size_t)(h0 + t) * HINT_STRIDE];
            __syncthreads();
            for (int t = wave; t < len; t += WS_WAVES) {
                const float4 rec = tile[t];
                const float c2 = geom::sqdist3(rec.x, rec.y, rec.z, p.x, p.y, p.z);
                if (c2 < near_c2) {
                    near_c2 = c2;
                    near_k = (h0 + t) * HINT_STRIDE;
                }
            }
            __syncthreads();
        }
        evaluate(lane, near_k);
        __syncthreads();
    }

    auto slack = [&]() {
        const float bound = __uint_as_float((unsigned)(qbest[lane] >> 32)); // NaN until a candidate exists
        return sqrtf(bound) * (1.f + 0x1p-10f) + 0x1p-12f * p_mag;
    };
    float s = slack();

    unsigned *my_queue = queue[wave];
    int qn = 0; // wave-uniform queue length
    auto pop_batch = [&](int first, int count) {
        if (lane < count) {
            const unsigned item = my_queue[first + lane];
            evaluate((int)(item >> 26), (int)(item & 0x3ffffffu));
        }
    };

    // ---- culled scan: chunks of sphere records are staged in LDS with one coalesced sweep (every
    //      load of the chunk in flight at once), then each wave tests its contiguous share of the chunk
    //      with wave-uniform ds_read_b128 (LDS broadcast) --------------------------------------------
    const unsigned long long live_mask = __ballot(live);
    for (int c0 = r_begin; c0 < r_end; c0 += WS_CHUNK) {
        const int len = min(WS_CHUNK, r_end - c0); // multiple of WS_PAD = 4 * WS_WAVES
        for (int t = threadIdx.x; t < len; t += WS_THREADS) tile[t] = sph[c0 + t];
        __syncthreads();
        const int per_wave = len / WS_WAVES; // multiple of 4
        const int t_begin = wave * per_wave;
        for (int t = t_begin; t < t_begin + per_wave; t += 4) {
            unsigned long long keep[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 rec = tile[t + j];
                const float c2 = geom::sqdist3(rec.x, rec.y, rec.z, p.x, p.y, p.z);
                const float reach = rec.w + s;
                keep[j] = __builtin_amdgcn_ballot_w64(!(c2 > reach * reach)) & live_mask;
            }
            if ((keep[0] | keep[1] | keep[2] | keep[3]) != 0ull) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long mask = keep[j];
                    if (mask != 0ull) {
                        const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if ((mask >> lane) & 1ull) my_queue[pos] = ((unsigned)lane << 26) | (unsigned)(c0 + t + j);
                        qn += __popcll(mask);
                    }
                }
                if (qn >= GEOM_WAVE) {
                    do {
                        qn -= GEOM_WAVE;
                        pop_batch(qn, GEOM_WAVE);
                    } while (qn >= GEOM_WAVE);
                    s = slack();
                }
            }
        }
        __syncthreads(); // the chunk is overwritten next; queued items carry global indices and survive
    }
    if (qn > 0) pop_batch(0, qn);
    __syncthreads();

    if (split > 1) { // partial result: merged across the tile's workgroups, finished by tri_finalize_kernel
        if (wave == 0 && live && qbest[lane] != KEY_NONE) atomicMin(&keys[(size_t)mesh * n + q], qbest[lane]);
        return;
    }
    if (wave == 0 && live) {
        const unsigned long long word = qbest[lane];
        float acc_d = __uint_as_float((unsigned)(word >> 32));
        int acc_k = (int)(unsigned)word;
        // "k == 0 ||" seed (tri_distance.cu:194)
        const float4 a = cor[0], bq = cor[1], c = cor[2];
        int opt0;
        const float d0 = geom::tri_pair_literal<FIX6>(p, geom::mk(a.x, a.y, a.z), geom::mk(bq.x, bq.y, bq.z),
                                                      geom::mk(c.x, c.y, c.z), opt0);
        if (d0 != d0 || word == KEY_NONE) {
            acc_d = d0;
            acc_k = opt0;
        }
        if (TRUNC) {
            const int last0 = ((m - 1) / geom::REF_TILE) * geom::REF_TILE;
            if (m - last0 < 4 && (last0 == 0 || acc_d > 10000.f)) {
                acc_d = 10000.f;
                acc_k = 0;
            }
        }
        const size_t o = (size_t)mesh * n + q;
        dist[o] = acc_d;
        point[o] = acc_k & 7;
        index[o] = acc_k >> 3;
    }
}

// split > 1 only: the merged word of every query -> outputs, with the "first triangle seeds" rule
template <bool TRUNC, bool FIX6>
__global__ __launch_bounds__(256) void tri_finalize_kernel(const float *__restrict__ xyz, int n, int m,
                                                           const float4 *__restrict__ first_all, size_t first_stride,
                                                           const unsigned long long *__restrict__ keys,
                                                           float *__restrict__ dist, int *__restrict__ point,
                                                           int *__restrict__ index)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int mesh = blockIdx.y;
    if (q >= n) return;
    const size_t o = (size_t)mesh * n + q;
    const unsigned long long word = keys[o];
    float acc_d = __uint_as_float((unsigned)(word >> 32));
    int acc_k = (int)(unsigned)word;
    const float4 *cor = first_all + (size_t)mesh * first_stride; // corners of (original) triangle 0
    const V3 p = load3(xyz + o * 3);
    const float4 a = cor[0], bq = cor[1], c = cor[2];
    int opt0;
    const float d0 = geom::tri_pair_literal<FIX6>(p, geom::mk(a.x, a.y, a.z), geom::mk(bq.x, bq.y, bq.z),
                                                  geom::mk(c.x, c.y, c.z), opt0);
    if (d0 != d0 || word == KEY_NONE) { // "k == 0 ||" seed (tri_distance.cu:194)
        acc_d = d0;
        acc_k = opt0;
    }
    if (TRUNC) {
        const int last0 = ((m - 1) / geom::REF_TILE) * geom::REF_TILE;
        if (m - last0 < 4 && (last0 == 0 || acc_d > 10000.f)) {
            acc_d = 10000.f;
            acc_k = 0;
        }
    }
    dist[o] = acc_d;
    point[o] = acc_k & 7;
    index[o] = acc_k >> 3;
}

template <bool INDEXED, bool TRUNC, bool FIX6>
int launch_ws_variant(const TriJob &job, const TriWs &ws, hipStream_t s)
{
    const int prep_items = ws.split > 1 && job.n > ws.m_pad ? job.n : ws.m_pad;
    hipLaunchKernelGGL((tri_prep_kernel<INDEXED, TRUNC, FIX6>), dim3((prep_items + 255) / 256, job.b), dim3(256), 0, s, job, ws);
    const int qtiles = (job.n + TRI_QUERIES - 1) / TRI_QUERIES;
    hipLaunchKernelGGL((tri_scan_ws_kernel<TRUNC, FIX6>), dim3(geom::xcd_grid(job.b, qtiles * ws.split)), dim3(WS_THREADS),
                       0, s, job.xyz, job.b, job.n, job.m, ws.m_pad, ws.sph, ws.cor, job.dist, job.point, job.index,
                       ws.split, ws.keys);
    if (ws.split > 1)
        hipLaunchKernelGGL((tri_finalize_kernel<TRUNC, FIX6>), dim3((job.n + 255) / 256, job.b), dim3(256), 0, s, job.xyz,
                           job.n, job.m, ws.cor, (size_t)ws.m_pad * 3, ws.keys, job.dist, job.point, job.index);
    return geom::launch_status();
}

// ---------------------------------------------------------------------------------------
// Grouped (two-level) scan: the fast path when the caller supplies a spatially coherent triangle
// order (`order` != null; geometrics_amd.tri_distance derives a Morton order of the face centroids once
// per face list).  Same result as every kernel above, bit for bit -- the permutation only decides which
// triangles share a group; keys carry the ORIGINAL triangle index, so ties still go to the lowest one.
//
//   prep   slot j holds triangle order[j]: sphere + corners as before (the original index rides in the
//          spare lane of the first corner), and every GRP = 16 consecutive slots get a GROUP sphere
//          {c_g, R_g}: c_g = mean of the member centres, R_g = max(|c_i - c_g| + r_eff_i), inflated by the
//          same margins.  Every point the decision tree can return for a member lies inside it, so
//              lower bound   d_ref(p, t) >= (|p - c_g| - R_g)^2     for every member t
//              upper bound   min_t d_ref(p, t) <= (|p - c_g| + R_g)^2
//          (R_g = inf as soon as one member is untrusted: such a group is never culled, never a seed.)
//   scan   a workgroup = 64 query points x 16 waves.
//          A1  lanes <-> queries, group spheres broadcast from LDS (320 records for 5120 triangles
//              instead of 5120): the smallest upper bound and its group -> per-query seed;
//          A1' the 16 members of that group are evaluated literally, one thread per (query, member): the
//              bound every later test uses starts from a real candidate next to the query;
//          A2  groups again: |p - c_g|^2 > (R_g + s)^2 drops the group, survivors (about ten per query)
//              are compacted into a per-wave queue of (query, group) items;
//          B   lanes <-> (item, member), 16 items per step (4 sphere loads per lane in flight together): a
//              group's member records are 256 contiguous bytes (L2); survivors go to a second queue of
//              (query, slot) items;
//          C   literal evaluation 64 items at a time, one ds_min_u64 per result (tightens every wave's s).
//          Waves take groups in a strided pattern so that spatially sorted queries still spread evenly.
// Per query this is ~2 x 320 group tests + ~100 member tests + a few tens of literal evaluations instead
// of 5120 member tests; with an incoherent order the group spheres are useless (every group survives) and
// the flat kernel above is the better choice -- which is why the hierarchy is opt-in through `order`.
constexpr int GRP = 16;
// Waves per workgroup is a template parameter: launching a wave costs ~1.5 ns chip-wide, so at the 8-mesh shard 376
// workgroups x 16 waves spend ~9 us just starting; 8 waves per workgroup measured 35.7 us against 44.7 (4 waves: 45.0).
// With few query tiles (one mesh) the longer per-wave chains of 8 waves cost more than the launch saves (24.5 vs 22.5 us).
constexpr int HS_MAX_WAVES = 16;
constexpr int HS_GCHUNK = 512;                   // group spheres staged per pass (8 KiB = 8192 triangles); LDS is shared with co-running kernels
constexpr int HS_QA = 16 + 4 * GEOM_WAVE;        // (query, group) items per wave
constexpr int HS_QB = 2 * GEOM_WAVE;             // (query, slot) items per wave
constexpr unsigned INF_BITS = 0x7f800000u;

// optional epilogue of the grouped scan: the point-to-surface quantities of the winning triangle (what
// geom_p2tri_loss_fwd_f32 computes in a launch of its own), INDEXED jobs only
struct SurfaceOut {
    const float *verts;
    const int64_t *faces;
    int nv;
    float *sqdist, *closest, *weights; // [b,n], [b,n,3], [b,n,3]; sqdist == null: no epilogue
    // gradient record of the gt point for the surface-loss backward (csrc/surface_gather.hip), rec == null: none
    float4 *rec;      // [b][per][2], the gt points follow the `rec_first` sampled points of their mesh
    float coef;
    int per, rec_first;
};
constexpr SurfaceOut NO_SURFACE{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0};

constexpr int TAIL_CTR_STRIDE = 32;
__host__ __device__ inline int tail_counters(int b) { return 2 * b + 1; }
// whether the merged-keys region of a workspace (b * n * 8 bytes) holds the counters
__host__ __device__ inline bool tail_counters_fit(int b, int n) { return (int64_t)tail_counters(b) * TAIL_CTR_STRIDE * 4 <= (int64_t)b * n * 8; }

struct TriGws {
    float4 *sph;   // [b][m_pad]     member spheres, slot order
    float4 *cor;   // [b][m_pad][3]  corners; cor[3j].w = original triangle index (int bits), -1 = padding
    float4 *grp;   // [b][m_pad/16]  group spheres
    float4 *first; // [b][3]         corners of original triangle 0 (the "k == 0 ||" seed)
    unsigned long long *keys; // [b][n], split > 1 only
    int m_pad, split;
};

// j = visiting slot, mesh = which mesh (whole waves share a mesh and 64 consecutive slots)
template <bool INDEXED, bool TRUNC, bool FIX6>
__device__ __forceinline__ void tri_prep_grouped_body(const TriJob &job, const TriGws &ws, const int *__restrict__ order, int j,
                                                      int mesh)
{
    // the tile-completion counters of the fused scan's finalize tail (ScanTail, tail_counters(b) lines of 128 bytes) live in
    // the merged-keys region, which a scan without triangle split never touches: zero before every scan that follows a prep
    if (ws.split == 1 && j < 3 && tail_counters_fit(job.b, job.n)) {
        int *done = reinterpret_cast<int *>(ws.keys);
        const int which = j == 0 ? mesh : j == 1 ? job.b + mesh : 2 * job.b;
        if (j < 2 || (j == 2 && mesh == 0)) done[(size_t)which * TAIL_CTR_STRIDE] = 0;
    }
    if (ws.split > 1 && j < job.n) ws.keys[(size_t)mesh * job.n + j] = KEY_NONE;
    if (j >= ws.m_pad) return; // m_pad is a multiple of 64: whole waves leave together
    int k = -1;
    if (j < job.m) {
        k = order[j];
        if (k < 0 || k >= job.m) k = -1; // not a permutation entry: the slot is dropped
    }
    float4 rec = make_float4(INFINITY, 0.f, 0.f, 0.f);
    V3 A = geom::mk(0.f, 0.f, 0.f), B = A, C = A;
    bool scanned = false; // takes part in the arg-min
    if (k >= 0) {
        fetch_triangle<INDEXED>(job, mesh, k, A, B, C);
        rec = bounding_sphere<FIX6>(A, B, C);
        scanned = !(TRUNC && geom::ref_tail_skipped(k, job.m));
        if (!scanned) rec.x = INFINITY;
        if (k == 0) {
            float4 *f = ws.first + (size_t)mesh * 3;
            f[0] = make_float4(A.x, A.y, A.z, 0.f);
            f[1] = make_float4(B.x, B.y, B.z, 0.f);
            f[2] = make_float4(C.x, C.y, C.z, 0.f);
        }
    }
    const size_t o = (size_t)mesh * ws.m_pad + j;
    ws.sph[o] = rec;
    ws.cor[3 * o + 0] = make_float4(A.x, A.y, A.z, __int_as_float(k));
    ws.cor[3 * o + 1] = make_float4(B.x, B.y, B.z, 0.f);
    ws.cor[3 * o + 2] = make_float4(C.x, C.y, C.z, 0.f);

    // group sphere over the 16 lanes of this group (xor shuffles stay inside an aligned 16-lane block)
    const bool centred = scanned && (fabsf(rec.x) + fabsf(rec.y) + fabsf(rec.z) < INFINITY); // finite centre
    float sx = centred ? rec.x : 0.f, sy = centred ? rec.y : 0.f, sz = centred ? rec.z : 0.f, cnt = centred ? 1.f : 0.f;
    float bad = (scanned && !centred) ? 1.f : 0.f; // a scanned member without a usable sphere
#pragma unroll
    for (int d = GRP / 2; d > 0; d >>= 1) {
        sx += __shfl_xor(sx, d);
        sy += __shfl_xor(sy, d);
        sz += __shfl_xor(sz, d);
        cnt += __shfl_xor(cnt, d);
        bad += __shfl_xor(bad, d);
    }
    float4 g = make_float4(INFINITY, 0.f, 0.f, 0.f); // no scanned member: culled for every finite query
    float reach = 0.f;
    if (cnt > 0.f) {
        const float inv = 1.f / cnt;
        g = make_float4(sx * inv, sy * inv, sz * inv, 0.f);
        if (centred) reach = sqrtf(geom::sqdist3(rec.x, rec.y, rec.z, g.x, g.y, g.z)) * (1.f + 0x1p-20f) + rec.w;
    }
#pragma unroll
    for (int d = GRP / 2; d > 0; d >>= 1) reach = fmaxf(reach, __shfl_xor(reach, d));
    if (cnt > 0.f) {
        const float mag = fabsf(g.x) + fabsf(g.y) + fabsf(g.z) + reach;
        g.w = reach * (1.f + 0x1p-10f) + 0x1p-12f * mag;
    }
    if (bad > 0.f) g = make_float4(cnt > 0.f ? g.x : 0.f, cnt > 0.f ? g.y : 0.f, cnt > 0.f ? g.z : 0.f, INFINITY);
    if ((j & (GRP - 1)) == 0) ws.grp[(size_t)mesh * (ws.m_pad / GRP) + j / GRP] = g;
}

template <bool INDEXED, bool TRUNC, bool FIX6>
__global__ __launch_bounds__(256) void tri_prep_grouped_kernel(TriJob job, TriGws ws, const int *__restrict__ order)
{
    tri_prep_grouped_body<INDEXED, TRUNC, FIX6>(job, ws, order, blockIdx.x * 256 + threadIdx.x, blockIdx.y);
}

// The per-step PREPARATION of the surface loss in one launch: everything that depends only on the vertex positions --
// the random face draws (+ the sampled points) of batch_sample and the triangle records / group spheres of the
// two-level scan.  Workgroups [0, draw_blocks) run a draw chunk, the rest 1024 triangle slots each.
template <bool FIX6, bool SORTED>
__global__ __launch_bounds__(DRAW_THREADS) void surface_prepare_kernel(int draw_blocks, int draw_chunks, int nv, const float *verts,
                                                                        int nf, const int64_t *faces, int num,
                                                                        unsigned long long *rng_state, int64_t *choices, float *u,
                                                                        float *v, float *points, TriJob job, TriGws ws,
                                                                        const int *__restrict__ order, int prep_chunks, DrawSort srt)
{
    if ((int)blockIdx.x < draw_blocks) { // SORTED: one workgroup per mesh (draw_chunks == 1) draws everything and sorts
        draw_samples_body<SORTED>(blockIdx.x % draw_chunks, blockIdx.x / draw_chunks, (unsigned long long)draw_blocks, nv, verts, nf,
                                  faces, num, nullptr, 0, rng_state, choices, u, v, points, srt);
    } else {
        const int pid = blockIdx.x - draw_blocks;
        tri_prep_grouped_body<true, false, FIX6>(job, ws, order, (pid % prep_chunks) * DRAW_THREADS + threadIdx.x, pid / prep_chunks);
    }
}

#ifdef SCAN_TILE_STAMPS // tools/probe only: phase boundaries of every triangle tile (wave 0) + when each wave reaches the closing barrier
__device__ long long scan_tri_phases[16 * 1024];
extern "C" int geom_probe_read_tri_phases(long long *host, int rows)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(scan_tri_phases), sizeof(long long) * 16 * (size_t)rows);
}
#define TRI_PHASE(k) do { if (threadIdx.x == 0 && bid < 1024) scan_tri_phases[16 * bid + (k)] = wall_clock64(); } while (0)
#define TRI_WAVE_DONE() do { if (lane == 0 && bid < 1024) scan_tri_phases[16 * bid + 8 + wave] = wall_clock64(); } while (0)
#else
#define TRI_PHASE(k) do { } while (0)
#define TRI_WAVE_DONE() do { } while (0)
#endif

// the tile's LDS as one object (see NNCullLds in nn_scan.h: the fused scan overlays the two bodies' LDS)
template <int HS_WAVES>
struct TriTileLds {
    float4 gtile[HS_GCHUNK];
    unsigned long long qbest[TRI_QUERIES]; // best evaluated (distance, triangle, region)
    unsigned long long qseed[TRI_QUERIES]; // smallest group upper bound and its group
    unsigned long long seed_part[HS_WAVES][TRI_QUERIES]; // per-wave candidates for it (no LDS atomics)
    unsigned qs[TRI_QUERIES];              // seed slack (float bits, positive: uint order == float order)
    float qp[3][TRI_QUERIES], qmag[TRI_QUERIES];
    unsigned queue_a[HS_WAVES][HS_QA];
    unsigned queue_b[HS_WAVES][HS_QB];
};

template <bool TRUNC, bool FIX6, int HS_WAVES>
__device__ __forceinline__ void tri_scan_grouped_body(int bid, const float *__restrict__ xyz, int b, int n, int m,
                                                      const TriGws &ws, float *__restrict__ dist, int *__restrict__ point,
                                                      int *__restrict__ index, const SurfaceOut &surf, TriTileLds<HS_WAVES> &L)
{
    static_assert(HS_WAVES <= GRP, "the seed reduction maps wave w to member lane w");
    constexpr int HS_THREADS = HS_WAVES * GEOM_WAVE;
    auto &gtile = L.gtile;
    auto &qbest = L.qbest;
    auto &qseed = L.qseed;
    auto &seed_part = L.seed_part;
    auto &qs = L.qs;
    auto &qp = L.qp;
    auto &qmag = L.qmag;
    auto &queue_a = L.queue_a;
    auto &queue_b = L.queue_b;

    const int split = ws.split, m_pad = ws.m_pad;
    int mesh, task;
    if (!geom::xcd_assign(bid, b, ((n + TRI_QUERIES - 1) / TRI_QUERIES) * split, mesh, task)) return;
    const int qtile = task / split, part = task - qtile * split;
    const int q0 = qtile * TRI_QUERIES;
    // this workgroup's group range, in units of 4 groups (m_pad is a multiple of 64)
    const int groups = m_pad / GRP;
    const int r_len = ((groups / 4 + split - 1) / split) * 4;
    const int r_begin = part * r_len, r_end = min(groups, r_begin + r_len);
    if (r_begin >= r_end) return;
    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = q0 + lane;
    const bool live = q < n;
    const float4 *__restrict__ sph = ws.sph + (size_t)mesh * m_pad;
    const float4 *__restrict__ cor = ws.cor + (size_t)mesh * m_pad * 3;
    const float4 *__restrict__ grp = ws.grp + (size_t)mesh * groups;
    V3 p = geom::mk(0.f, 0.f, 0.f);
    if (live) p = load3(xyz + ((size_t)mesh * n + q) * 3);
    if (wave == 0) {
        qbest[lane] = KEY_NONE;
        qseed[lane] = KEY_NONE;
        qs[lane] = INF_BITS;
        qp[0][lane] = p.x;
        qp[1][lane] = p.y;
        qp[2][lane] = p.z;
        qmag[lane] = fabsf(p.x) + fabsf(p.y) + fabsf(p.z);
    }

    // cull slack of query ql: sqrt(best evaluated distance) with margins, or the sphere-derived seed
    auto slack = [&](int ql) {
        const float bound = __uint_as_float((unsigned)(qbest[ql] >> 32)); // NaN until a candidate exists
        const float sb = __builtin_amdgcn_sqrtf(bound) * (1.f + 0x1p-10f) + 0x1p-12f * qmag[ql]; // 1-ulp sqrt, 2^-10 margin
        return fminf(sb, __uint_as_float(qs[ql])); // fminf drops the NaN; +inf (nothing known) culls nothing
    };
    auto evaluate = [&](int ql, int j) {
        if (j >= m_pad) return;
        const float4 a = cor[3 * (size_t)j + 0], bq = cor[3 * (size_t)j + 1], c = cor[3 * (size_t)j + 2];
        const int k = __float_as_int(a.w);
        if (k < 0 || (TRUNC && geom::ref_tail_skipped(k, m))) return;
        const V3 pq = geom::mk(qp[0][ql], qp[1][ql], qp[2][ql]);
        int opt;
        const float d = geom::tri_pair_literal<FIX6>(pq, geom::mk(a.x, a.y, a.z), geom::mk(bq.x, bq.y, bq.z),
                                                     geom::mk(c.x, c.y, c.z), opt);
        if (d == d) {
            const unsigned long long key = pack_key(d, k, opt);
            if (key < qbest[ql]) atomicMin(&qbest[ql], key); // most candidates lose against the seed: plain read first
        }
    };

    unsigned *qa = queue_a[wave], *qb = queue_b[wave];
    int na = 0, nb = 0; // wave-uniform queue lengths
    const unsigned long long live_mask = __ballot(live);
    auto drain_b = [&](int first, int count) {
        if (lane < count) {
            const unsigned item = qb[first + lane];
            evaluate((int)(item >> 26), (int)(item & 0x3ffffffu));
        }
    };
    // phase B for `count` (<= 16) items of queue A starting at `first`: a lane takes member (lane % 16) of items
    // lane/16, lane/16 + 4, ...; the four sphere loads of a lane are issued together (one L2 round trip per step)
    auto members = [&](int first, int count) {
        const int it = lane >> 4, mem = lane & (GRP - 1);
        float4 rec[4];
        int ql[4], slot[4];
        bool valid[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = r * 4 + it;
            valid[r] = idx < count;
            const unsigned item = valid[r] ? qa[first + idx] : 0u;
            ql[r] = (int)(item >> 26);
            slot[r] = (int)(item & 0x3ffffffu) * GRP + mem;
            rec[r] = make_float4(INFINITY, 0.f, 0.f, 0.f);
            if (valid[r]) rec[r] = sph[slot[r]];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c2 = geom::sqdist3(rec[r].x, rec[r].y, rec[r].z, qp[0][ql[r]], qp[1][ql[r]], qp[2][ql[r]]);
            const float reach = rec[r].w + slack(ql[r]);
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(valid[r] && !(c2 > reach * reach));
            if (mask != 0ull) {
                const int pos = nb + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if ((mask >> lane) & 1ull) qb[pos] = ((unsigned)ql[r] << 26) | (unsigned)slot[r];
                nb += __popcll(mask);
                if (nb >= GEOM_WAVE) {
                    nb -= GEOM_WAVE;
                    drain_b(nb, GEOM_WAVE);
                }
            }
        }
    };

    TRI_PHASE(0); // entry (queries requested)
    for (int c0 = r_begin; c0 < r_end; c0 += HS_GCHUNK) {
        const int len = min(HS_GCHUNK, r_end - c0); // multiple of 4
        for (int t = threadIdx.x; t < len; t += HS_THREADS) gtile[t] = grp[c0 + t];
        __syncthreads();
        TRI_PHASE(1); // group spheres staged
        const int batches = len / 4;

        // ---- A1: smallest upper bound (|p - c_g| + R_g) over this wave's groups ----
        {
            float u_best = INFINITY;
            int g_best = 0;
            for (int bi = wave; bi < batches; bi += HS_WAVES) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 rec = gtile[bi * 4 + jj];
                    // raw v_sqrt_f32 (1 ulp): only ranks the groups, and the fallback slack below is inflated by 2^-10
                    const float u = __builtin_amdgcn_sqrtf(geom::sqdist3(rec.x, rec.y, rec.z, p.x, p.y, p.z)) + rec.w;
                    if (u < u_best) { // NaN / inf never win
                        u_best = u;
                        g_best = c0 + bi * 4 + jj;
                    }
                }
            }
            seed_part[wave][lane] = (live && u_best < INFINITY)
                                        ? (((unsigned long long)__float_as_uint(u_best) << 32) | (unsigned)g_best)
                                        : KEY_NONE;
        }
        __syncthreads();
        TRI_PHASE(2); // A1 done
        // ---- A1': the 16 members of the seed group are evaluated literally, one thread per (query, member):
        //      the cull bound starts from a REAL candidate next to the query, not from a sphere estimate ----
        for (int t = threadIdx.x; t < TRI_QUERIES * GRP; t += HS_THREADS) { // wave-aligned: whole 16-lane groups
            const int ql = t >> 4, mem = t & (GRP - 1);
            unsigned long long word = mem < HS_WAVES ? seed_part[mem][ql] : KEY_NONE; // lane `mem` holds wave `mem`'s candidate
#pragma unroll
            for (int d = GRP / 2; d > 0; d >>= 1) {
                const unsigned long long other = __shfl_xor(word, d);
                word = other < word ? other : word;
            }
            if (mem == 0) qseed[ql] = word;
            if (word != KEY_NONE) {
                evaluate(ql, (int)(unsigned)word * GRP + mem);
                if (mem == 0) { // sphere-derived fallback for a seed group whose evaluations are all NaN
                    const float s0 = __uint_as_float((unsigned)(word >> 32)) * (1.f + 0x1p-10f) + 0x1p-12f * qmag[ql];
                    if (s0 < INFINITY) atomicMin(&qs[ql], __float_as_uint(s0));
                }
            }
        }
        __syncthreads();
        TRI_PHASE(3); // A1' done

        // ---- A2 + B + C ----
        const int seed_g = (int)(unsigned)qseed[lane]; // already evaluated (KEY_NONE -> -1: matches no group)
        float s = slack(lane);
        for (int bi = wave; bi < batches; bi += HS_WAVES) {
            unsigned long long keep[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 rec = gtile[bi * 4 + jj];
                const float c2 = geom::sqdist3(rec.x, rec.y, rec.z, p.x, p.y, p.z);
                const float reach = rec.w + s;
                keep[jj] = __builtin_amdgcn_ballot_w64(!(c2 > reach * reach) && (c0 + bi * 4 + jj != seed_g)) & live_mask;
            }
            if ((keep[0] | keep[1] | keep[2] | keep[3]) != 0ull) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned long long mask = keep[jj];
                    if (mask != 0ull) {
                        const int pos = na + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if ((mask >> lane) & 1ull) qa[pos] = ((unsigned)lane << 26) | (unsigned)(c0 + bi * 4 + jj);
                        na += __popcll(mask);
                    }
                }
                if (na >= 16) {
                    do {
                        na -= 16;
                        members(na, 16);
                    } while (na >= 16);
                    s = slack(lane);
                }
            }
        }
        if (na > 0) { // the chunk's remaining items (their group ids are global, but finishing here keeps s fresh)
            members(0, na);
            na = 0;
        }
        TRI_WAVE_DONE(); // this wave's A2 + B + C of the chunk
        __syncthreads(); // gtile / qseed are rewritten by the next chunk
        TRI_PHASE(4); // all waves through A2 + B + C
    }
    if (nb > 0) drain_b(0, nb);
    __syncthreads();
    TRI_PHASE(5); // final drain

    if (split > 1) {
        if (wave == 0 && live && qbest[lane] != KEY_NONE) atomicMin(&ws.keys[(size_t)mesh * n + q], qbest[lane]);
        return;
    }
    if (wave == 0 && live) {
        const unsigned long long word = qbest[lane];
        float acc_d = __uint_as_float((unsigned)(word >> 32));
        int acc_k = (int)(unsigned)word;
        const float4 *f = ws.first + (size_t)mesh * 3; // "k == 0 ||" seed (tri_distance.cu:194)
        const float4 a = f[0], bq = f[1], c = f[2];
        int opt0;
        const float d0 = geom::tri_pair_literal<FIX6>(p, geom::mk(a.x, a.y, a.z), geom::mk(bq.x, bq.y, bq.z),
                                                      geom::mk(c.x, c.y, c.z), opt0);
        if (d0 != d0 || word == KEY_NONE) {
            acc_d = d0;
            acc_k = opt0;
        }
        if (TRUNC) {
            const int last0 = ((m - 1) / geom::REF_TILE) * geom::REF_TILE;
            if (m - last0 < 4 && (last0 == 0 || acc_d > 10000.f)) {
                acc_d = 10000.f;
                acc_k = 0;
            }
        }
        const size_t o = (size_t)mesh * n + q;
        dist[o] = acc_d;
        point[o] = acc_k & 7;
        geom::store_agent(index + o, (int)(acc_k >> 3)); // index and sqdist: read by the finalize tail of the same launch
        if (surf.sqdist) { // point-to-surface epilogue (calc_point_to_line on the winner, utils.py:506-550)
            const float *V = surf.verts + (size_t)mesh * surf.nv * 3;
            const int64_t f = acc_k >> 3;
            const V3 A = load3(V + 3 * surf.faces[3 * f + 0]);
            const V3 B = load3(V + 3 * surf.faces[3 * f + 1]);
            const V3 C = load3(V + 3 * surf.faces[3 * f + 2]);
            V3 w;
            const V3 hit = geom::closest_on_triangle(p, A, B, C, acc_k & 7, w);
            const V3 d = hit - p;
            geom::store_agent(surf.sqdist + o, geom::dot3(d, d));
            surf.closest[3 * o + 0] = hit.x, surf.closest[3 * o + 1] = hit.y, surf.closest[3 * o + 2] = hit.z;
            surf.weights[3 * o + 0] = w.x, surf.weights[3 * o + 1] = w.y, surf.weights[3 * o + 2] = w.z;
            if (surf.rec) { // backward record: (closest - p) * coef, corner weights; flag: zero weights are skipped
                float4 *r = surf.rec + 2 * ((size_t)mesh * surf.per + surf.rec_first + q);
                r[0] = make_float4(d.x * surf.coef, d.y * surf.coef, d.z * surf.coef, 1.f);
                r[1] = make_float4(w.x, w.y, w.z, 0.f);
            }
        }
    }
    TRI_PHASE(6); // epilogue (wave 0)
}

template <bool TRUNC, bool FIX6, int HS_WAVES>
__global__ __launch_bounds__(HS_WAVES * GEOM_WAVE) void tri_scan_grouped_kernel(const float *__restrict__ xyz, int b, int n, int m,
                                                                       TriGws ws, float *__restrict__ dist,
                                                                       int *__restrict__ point, int *__restrict__ index,
                                                                       SurfaceOut surf)
{
    __shared__ TriTileLds<HS_WAVES> lds;
    tri_scan_grouped_body<TRUNC, FIX6, HS_WAVES>(blockIdx.x, xyz, b, n, m, ws, dist, point, index, surf, lds);
}

// ---------------------------------------------------------------------------------------
// The fused surface scan: the point-to-triangle tiles of the two-level scan AND the Chamfer NN jobs of both
// directions in ONE launch (BASELINE.json north_star: "point-to-triangle projection fused into the same tile pass").
// A heterogeneous grid: workgroups [0, tri_blocks) run a tri tile, the rest an NN tile.  The two bodies are
// complementary -- the tri tile is a chain of short dependent phases (VALU 24 % busy, 27 KB of LDS), the NN tile is
// pure VALU issue (4 KB of LDS) -- so sharing the CUs hides the former's latency under the latter's arithmetic instead
// of running them back to back.  Tri tiles come first in the grid: they are the long ones.
// Measured alternatives (8-mesh shard, us incl. the 5 us prep launch; separate launches: 66.7): tri tiles first 56.9,
// NN tiles first 64.9, interleaved 8:16 70.3 -- the launch lasts as long as its last TRI tile (a 31 us chain), so all of
// them have to start at once; s_setprio(3) on the tri waves: no change (they are not starved of issue slots); an 80-VGPR
// build (3 workgroups per CU instead of 2, 13 registers spilled): 55.3.  PMC of this launch: 32.0 M VALU instructions
// = 40 T lane-ops/s, 0.51 of the 78.6 T spec issue rate and 0.78 of the 51.5 T the chip sustains on un-packed v_fma_f32
// (MI355X_MICROARCH.md: 103 TFLOP/s measured) -- the fused launch is VALU-issue bound, what is left is instruction count.

#ifdef SCAN_TILE_STAMPS
__device__ long long scan_tile_stamps[4 * 4096]; // per workgroup {start, end (100 MHz wall clock), kind, hardware id}
extern "C" int geom_probe_read_role_phases(long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(scan_role_phases), sizeof(long long) * 256);
}
extern "C" int geom_probe_read_scan_stamps(long long *host, int rows)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(scan_tile_stamps), sizeof(long long) * 4 * (size_t)rows);
}
#endif

// The FINALIZE of the surface loss (surface_gather.hip: the loss sum + the points ordered by face for the gather backward)
// as extra workgroups of the scan launch instead of a launch of its own (11.4 us of a 9-workgroup chain behind a 42 us
// launch whose meshes finish at different times): every tile signs off in its mesh's counter (triangle tiles) and in the
// launch's counter (every tile) once its stores of the three arrays the roles read are acknowledged (geom::store_agent);
// role workgroup r < b waits for mesh r's triangle tiles and orders that mesh, the last role waits for every tile and sums
// the loss.  The roles LEAD the grid (resident from the start, their scan-independent half done before the wait); they
// can never starve the tiles of slots (b + 1 workgroups against hundreds of slots; tiles wait for nothing), so the launch
// always drains.  The counters are zeroed by the prep launch and reset by their waiter.  Same finalize body as the stand-alone
// launch (finalize_body.h): same bits.
struct ScanTail {
    geom_finalize::FinalizeArgs fin;
    int *done;       // tail_counters(b) counters, TAIL_CTR_STRIDE ints apart; nullptr: no tail in this launch
    int lead;        // workgroups [0, lead) are finalize roles (roles rounded up to a multiple of 8), tiles follow
    int tiles;       // tile workgroups of the launch (triangle + Chamfer, padding included): what the loss role waits for
    int roles;       // b + 1 with ordering, 1 (the loss) without
    int expect_mesh; // triangle tiles per mesh
    int expect_job;  // Chamfer tiles per (direction, mesh)
};
// b + 1 role workgroups lead the launch and wait on slots the tiles need too: at most a quarter of a slot per CU goes to them
// (64 meshes on the 256 CUs of an MI355X; a partition with fewer CUs takes fewer)
inline int scan_tail_max_meshes()
{
    static int most[64] = {0}; // per device: a process may drive partitions of different sizes
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    if (!most[dev]) {
        int cus = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        most[dev] = cus >= 8 ? cus / 4 : 1;
    }
    return most[dev];
}
constexpr int SCAN_TAIL_LDS_INTS = 11776; // 46 KB: three workgroups per CU still fit (the BASELINE mesh needs 11 173)

// Completion counters (TAIL_CTR_STRIDE, tail_counters()), ONE 128-byte line each (every tile's sign-off is a memory-side atomic; 2 256 of them plus the
// pollers on one line slowed every tile of the launch by ~15 %): [0, b) triangle tiles of mesh i; [b, 2b) the Chamfer tiles of
// mesh i whose queries are the sampled points (the other direction's results are read by nobody inside the launch);
// [2b] ordering roles that are past their wait (the loss role's stand-in for the triangle tiles).

// which counter a tile signs (-1: a padding workgroup) -- computed in front of the tile's body so that ONE register lives
// across it (the geometry this takes -- b, n, the tile counts -- pushed scalar registers of the brute-force Chamfer loop
// into spills when it was evaluated behind the body: 49 -> 59 us)
__device__ __forceinline__ int scan_tile_counter(int bid, int tri_blocks, int b, int n, int nn_tiles)
{
    int job, tile;
    if (bid < tri_blocks) return geom::xcd_assign(bid, b, (n + TRI_QUERIES - 1) / TRI_QUERIES, job, tile) ? job : -1;
    // Chamfer tiles: only the direction the loss needs (the sampled points' squared distances) is counted
    return geom::xcd_assign(bid - tri_blocks, 2 * b, nn_tiles, job, tile) && geom::nn_job_dir(job, b) == 1 ? b + job % b : -1;
}

__device__ __forceinline__ void scan_tile_done(int *done, int counter)
{
    // every wave: its agent-scope stores (geom::store_agent: the three arrays the role workgroups read) are
    // acknowledged before the count moves; everything else this tile wrote is read by later launches only
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0 && counter >= 0)
        __hip_atomic_fetch_add(done + (size_t)counter * TAIL_CTR_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0 of a role workgroup: lane i < count polls counter first + i until it reaches its target, all lanes together (a
// counter that is already there costs no round trip of its own), then resets it for the next launch on this workspace.
// Returns the largest count seen, or -1 when TAIL_SPIN_LIMIT polls (seconds) went by: a role never hangs the device -- it
// gives up, and the launch's loss comes out as NaN (scan_tail_role).
constexpr int TAIL_SPIN_LIMIT = 1 << 21;
constexpr int TAIL_GAVE_UP = 1 << 20; // added to the ordering roles' count by a role that gave up
__device__ __forceinline__ int tail_wait_counters(int *done, int first, int count, int expect)
{
    int seen = 0;
    for (int c0 = 0; c0 < count; c0 += GEOM_WAVE) {
        const int i = c0 + (int)threadIdx.x;
        int *ctr = done + (size_t)(first + i) * TAIL_CTR_STRIDE;
        bool ok = i >= count;
        int mine = 0, polls = 0;
        while (true) {
            if (!ok) {
                mine = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = mine >= expect;
            }
            if (__all(ok)) break;
            if (++polls > TAIL_SPIN_LIMIT) { // give up; the counter goes back to zero all the same (a tile that signs off later
                                            // leaves a stale count: see geom_hip.h -- re-run the prepare launch after a NaN loss)
                if (i < count) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return -1;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        if (i < count) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int d = GEOM_WAVE / 2; d > 0; d >>= 1) mine = max(mine, __shfl_xor(mine, d, GEOM_WAVE));
        seen = max(seen, mine);
    }
    return seen;
}

struct ScanTailWait {
    int *done;
    int *poison; // LDS: set when this role's wait gave up (the loss role: also when an ordering role did)
    int role, b, ordering, expect_mesh, expect_job; // role == b: the loss role
    __device__ __forceinline__ bool operator()() const
    {
        if (threadIdx.x < GEOM_WAVE) {
            int *roles_past = done + (size_t)2 * b * TAIL_CTR_STRIDE;
            if (role < b) { // mesh `role`'s triangle tiles; then tell the loss role
                const int got = tail_wait_counters(done, role, 1, expect_mesh);
                if (threadIdx.x == 0) {
                    __hip_atomic_fetch_add(roles_past, got < 0 ? TAIL_GAVE_UP + 1 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *poison = got < 0;
                }
            } else {        // the Chamfer tiles whose distances are summed, and the triangle tiles directly or through their ordering roles
                int got = tail_wait_counters(done, b, b, expect_job);
                if (got >= 0) got = ordering ? tail_wait_counters(done, 2 * b, 1, b) : tail_wait_counters(done, 0, b, expect_mesh);
                if (threadIdx.x == 0) *poison = got < 0 || got >= TAIL_GAVE_UP;
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // nothing stale from this XCD's L2
#ifdef SCAN_TILE_STAMPS
        if (threadIdx.x == 0) scan_tile_stamps[4 * (size_t)blockIdx.x + 3] = wall_clock64();
#endif
        return *poison == 0; // (every thread, behind the barrier above)
    }
};

// role r of the launch's leading block of workgroups (roles rounded up to a multiple of 8; they are resident from the
// start: what does not depend on the scans -- zeroing, binning the sampled points by their drawn faces -- is done before
// the wait)
__device__ __forceinline__ void scan_tail_role(const ScanTail &t, int role, int *lds_ints)
{
    if (role >= t.roles) return; // padding (the tile workgroups keep their blockIdx % 8 = XCD mapping)
    const int b = t.fin.b;
    const bool loss_role = role == t.roles - 1;
    int *poison = lds_ints + SCAN_TAIL_LDS_INTS - 1; // behind what the body uses (the host leaves this int free)
    const ScanTailWait wait{t.done, poison, loss_role ? b : role, b, t.roles > 1, t.expect_mesh, t.expect_job};
    // the variant that keeps the points' faces / arrival slots in the global scratch (loads batched 6-12 deep), the
    // record-forming code compiled out.  The register variant does not survive this launch's budget of 80 registers: with 16
    // points per thread it spilled 2 454 registers (55 us to bin 3000 points), with 12 and the lean binning code it fitted in
    // one build (12.4 -> 13 us behind the wait: no gain) and spilled again in the next (20 us to bin, 70 us launch)
    // (a wait that gave up: the body returns at once -- the loss comes out as NaN, the role's status word of the backward
    // scratch says "not ordered", and nothing of the incomplete results is read)
    geom_finalize::surface_finalize_body<false, 8 * GEOM_WAVE, ScanTailWait, 16, true>(t.fin, lds_ints, loss_role ? b : role, wait);
}

// CULL: the Chamfer tiles take the culled scan (nn_culled_body).  Its tiles are latency chains, not issue-bound loops; the two
// bodies' LDS is overlaid (a workgroup is one or the other: 46 KB instead of 27 + 43).  Register budget: 5 waves per SIMD =
// 96 registers -- the triangle body wants 104 on its own, fits 96 without a spill, and at the round-4 cap of 80 (6 waves per
// SIMD, three workgroups per CU) it spilled 13 registers and re-read 12 MB of scratch per launch.  A/B on one box, alternating
// (tools/probe/scan_cap_ab.sh): 49.8 / 49.2 / 48.5 us at 96 registers against 50.7 / 49.4 / 49.2 at 80 -- the third workgroup
// per CU bought nothing once the Chamfer tiles were culled.  The brute-force variant keeps two workgroups per CU: it IS
// issue-bound, and capping its registers only added spills (round 2: 55.3 against 48 us).
template <bool FIX6, bool FMA, bool CULL>
__global__ __launch_bounds__(8 * GEOM_WAVE, CULL ? 5 : 1) void surface_scan_kernel(const float *__restrict__ xyz, int b, int n, int m,
                                                                                  TriGws ws, float *__restrict__ dist,
                                                                                  int *__restrict__ point, int *__restrict__ index,
                                                                                  SurfaceOut surf, NNJob job, NNRecords rr,
                                                                                  int tri_blocks, NNCull cull, ScanTail tail)
{
    static_assert(NNS_THREADS == 8 * GEOM_WAVE, "both bodies are written for 8-wave workgroups");
#ifdef SCAN_TILE_STAMPS // tools/probe only: when each tile of the launch started and ended, and where (see scan_tile_stamps.py)
    const long long stamp_t0 = wall_clock64();
    struct StampOnExit {
        long long t0;
        int tri_blocks, lead;
        __device__ ~StampOnExit()
        {
            __syncthreads();
            if (threadIdx.x == 0) {
                long long *row = scan_tile_stamps + 4 * (size_t)blockIdx.x;
                row[0] = t0, row[1] = wall_clock64(), row[2] = (int)blockIdx.x < lead ? 2 : ((int)blockIdx.x - lead < tri_blocks ? 0 : 1);
                if ((int)blockIdx.x >= lead) row[3] = __smid(); // roles: the end of their wait (ScanTailWait)
            }
        }
    } stamp_on_exit{stamp_t0, tri_blocks, tail.lead};
#endif
    if constexpr (CULL) {
        __shared__ union Lds {
            TriTileLds<8> tri;
            NNCullLds nn;
            int fin[SCAN_TAIL_LDS_INTS];
            __device__ Lds() {}
        } lds;
        if ((int)blockIdx.x < tail.lead) {
            scan_tail_role(tail, blockIdx.x, lds.fin);
            return;
        }
        const int bid = (int)blockIdx.x - tail.lead; // tile index (tail.lead = 0 without a finalize tail)
        const int sign = tail.done ? scan_tile_counter(bid, tri_blocks, b, n, tail.expect_job) : -1;
        // (s_setprio(3) on either kind of tile: 41.6 / 43.3 against 42.4 us -- within the noise, not kept)
        if (bid < tri_blocks) tri_scan_grouped_body<false, FIX6, 8>(bid, xyz, b, n, m, ws, dist, point, index, surf, lds.tri);
        else nn_culled_body<FMA>(job, cull, bid - tri_blocks, rr, lds.nn);
        if (tail.done) scan_tile_done(tail.done, sign);
    } else { // the brute-force Chamfer tiles: no finalize tail (its scalar registers and LDS cost this issue-bound variant 10 us)
        __shared__ TriTileLds<8> lds;
        const int bid = blockIdx.x;
        if (bid < tri_blocks) tri_scan_grouped_body<false, FIX6, 8>(bid, xyz, b, n, m, ws, dist, point, index, surf, lds);
        else nn_scalar_body<FMA>(job, bid - tri_blocks, rr);
    }
}

template <bool INDEXED, bool TRUNC, bool FIX6>
int launch_grouped_variant(const TriJob &job, const TriGws &ws, const int *order, hipStream_t s, const SurfaceOut &surf)
{
    const int prep_items = ws.split > 1 && job.n > ws.m_pad ? job.n : ws.m_pad;
    hipLaunchKernelGGL((tri_prep_grouped_kernel<INDEXED, TRUNC, FIX6>), dim3((prep_items + 255) / 256, job.b), dim3(256), 0, s,
                       job, ws, order);
    const int qtiles = (job.n + TRI_QUERIES - 1) / TRI_QUERIES;
    const SurfaceOut so = ws.split > 1 ? NO_SURFACE : surf;
    const dim3 grid(geom::xcd_grid(job.b, qtiles * ws.split));
    if ((int64_t)job.b * qtiles * ws.split >= 256) // enough workgroups to fill the chip: fewer, longer-lived waves
        hipLaunchKernelGGL((tri_scan_grouped_kernel<TRUNC, FIX6, 8>), grid, dim3(8 * GEOM_WAVE), 0, s, job.xyz, job.b, job.n,
                           job.m, ws, job.dist, job.point, job.index, so);
    else
        hipLaunchKernelGGL((tri_scan_grouped_kernel<TRUNC, FIX6, HS_MAX_WAVES>), grid, dim3(HS_MAX_WAVES * GEOM_WAVE), 0, s,
                           job.xyz, job.b, job.n, job.m, ws, job.dist, job.point, job.index, so);
    if (ws.split > 1)
        hipLaunchKernelGGL((tri_finalize_kernel<TRUNC, FIX6>), dim3((job.n + 255) / 256, job.b), dim3(256), 0, s, job.xyz,
                           job.n, job.m, ws.first, (size_t)3, ws.keys, job.dist, job.point, job.index);
    return geom::launch_status();
}

// How many workgroups share a query tile.  A workgroup holds 16 waves and at most two fit a CU.  With
// fewer query tiles than CUs (1-5 meshes of 3000 points) most of the chip would idle, so the triangle
// range of a tile is split over up to 4 workgroups (measured: 1 mesh 51 -> 26 us, 3 meshes 72 -> 50 us).
// Every part repeats the hint phase (~10 us), so from 256 tiles on the split costs more than the better
// balance returns (8 meshes: 73 us unsplit, 86 us at split 3) and parts stay >= 1024 triangles.
inline int ws_split(int b, int n, int m_pad)
{
    const int64_t tiles = (int64_t)b * ((n + TRI_QUERIES - 1) / TRI_QUERIES);
    if (tiles <= 0 || tiles >= 256) return 1;
    int split = (int)((512 + tiles - 1) / tiles);
    if (split > m_pad / 1024) split = m_pad / 1024;
    if (split > 4) split = 4;
    return split < 1 ? 1 : split;
}

inline size_t ws_bytes_needed(int b, int n, int m_pad)
{
    // member spheres + corners, group spheres, triangle-0 corners, merged keys
    return (size_t)b * m_pad * 4 * sizeof(float4) + (size_t)b * (m_pad / GRP) * sizeof(float4) + (size_t)b * 3 * sizeof(float4) +
           (size_t)b * n * 8;
}

// *fused = whether the scan wrote the point-to-surface outputs itself (else the caller launches the separate kernel)
template <bool INDEXED>
int launch_tri_ws(const TriJob &job, const int *order, unsigned flags, void *workspace, size_t ws_bytes, void *stream,
                  const SurfaceOut &surf = NO_SURFACE, bool *fused = nullptr)
{
    if (fused) *fused = false;
    const int m_pad = ws_pad(job.m);
    if (!workspace || ws_bytes < ws_bytes_needed(job.b, job.n, m_pad) || ((uintptr_t)workspace & 15)) return GEOM_EINVAL;
    float4 *base = static_cast<float4 *>(workspace);
    if (order) { // coherent order supplied: two-level scan
        float4 *grp = base + (size_t)job.b * m_pad * 4;
        float4 *first = grp + (size_t)job.b * (m_pad / GRP);
        TriGws gws{base, base + (size_t)job.b * m_pad, grp, first, reinterpret_cast<unsigned long long *>(first + (size_t)job.b * 3),
                   m_pad, ws_split(job.b, job.n, m_pad)};
        hipStream_t gs = static_cast<hipStream_t>(stream);
        const bool gtrunc = flags & GEOM_FLAG_REF_TAIL_TRUNC, gfix6 = flags & GEOM_FLAG_FIX_REGION6;
        if (fused) *fused = gws.split == 1 && surf.sqdist != nullptr;
        if (gtrunc && gfix6) return launch_grouped_variant<INDEXED, true, true>(job, gws, order, gs, surf);
        if (gtrunc) return launch_grouped_variant<INDEXED, true, false>(job, gws, order, gs, surf);
        if (gfix6) return launch_grouped_variant<INDEXED, false, true>(job, gws, order, gs, surf);
        return launch_grouped_variant<INDEXED, false, false>(job, gws, order, gs, surf);
    }
    TriWs ws{base, base + (size_t)job.b * m_pad, reinterpret_cast<unsigned long long *>(base + (size_t)job.b * m_pad * 4),
             m_pad, ws_split(job.b, job.n, m_pad)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool trunc = flags & GEOM_FLAG_REF_TAIL_TRUNC, fix6 = flags & GEOM_FLAG_FIX_REGION6;
    if (trunc && fix6) return launch_ws_variant<INDEXED, true, true>(job, ws, s);
    if (trunc) return launch_ws_variant<INDEXED, true, false>(job, ws, s);
    if (fix6) return launch_ws_variant<INDEXED, false, true>(job, ws, s);
    return launch_ws_variant<INDEXED, false, false>(job, ws, s);
}

template <bool INDEXED>
int launch_tri(const TriJob &job, unsigned flags, void *stream)
{
    dim3 grid((job.n + TRI_QUERIES - 1) / TRI_QUERIES, job.b, 1);
    dim3 block(TRI_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool trunc = flags & GEOM_FLAG_REF_TAIL_TRUNC, fix6 = flags & GEOM_FLAG_FIX_REGION6;
    if (!(flags & GEOM_FLAG_TRI_BRUTE_FORCE)) {
        if (trunc && fix6)
            hipLaunchKernelGGL((tri_distance_culled_kernel<INDEXED, true, true>), grid, block, 0, s, job);
        else if (trunc)
            hipLaunchKernelGGL((tri_distance_culled_kernel<INDEXED, true, false>), grid, block, 0, s, job);
        else if (fix6)
            hipLaunchKernelGGL((tri_distance_culled_kernel<INDEXED, false, true>), grid, block, 0, s, job);
        else
            hipLaunchKernelGGL((tri_distance_culled_kernel<INDEXED, false, false>), grid, block, 0, s, job);
        return geom::launch_status();
    }
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
    if (b > 65535 || m >= (1 << 26)) return GEOM_ETOOBIG;
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
    if (b > 65535 || nf >= (1 << 26)) return GEOM_ETOOBIG;
    TriJob job{xyz, nullptr, nullptr, nullptr, verts, faces, dist, point, index, b, n, nf, nv};
    return launch_tri<true>(job, flags, stream);
}

// where the finalize tail's completion counters sit in a workspace of geom_tri_distance_workspace_bytes(b, n, m) bytes
// (int32 words, 32 apart: [0, b) triangle tiles per mesh, [b, 2b) Chamfer tiles per mesh, [2b] ordering roles) -- for tests
// that provoke the roles' give-up path; 0 when the counters do not fit (no tail for such a call)
extern "C" size_t geom_surface_tail_counters_offset(int b, int n, int m)
{
    if (b <= 0 || n <= 0 || m <= 0 || !tail_counters_fit(b, n)) return 0;
    const int m_pad = ws_pad(m);
    return ((size_t)b * m_pad * 4 + (size_t)b * (m_pad / GRP) + (size_t)b * 3) * sizeof(float4);
}

extern "C" size_t geom_tri_distance_workspace_bytes(int b, int n, int m)
{
    if (b <= 0 || m <= 0 || n < 0) return 0;
    return ws_bytes_needed(b, n, ws_pad(m));
}

extern "C" int geom_tri_distance_ws_f32(int b, int n, const float *xyz, int m,
                                        const float *tri1, const float *tri2, const float *tri3, const int *order,
                                        float *dist, int *point, int *index, unsigned flags,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    if (b < 0 || n < 0 || m < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (m == 0) return GEOM_EINVAL;
    if (!xyz || !tri1 || !tri2 || !tri3 || !dist || !point || !index) return GEOM_EINVAL;
    if (b > 65535 || m >= (1 << 26)) return GEOM_ETOOBIG;
    TriJob job{xyz, tri1, tri2, tri3, nullptr, nullptr, dist, point, index, b, n, m, 0};
    if (flags & GEOM_FLAG_TRI_BRUTE_FORCE) return launch_tri<false>(job, flags, stream);
    return launch_tri_ws<false>(job, order, flags, workspace, workspace_bytes, stream);
}

extern "C" int geom_tri_distance_indexed_ws_f32(int b, int n, const float *xyz, int nv, const float *verts,
                                                int nf, const int64_t *faces, const int *order,
                                                float *dist, int *point, int *index, unsigned flags,
                                                void *workspace, size_t workspace_bytes, void *stream)
{
    if (b < 0 || n < 0 || nf < 0 || nv < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (nf == 0 || nv == 0) return GEOM_EINVAL;
    if (!xyz || !verts || !faces || !dist || !point || !index) return GEOM_EINVAL;
    if (b > 65535 || nf >= (1 << 26)) return GEOM_ETOOBIG;
    TriJob job{xyz, nullptr, nullptr, nullptr, verts, faces, dist, point, index, b, n, nf, nv};
    if (flags & GEOM_FLAG_TRI_BRUTE_FORCE) return launch_tri<true>(job, flags, stream);
    return launch_tri_ws<true>(job, order, flags, workspace, workspace_bytes, stream);
}

// tri_distance + the point-to-surface quantities of the winner in one call: the two-level scan writes them from its
// epilogue; the other scans (no order, brute force, split query tiles) are followed by the separate kernel.
extern "C" int geom_tri_surface_fwd_f32(int b, int n, const float *xyz, int nv, const float *verts, int nf,
                                        const int64_t *faces, const int *order, float *dist, int *point, int *index,
                                        float *sqdist, float *closest, float *weights, unsigned flags, void *workspace,
                                        size_t workspace_bytes, void *stream)
{
    if (b < 0 || n < 0 || nf < 0 || nv < 0) return GEOM_EINVAL;
    if (b == 0 || n == 0) return 0;
    if (nf == 0 || nv == 0) return GEOM_EINVAL;
    if (!xyz || !verts || !faces || !dist || !point || !index || !sqdist || !closest || !weights) return GEOM_EINVAL;
    if (b > 65535 || nf >= (1 << 26)) return GEOM_ETOOBIG;
    TriJob job{xyz, nullptr, nullptr, nullptr, verts, faces, dist, point, index, b, n, nf, nv};
    bool fused = false;
    int rc;
    if (flags & GEOM_FLAG_TRI_BRUTE_FORCE) rc = launch_tri<true>(job, flags, stream);
    else rc = launch_tri_ws<true>(job, order, flags, workspace, workspace_bytes, stream,
                                  SurfaceOut{verts, faces, nv, sqdist, closest, weights}, &fused);
    if (rc != 0 || fused) return rc;
    return geom_p2tri_loss_fwd_f32(b, n, xyz, nv, verts, nf, faces, point, index, sqdist, closest, weights, stream);
}

namespace {
template <bool FMA>
__global__ __launch_bounds__(NNS_THREADS) void nn_records_kernel(NNJob job, NNRecords rr) { nn_scalar_body<FMA>(job, blockIdx.x, rr); }
} // namespace

// The two arg-min scans of the surface loss in one call (utils.py:451 + 470: chamfer_dist, then tri_dist on the gathered
// corners).  gt [b,n_gt,3] against the sampled points [b,num,3]: NN both ways (sq_gt / idx_p for the gt points, sq_pred /
// idx_g for the sampled points, as geom_chamfer_nn_f32(gt, points)); and, when verts != NULL, gt against the mesh
// (tri_dist / option / index + the point-to-surface quantities sq / closest / weights, as geom_tri_surface_fwd_f32).
// With a coherent triangle order, the default arithmetic flags and enough query tiles to fill the chip both scans run
// as ONE heterogeneous launch after the triangle-record prep; otherwise as the separate launches.  order_scratch (may be
// NULL): the finalize scratch of geom_surface_finalize_f32 -- the scans then also write every point's gradient record
// into it (u, v: the sampled points' draws; coef_*: the two gradient coefficients); *records_written tells the caller
// whether they did (the split / brute-force / truncation variants leave them to the finalize pass).
extern "C" int geom_surface_scan_f32(int b, int n_gt, const float *gt, int num, const float *points, float *sq_gt,
                                     int *idx_p, float *sq_pred, int *idx_g, int nv, const float *verts, int nf,
                                     const int64_t *faces, const int *tri_order, float *tri_dist, int *option, int *index,
                                     float *sq, float *closest, float *weights, const float *u, const float *v,
                                     float coef_sample, float coef_other, int *order_scratch, unsigned flags,
                                     void *workspace, size_t workspace_bytes, int *records_written,
                                     const geom_surface_cull *cull, geom_surface_tail *tail, void *stream)
{
    if (records_written) *records_written = 0;
    if (tail) tail->finalized = 0;
    if (b < 0 || n_gt < 0 || num < 0 || nf < 0 || nv < 0) return GEOM_EINVAL;
    if (b == 0) return 0;
    if (n_gt == 0 || num == 0) return GEOM_EINVAL;
    if (!gt || !points || !sq_gt || !idx_p || !sq_pred || !idx_g) return GEOM_EINVAL;
    const bool tri = verts != nullptr;
    if (tri && (!faces || !tri_dist || !option || !index || !sq || !closest || !weights || nf == 0 || nv == 0)) return GEOM_EINVAL;
    if (order_scratch && (!u || !v || ((uintptr_t)order_scratch & 15))) return GEOM_EINVAL;
    if (b > 65535 || nf >= (1 << 26)) return GEOM_ETOOBIG;
    const unsigned nn_flags = flags & (GEOM_FLAG_REF_TAIL_TRUNC | GEOM_FLAG_NN_FMA);
    const unsigned tri_flags = flags & (GEOM_FLAG_REF_TAIL_TRUNC | GEOM_FLAG_FIX_REGION6 | GEOM_FLAG_TRI_BRUTE_FORCE);
    if ((nn_flags & GEOM_FLAG_REF_TAIL_TRUNC) && (nn_flags & GEOM_FLAG_NN_FMA)) return GEOM_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t cap = (int64_t)num + n_gt;
    float4 *rec = order_scratch ? reinterpret_cast<float4 *>(order_scratch + geom_surface_order_ints(b, nf, cap)) : nullptr;
    NNJob job{gt, points, sq_gt, sq_pred, idx_p, idx_g, b, n_gt, num};
    const int longer = n_gt > num ? n_gt : num;
    const int nn_tiles = (longer + NN_QUERIES - 1) / NN_QUERIES;
    if ((int64_t)geom::NUM_XCD * nn_tiles * ((2 * (int64_t)b + 7) / 8) > 0x3fffffffLL) return GEOM_ETOOBIG;
    const unsigned nn_blocks = geom::xcd_grid(2 * b, nn_tiles);
    const bool fma = nn_flags & GEOM_FLAG_NN_FMA;
    // records of the NN side: the sampled points always; the gt points through their nearest sample only without a tri scan
    NNRecords rr{rec, u, v, coef_sample, coef_other, (int)(tri ? cap : cap), tri ? 0 : 1};

    if (tri) {
        const int m_pad = ws_pad(nf);
        const int split = ws_split(b, n_gt, m_pad);
        const bool fusable = tri_order && !(tri_flags & (GEOM_FLAG_REF_TAIL_TRUNC | GEOM_FLAG_TRI_BRUTE_FORCE)) &&
                             !(nn_flags & GEOM_FLAG_REF_TAIL_TRUNC) && split == 1 &&
                             (int64_t)b * ((n_gt + TRI_QUERIES - 1) / TRI_QUERIES) >= 256;
        if (fusable) {
            if (!workspace || workspace_bytes < ws_bytes_needed(b, n_gt, m_pad) || ((uintptr_t)workspace & 15)) return GEOM_EINVAL;
            float4 *base = static_cast<float4 *>(workspace);
            float4 *grp = base + (size_t)b * m_pad * 4;
            float4 *first = grp + (size_t)b * (m_pad / GRP);
            TriGws gws{base, base + (size_t)b * m_pad, grp, first, reinterpret_cast<unsigned long long *>(first + (size_t)b * 3), m_pad, 1};
            TriJob tj{gt, nullptr, nullptr, nullptr, verts, faces, tri_dist, option, index, b, n_gt, nf, nv};
            const bool fix6 = tri_flags & GEOM_FLAG_FIX_REGION6;
            if (flags & GEOM_FLAG_TRI_WS_READY) {} // records of an earlier call on the same mesh: no prep launch
            else if (fix6) hipLaunchKernelGGL((tri_prep_grouped_kernel<true, false, true>), dim3((m_pad + 255) / 256, b), dim3(256), 0, s, tj, gws, tri_order);
            else hipLaunchKernelGGL((tri_prep_grouped_kernel<true, false, false>), dim3((m_pad + 255) / 256, b), dim3(256), 0, s, tj, gws, tri_order);
            const int qtiles = (n_gt + TRI_QUERIES - 1) / TRI_QUERIES;
            const unsigned tri_blocks = geom::xcd_grid(b, qtiles);
            SurfaceOut so{verts, faces, nv, sq, closest, weights, rec, coef_other, (int)cap, num};
            // the Chamfer tiles take the culled scan when the caller handed over the indices of both clouds (the gt
            // cloud's: static; the sampled points': written by geom_surface_prepare_f32 of the same step)
            const bool culled = cull && cull->gt_index && cull->sample_index;
            if (culled && (((uintptr_t)cull->gt_index | (uintptr_t)cull->sample_index) & 15)) return GEOM_EINVAL;
            // the finalize pass as extra workgroups of this launch (ScanTail) when the caller asks for it and a mesh's
            // faces + points fit the launch's LDS; otherwise the caller launches geom_surface_finalize_f32 as before
            ScanTail st{};
            if (tail && (!tail->loss || !tail->choices || (tail->want_order && !rec))) return GEOM_EINVAL;
            if (tail && culled) { // the launch with the culled Chamfer tiles only (three workgroups per CU, latency-bound tiles)
                const int64_t per64 = (int64_t)num + n_gt;
                // (b <= scan_tail_max_meshes(): the roles wait from the start of the launch on slots the tiles need too -- a few
                // dozen of 768 cost nothing, hundreds would crowd the tiles out; larger batches finalize in a launch of their own)
                if (per64 <= 0x3fffffff && b <= scan_tail_max_meshes() && tail_counters_fit(b, n_gt) &&
                    geom_finalize::finalize_lds_ints(nf, (int)per64, 8 * GEOM_WAVE, tail->want_order != 0) < (size_t)SCAN_TAIL_LDS_INTS) {
                    int *off = order_scratch, *seg = off ? off + (int64_t)b * (nf + 1) : nullptr;
                    int *pface = seg ? seg + (int64_t)b * cap : nullptr, *slot = pface ? pface + (int64_t)b * cap : nullptr;
                    st.fin = geom_finalize::FinalizeArgs{tail->choices, u, v, points, gt, idx_g, nullptr, index, closest, weights, sq_pred, sq,
                                                         tail->scale_sample, tail->scale_other, coef_sample, coef_other, b, nf, num, n_gt,
                                                         geom_finalize::OTHER_TRI, (int)per64, tail->want_order ? 1 : 0, rec ? 1 : 0,
                                                         off, seg, pface, slot, rec, tail->loss,
                                                         order_scratch ? order_scratch + geom_surface_status_offset(b, nf, cap) : nullptr};
                    st.done = reinterpret_cast<int *>(gws.keys);
                    st.tiles = (int)(tri_blocks + nn_blocks);
                    st.roles = tail->want_order ? b + 1 : 1;
                    st.lead = (st.roles + geom::NUM_XCD - 1) / geom::NUM_XCD * geom::NUM_XCD;
                    st.expect_mesh = qtiles;
                    st.expect_job = nn_tiles;
                    tail->finalized = 1;
                }
            }
            const dim3 grid(st.lead + tri_blocks + nn_blocks), block(8 * GEOM_WAVE);
            NNCull nc{};
            if (culled) {
                const float4 *s1 = reinterpret_cast<const float4 *>(cull->gt_index), *s2 = reinterpret_cast<const float4 *>(cull->sample_index);
                nc = NNCull{reinterpret_cast<const float *>(s1 + (size_t)b * (n_gt / NNS_GROUP)),
                            reinterpret_cast<const float *>(s2 + (size_t)b * (num / NNS_GROUP)), cull->gt_order, nullptr, s1, s2}; // samples: identity order
            }
#define GEOM_LAUNCH_SCAN(F6, FM, CU)                                                                                            \
    hipLaunchKernelGGL((surface_scan_kernel<F6, FM, CU>), grid, block, 0, s, gt, b, n_gt, nf, gws, tri_dist, option, index, so, job, \
                       rr, (int)tri_blocks, nc, st)
#define GEOM_LAUNCH_SCAN2(F6, FM)                                                                                               \
    do {                                                                                                                        \
        if (culled) GEOM_LAUNCH_SCAN(F6, FM, true);                                                                             \
        else GEOM_LAUNCH_SCAN(F6, FM, false);                                                                                   \
    } while (0)
            if (fix6 && fma) GEOM_LAUNCH_SCAN2(true, true);
            else if (fix6) GEOM_LAUNCH_SCAN2(true, false);
            else if (fma) GEOM_LAUNCH_SCAN2(false, true);
            else GEOM_LAUNCH_SCAN2(false, false);
#undef GEOM_LAUNCH_SCAN2
#undef GEOM_LAUNCH_SCAN
            if (records_written) *records_written = rec != nullptr;
            return geom::launch_status();
        }
        // the other tri variants: separate launches, records left to the finalize pass
        const int rc = geom_tri_surface_fwd_f32(b, n_gt, gt, nv, verts, nf, faces, tri_order, tri_dist, option, index, sq, closest,
                                                weights, tri_flags, workspace, workspace_bytes, stream);
        if (rc != 0) return rc;
        return geom_chamfer_nn_f32(b, n_gt, gt, num, points, sq_gt, idx_p, sq_pred, idx_g, nn_flags, stream);
    }
    // no tri scan (batch_point_to_point): the NN launch writes both kinds of records
    if (nn_flags & GEOM_FLAG_REF_TAIL_TRUNC) return geom_chamfer_nn_f32(b, n_gt, gt, num, points, sq_gt, idx_p, sq_pred, idx_g, nn_flags, stream);
    if (fma) hipLaunchKernelGGL(nn_records_kernel<true>, dim3(nn_blocks), dim3(NNS_THREADS), 0, s, job, rr);
    else hipLaunchKernelGGL(nn_records_kernel<false>, dim3(nn_blocks), dim3(NNS_THREADS), 0, s, job, rr);
    if (records_written) *records_written = rec != nullptr;
    return geom::launch_status();
}

// Draws + sampled points of batch_sample (exactly geom_draw_samples_rng_f32) and, when the surface scan of the same
// step will run fused (same conditions as in geom_surface_scan_f32: coherent tri_order, no truncation / brute-force
// flags, >= 256 query tiles of n_gt points), the triangle records of that scan -- in ONE launch; *prepared = 1 then and
// the scan is called with GEOM_FLAG_TRI_WS_READY.  Otherwise only the draws are made (*prepared = 0).
extern "C" int geom_surface_prepare_f32(int b, int nv, const float *verts, int nf, const int64_t *faces, int num,
                                        uint64_t *rng_state, int64_t *choices, float *u, float *v, float *points, int n_gt,
                                        const int *tri_order, unsigned flags, void *workspace, size_t workspace_bytes,
                                        int *prepared, const geom_surface_cull *cull, void *stream)
{
    if (prepared) *prepared = 0;
    if (b < 0 || nv < 0 || nf < 0 || num < 0 || n_gt < 0) return GEOM_EINVAL;
    if (nf > DRAW_MAX_FACES) return GEOM_EUNSUPPORTED;
    if (b == 0 || num == 0) return 0;
    if (nf == 0 || !verts || !faces || !rng_state || !choices || !u || !v) return GEOM_EINVAL;
    if (b > 65535) return GEOM_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int draw_chunks = (num + DRAW_THREADS - 1) / DRAW_THREADS;
    const int m_pad = ws_pad(nf);
    const bool fusable = tri_order && workspace && n_gt > 0 &&
                         !(flags & (GEOM_FLAG_REF_TAIL_TRUNC | GEOM_FLAG_TRI_BRUTE_FORCE)) && ws_split(b, n_gt, m_pad) == 1 &&
                         (int64_t)b * ((n_gt + TRI_QUERIES - 1) / TRI_QUERIES) >= 256;
    if (!fusable) return geom_draw_samples_rng_f32(b, nv, verts, nf, faces, num, rng_state, choices, u, v, points, stream);
    if (workspace_bytes < ws_bytes_needed(b, n_gt, m_pad) || ((uintptr_t)workspace & 15)) return GEOM_EINVAL;
    float4 *base = static_cast<float4 *>(workspace);
    float4 *grp = base + (size_t)b * m_pad * 4;
    float4 *first = grp + (size_t)b * (m_pad / GRP);
    TriGws gws{base, base + (size_t)b * m_pad, grp, first, reinterpret_cast<unsigned long long *>(first + (size_t)b * 3), m_pad, 1};
    TriJob tj{nullptr, nullptr, nullptr, nullptr, verts, faces, nullptr, nullptr, nullptr, b, n_gt, nf, nv};
    const int prep_chunks = (m_pad + DRAW_THREADS - 1) / DRAW_THREADS;
    unsigned long long *st = reinterpret_cast<unsigned long long *>(rng_state);
    // the culled Chamfer scan of the same step wants the samples in a visiting order: the draw workgroups then generate them
    // in the order of their faces' positions in tri_order (sorted uniforms from exponential spacings: draw_body.h) and write
    // the samples' index -- run spheres + visiting-order copy
    const bool sorted = cull && cull->sample_index && num >= NN_QUERIES && num < DRAW_SORT_CHUNKS * DRAW_THREADS;
    if (sorted && ((uintptr_t)cull->sample_index & 15)) return GEOM_EINVAL;
    const int draw_blocks = draw_chunks * b;
    const dim3 grid(draw_blocks + prep_chunks * b), block(DRAW_THREADS);
    if (sorted) {
        float4 *sph = reinterpret_cast<float4 *>(cull->sample_index);
        DrawSort srt{tri_order, cull->faces_in_order, reinterpret_cast<float *>(sph + (size_t)b * (num / NNS_GROUP)), sph};
        if (flags & GEOM_FLAG_FIX_REGION6)
            hipLaunchKernelGGL((surface_prepare_kernel<true, true>), grid, block, 0, s, draw_blocks, draw_chunks, nv, verts, nf, faces,
                               num, st, choices, u, v, points, tj, gws, tri_order, prep_chunks, srt);
        else
            hipLaunchKernelGGL((surface_prepare_kernel<false, true>), grid, block, 0, s, draw_blocks, draw_chunks, nv, verts, nf, faces,
                               num, st, choices, u, v, points, tj, gws, tri_order, prep_chunks, srt);
        if (prepared) *prepared = 3; // bit 0: triangle records; bit 1: the sampled points' index
        return geom::launch_status();
    }
    if (flags & GEOM_FLAG_FIX_REGION6)
        hipLaunchKernelGGL((surface_prepare_kernel<true, false>), grid, block, 0, s, draw_blocks, draw_chunks, nv, verts, nf, faces, num,
                           st, choices, u, v, points, tj, gws, tri_order, prep_chunks, DrawSort{});
    else
        hipLaunchKernelGGL((surface_prepare_kernel<false, false>), grid, block, 0, s, draw_blocks, draw_chunks, nv, verts, nf, faces, num,
                           st, choices, u, v, points, tj, gws, tri_order, prep_chunks, DrawSort{});
    if (prepared) *prepared = 1;
    return geom::launch_status();
}
