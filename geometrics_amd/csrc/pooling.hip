// batched_pooling for gfx950: project every vertex into the image and
// bilinearly pool the encoder's feature maps at that pixel -- the 960 image features of the 963-wide
// 0N-GCN input.
//
// The reference (utils.py:316-389) does this per feature map with ~10 eager ops: a permuted copy of
// the whole map, four index_selects, eight broadcast products, a cat; ~40 kernels per call, three
// calls per step.  Here ONE kernel per direction handles all maps:
//   * a workgroup owns 64 vertices of one mesh; the camera projection is done once per vertex;
//   * lanes <-> vertices, waves <-> channels: for a channel the 64 lanes read the same [dim x dim]
//     plane (L1/L2 resident), results go through an LDS tile so that the [B,V,960] output is written
//     in contiguous 256-byte runs;
//   * backward, vertices: the output gradient comes in through the same LDS transpose; every (vertex,
//     channel) accumulates d/d(pixel x,y), chained through clamp, the perspective divide and the camera
//     matrix to grad_verts in closed form;
//   * backward, maps: inverted into a gather -- a binning kernel builds texel -> (vertex, weight) lists, then
//     a wave owns one texel x 64 channels and sums its list with coalesced reads (no float atomics).
// The reference's interpolation quirk is kept: weights are (ceil - x, x - floor), so a coordinate that
// is exactly integral (incl. clamped ones) gets all-zero weights (utils.py:346-350, 372-379).
#include "geom_common.h"

namespace {

constexpr int PL_THREADS = 256;
constexpr int PL_WAVES = PL_THREADS / GEOM_WAVE;
constexpr int PL_VERTS = GEOM_WAVE; // vertices per workgroup
constexpr int PL_MAX_CHUNKS = 32;   // grid.z of the pooling launches (more 64-channel chunks than this: a workgroup walks several)
// reference constants (utils.py:321, 329-335)
constexpr float PL_SCALE = 0.57f, PL_FOCAL = 248.f, PL_HALF = 224.f / 2.0f, PL_NORM = 223.f;

struct PoolArgs {
    const float *verts, *cam_mat, *cam_pos;
    const float *blocks[GEOM_POOL_MAX_LEVELS];
    float *grad_blocks[GEOM_POOL_MAX_LEVELS];
    int channels[GEOM_POOL_MAX_LEVELS], dims[GEOM_POOL_MAX_LEVELS];
    int levels, b, nv, ctot;
    int chunks; // 64-channel chunks of all maps together
    int ld;     // floats between two vertex rows of the pooled features / of their gradient (>= ctot: a column slice of a wider buffer)
};

struct Projection {
    float X, Y, Z, xs, ys;
};

__device__ __forceinline__ Projection project(const PoolArgs &a, int mesh, int v)
{
    const float *p = a.verts + ((size_t)mesh * a.nv + v) * 3;
    const float *M = a.cam_mat + (size_t)mesh * 9, *cp = a.cam_pos + (size_t)mesh * 3;
    const float ax = p[0] * PL_SCALE - cp[0], ay = p[1] * PL_SCALE - cp[1], az = p[2] * PL_SCALE - cp[2];
    Projection r;
    r.X = (ax * M[0] + ay * M[1]) + az * M[2];
    r.Y = (ax * M[3] + ay * M[4]) + az * M[5];
    r.Z = (ax * M[6] + ay * M[7]) + az * M[8];
    const float h = (-r.Y) / (-r.Z) * PL_FOCAL + PL_HALF;
    const float w = r.X / (-r.Z) * PL_FOCAL + PL_HALF;
    r.xs = h / PL_NORM;
    r.ys = w / PL_NORM;
    return r;
}

struct Texel {
    int i11, i12, i21, i22;
    float A, B, G, H; // x2-x, x-x1, y2-y, y-y1
    bool in_x, in_y;  // clamp passes the gradient
    // the four texels as TWO 8-byte reads: (x1, yb), (x1, yb + 1) and (x2, yb), (x2, yb + 1) are neighbours in the [dim x dim]
    // plane; yb = y1 except in the last column (there y1 = y2 = dim - 1 and the pair starts one texel earlier)
    int p1, p2;        // index of the pair in row x1 / x2
    bool y1_hi, y2_hi; // y1 / y2 is the pair's second element
};

__device__ __forceinline__ Texel texel(float xs, float ys, int dim)
{
    const float rx = xs * dim, ry = ys * dim, hi = (float)(dim - 1);
    const float cx = fminf(fmaxf(rx, 0.f), hi), cy = fminf(fmaxf(ry, 0.f), hi);
    const float x1 = floorf(cx), x2 = ceilf(cx), y1 = floorf(cy), y2 = ceilf(cy);
    Texel t;
    t.A = x2 - cx, t.B = cx - x1, t.G = y2 - cy, t.H = cy - y1;
    const int ix1 = (int)x1, ix2 = (int)x2, iy1 = (int)y1, iy2 = (int)y2;
    t.i11 = ix1 * dim + iy1, t.i12 = ix1 * dim + iy2, t.i21 = ix2 * dim + iy1, t.i22 = ix2 * dim + iy2;
    t.in_x = rx >= 0.f && rx <= hi;
    t.in_y = ry >= 0.f && ry <= hi;
    const int yb = min(iy1, max(dim - 2, 0));
    t.p1 = ix1 * dim + yb, t.p2 = ix2 * dim + yb;
    t.y1_hi = iy1 != yb, t.y2_hi = iy2 != yb;
    return t;
}

typedef unsigned pl_u32x2 __attribute__((ext_vector_type(2)));

// a mesh's planes of one map as a buffer (the pair reads are 4-byte aligned only: buffer loads take that)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float *blk, int C, int texels)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(blk), 0, (int)((size_t)C * texels * 4), 0x00020000);
}

struct Quad {
    float c11, c12, c21, c22;
};

// the four texels of channel `ch` of a map of 2 x 2 or more: two 8-byte reads
__device__ __forceinline__ Quad quad_of(__amdgpu_buffer_rsrc_t r, const Texel &t, int ch, int texels)
{
    Quad q;
    const unsigned base = (unsigned)ch * (unsigned)texels;
    const pl_u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, (base + (unsigned)t.p1) * 4u, 0, 0);
    const pl_u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, (base + (unsigned)t.p2) * 4u, 0, 0);
    q.c11 = __uint_as_float(t.y1_hi ? a.y : a.x), q.c12 = __uint_as_float(t.y2_hi ? a.y : a.x);
    q.c21 = __uint_as_float(t.y1_hi ? b.y : b.x), q.c22 = __uint_as_float(t.y2_hi ? b.y : b.x);
    return q;
}

// which 64-channel chunk of which map a workgroup owns (grid.z runs over the chunks of all maps in order): 15 chunks for the
// reference's four VGG maps -- 8 vertex tiles x 16 meshes alone are 128 workgroups walking 960 channels each (100 us
// forward, 186 us backward at the reference's training shape); with the chunks side by side the launch fills the chip
struct Chunk {
    int level, cc, off; // map, first channel inside it, first output column of the map
};

__device__ __forceinline__ Chunk chunk_of(const PoolArgs &a, int z)
{
    Chunk c{0, 0, 0};
    while (c.level + 1 < a.levels && z >= (a.channels[c.level] + GEOM_WAVE - 1) / GEOM_WAVE) {
        z -= (a.channels[c.level] + GEOM_WAVE - 1) / GEOM_WAVE;
        c.off += a.channels[c.level];
        ++c.level;
    }
    c.cc = z * GEOM_WAVE;
    return c;
}

// The four texels of a wave's channels c = wave, wave + 4, ... of one chunk, EIGHT channels' reads in flight.  With lanes <->
// vertices every read is a gather of 64 addresses, and the loop over a wave's 16 channels was a chain of 16 dependent round
// trips (issue two reads, wait, use them): 23 us per call at the training shape whatever the traffic -- XCD-aware mapping of
// the work moved it by 1 us.  (Also built: the 64 planes of a 14 x 14 / 7 x 7 chunk copied to LDS and looked up there -- 47 us:
// a workgroup serves only 64 vertices per staged chunk, the copy's round trip and two barriers are not amortised.)
constexpr int PL_INFLIGHT = 8;

template <class Body>
__device__ __forceinline__ void chunk_quads(const PoolArgs &a, const Chunk &ck, int mesh, const Texel &t, int nch, Body body)
{
    const int dim = a.dims[ck.level], C = a.channels[ck.level], texels = dim * dim;
    const __amdgpu_buffer_rsrc_t r_blk = plane_rsrc(a.blocks[ck.level] + (size_t)mesh * C * texels, C, texels);
    const int wave = threadIdx.x >> 6;
    if (texels == 1) { // a 1 x 1 map: its one value four times (the weights are zero: the coordinate is integral)
        for (int c = wave; c < nch; c += PL_WAVES) {
            const float x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_blk, (unsigned)(ck.cc + c) * 4u, 0, 0));
            body(c, Quad{x, x, x, x});
        }
        return;
    }
    for (int c = wave; c < nch; c += PL_WAVES * PL_INFLIGHT) {
        Quad q[PL_INFLIGHT];
#pragma unroll
        for (int j = 0; j < PL_INFLIGHT; ++j) {
#ifdef PL_PROBE_NO_LOADS
            q[j] = Quad{t.A, t.B, t.G, (float)(c + j)};
#else
            q[j] = quad_of(r_blk, t, ck.cc + min(c + PL_WAVES * j, nch - 1), texels);
#endif
        }
#pragma unroll
        for (int j = 0; j < PL_INFLIGHT; ++j)
            if (c + PL_WAVES * j < nch) body(c + PL_WAVES * j, q[j]);
    }
}

// Which (vertex tile, mesh, chunk slice) a workgroup of the 1-D pooling launches owns.  Workgroups go to the eight XCDs round
// robin (XCD = blockIdx.x % 8; the grid is a multiple of 8): an XCD is given CONSECUTIVE work items, ordered tile-fastest, so the
// vertex tiles of one (mesh, chunk) -- which all read the same 64 planes -- run on one XCD, one after the other, and the planes
// come out of HBM once.  (As a 3-D grid with the tiles in x, the eight tiles of a mesh went to eight XCDs and every L2 fetched
// every map: 29 us forward at the training shape, 24 with paired reads, against ~10 for the bytes.)
struct PoolWork {
    int v0, mesh, z;
    bool live;
};

__device__ __forceinline__ PoolWork pool_work(const PoolArgs &a, int zc, int bid, int nblocks)
{
    const int tiles = (a.nv + PL_VERTS - 1) / PL_VERTS;
    const int w = (bid & 7) * (nblocks >> 3) + (bid >> 3);
    PoolWork k;
    k.live = w < tiles * a.b * zc;
    const int r = w / tiles;
    k.v0 = (w - r * tiles) * PL_VERTS, k.mesh = r / zc, k.z = r - k.mesh * zc; // (chunks inside a mesh: every XCD gets maps of every size)
    return k;
}

static inline unsigned pool_grid(int nv, int b, int zc)
{
    const size_t n = (size_t)((nv + PL_VERTS - 1) / PL_VERTS) * b * zc;
    return (unsigned)((n + 7) / 8 * 8);
}

// What the caller is going to concatenate IN FRONT of the pooled features (GEOMetrics.py:123,128: the previous features;
// models.py:241: the coordinates), copied into the free columns of the wide buffer by workgroups BESIDE the pooling ones -- the
// narrow copies were launches of their own (and 5.9 MB each for the previous features).
constexpr int PL_MAX_FRONTS = 2;
struct PoolFronts {
    const float *src[PL_MAX_FRONTS]; // [b * nv, width] contiguous
    int width[PL_MAX_FRONTS], col[PL_MAX_FRONTS], first_block[PL_MAX_FRONTS + 1];
    int count;
    float *dst;     // the wide buffer (row pitch a.ld)
    int work_blocks; // the pooling workgroups in front of the copying ones
};

__global__ __launch_bounds__(PL_THREADS) void pool_fwd_kernel(PoolArgs a, float *out, int zc, PoolFronts fr)
{
    __shared__ float tile[PL_VERTS][GEOM_WAVE + 1];
    if ((int)blockIdx.x >= fr.work_blocks) {
        const int fb = (int)blockIdx.x - fr.work_blocks;
        const int f = (fr.count > 1 && fb >= fr.first_block[1]) ? 1 : 0;
        const int64_t total = (int64_t)a.b * a.nv * fr.width[f];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t idx = ((int64_t)(fb - fr.first_block[f]) * 4 + q) * PL_THREADS + threadIdx.x;
            if (idx < total) {
                const int64_t row = idx / fr.width[f];
                fr.dst[row * a.ld + fr.col[f] + (int)(idx - row * fr.width[f])] = fr.src[f][idx];
            }
        }
        return;
    }
    const PoolWork k = pool_work(a, zc, blockIdx.x, fr.work_blocks);
    if (!k.live) return;
    const int mesh = k.mesh, v0 = k.v0;
    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int v = min(v0 + lane, a.nv - 1);
    const Projection pr = project(a, mesh, v);
    for (int z = k.z; z < a.chunks; z += zc) { // (one chunk per workgroup up to PL_MAX_CHUNKS chunks)
        const Chunk ck = chunk_of(a, z);
        const int dim = a.dims[ck.level], C = a.channels[ck.level], cc = ck.cc;
        const Texel t = texel(pr.xs, pr.ys, dim);
        const int nch = min(GEOM_WAVE, C - cc);
        chunk_quads(a, ck, mesh, t, nch, [&](int c, const Quad &q) {
            const float s1 = (t.A * q.c11) * t.G, s2 = (t.H * q.c12) * t.A;
            const float s3 = (t.G * q.c21) * t.B, s4 = (t.B * q.c22) * t.H;
            tile[lane][c] = ((s1 + s2) + s3) + s4;
        });
        __syncthreads();
        // a wave writes one vertex's 64 channels per round (lanes <-> channels: 256 contiguous bytes, no index arithmetic)
        float *o = out + ((size_t)mesh * a.nv + v0) * a.ld + ck.off + cc + lane;
        for (int vv = threadIdx.x >> 6; vv < min(PL_VERTS, a.nv - v0); vv += PL_WAVES) {
#ifdef PL_PROBE_NO_STORE
            if (tile[vv][lane] == 1.2345f)
#endif
            if (lane < nch) o[(size_t)vv * a.ld] = tile[vv][lane];
        }
        __syncthreads();
    }
}

// d loss / d verts: vertex-tiled like the forward (the gradient w.r.t. a vertex sums over all channels).  A workgroup owns
// one 64-channel chunk and leaves its share of d loss / d (xs, ys) per vertex in `partial` [chunk][mesh][vertex][2];
// pool_bwd_verts_finish_body adds the chunks up in chunk order (fixed: bit-reproducible) and chains the sum through the
// clamp, the perspective divide and the camera matrix.
struct VertsLds {
    float tile[PL_VERTS][GEOM_WAVE + 1];
    float part[PL_WAVES][2][PL_VERTS];
};

__device__ __forceinline__ void pool_bwd_verts_body(const PoolArgs &a, const float *grad_out, float *partial, int zc, int bid,
                                                    int nblocks, VertsLds &L)
{
    auto &tile = L.tile;
    auto &part = L.part;
    const PoolWork k = pool_work(a, zc, bid, nblocks);
    if (!k.live) return;
    const int mesh = k.mesh, v0 = k.v0;
    const int lane = threadIdx.x & (GEOM_WAVE - 1), wave = threadIdx.x >> 6;
    const bool live = v0 + lane < a.nv;
    const int v = min(v0 + lane, a.nv - 1);
    const Projection pr = project(a, mesh, v);
    float ax = 0.f, ay = 0.f; // this wave's share of d loss / d (xs, ys) of the lane's vertex over the workgroup's chunks
    for (int z = k.z; z < a.chunks; z += zc) {
        const Chunk ck = chunk_of(a, z);
        const int dim = a.dims[ck.level], C = a.channels[ck.level], cc = ck.cc;
        const Texel t = texel(pr.xs, pr.ys, dim);
        const int nch = min(GEOM_WAVE, C - cc);
        const float *gi = grad_out + ((size_t)mesh * a.nv + v0) * a.ld + ck.off + cc + (lane < nch ? lane : 0);
        for (int vv = wave; vv < PL_VERTS; vv += PL_WAVES) // (a wave reads one vertex's 64 channels per round)
            tile[vv][lane] = (v0 + vv < a.nv && lane < nch) ? gi[(size_t)vv * a.ld] : 0.f;
        __syncthreads();
        float gx = 0.f, gy = 0.f;
        chunk_quads(a, ck, mesh, t, nch, [&](int c, const Quad &q) { // (a padding lane's sums are dropped below)
            const float g = tile[lane][c];
            const float c11 = q.c11, c12 = q.c12, c21 = q.c21, c22 = q.c22;
            gx += g * (((-c11 * t.G) - (t.H * c12)) + ((t.G * c21) + (c22 * t.H)));
            gy += g * (((-t.A * c11) + (c12 * t.A)) + ((-c21 * t.B) + (t.B * c22)));
        });
        if (t.in_x) ax += gx * dim; // the clamp passes the gradient only inside the map
        if (t.in_y) ay += gy * dim;
        __syncthreads();
    }
    part[wave][0][lane] = ax;
    part[wave][1][lane] = ay;
    __syncthreads();
    if (wave == 0 && live) {
        float sx = 0.f, sy = 0.f;
        for (int w = 0; w < PL_WAVES; ++w) {
            sx += part[w][0][lane];
            sy += part[w][1][lane];
        }
        float *o = partial + (((size_t)k.z * a.b + mesh) * a.nv + v) * 2;
        o[0] = sx, o[1] = sy;
    }
}

__device__ __forceinline__ void pool_bwd_verts_finish_body(const PoolArgs &a, const float *partial, int chunks, float *grad_verts,
                                                           int mesh, int v)
{
    if (v >= a.nv) return;
    const Projection pr = project(a, mesh, v);
    float sx = 0.f, sy = 0.f;
    for (int z = 0; z < chunks; ++z) {
        const float *p = partial + (((size_t)z * a.b + mesh) * a.nv + v) * 2;
        sx += p[0], sy += p[1];
    }
    // xs = h/223, ys = w/223; h = (-Y)/(-Z)*F + 112, w = X/(-Z)*F + 112
    const float gh = sx / PL_NORM, gw = sy / PL_NORM;
    const float d = -pr.Z;
    const float gX = gw * (PL_FOCAL / d);
    const float gY = -gh * (PL_FOCAL / d);
    const float gZ = (gh * ((-pr.Y) * PL_FOCAL) + gw * (pr.X * PL_FOCAL)) / (d * d); // d(.)/dd * dd/dZ, dd/dZ = -1 twice
    const float *M = a.cam_mat + (size_t)mesh * 9;
    float *gv = grad_verts + ((size_t)mesh * a.nv + v) * 3;
    gv[0] = PL_SCALE * ((gX * M[0] + gY * M[3]) + gZ * M[6]);
    gv[1] = PL_SCALE * ((gX * M[1] + gY * M[4]) + gZ * M[7]);
    gv[2] = PL_SCALE * ((gX * M[2] + gY * M[5]) + gZ * M[8]);
}

// d loss / d maps as a GATHER (no floating-point atomics anywhere).
//
// Scattering a vertex's four texel contributions is hostile to this hardware: the maps are as small as
// 7x7, so global fp32 atomics from 2562 vertices pile up on a few addresses (4.4 ms per call measured),
// LDS ds_add_f32 costs ~1700 cycles per wave instruction whatever the addresses (2 ms), and wave-private
// LDS planes with plain adds are a serial walk over the vertices (0.8 ms).  So the scatter is inverted:
//   1. the binning role (pool_bin_body), one workgroup per (mesh, level): counts the contributions per texel (INTEGER LDS
//      atomics), scans, and fills texel -> (vertex, weight) lists in the workspace;
//   2. the gather role (pg_run): a workgroup owns a run of texels x up to 256 channels and sums w * grad_out[b, v, c] over
//      the texels' lists -- every read is a contiguous run of the gradient row, every output is written exactly once
//      (nothing to zero-initialise).
// The order inside a list follows the atomic cursor, so the fp32 summation order can differ between runs
// (as with torch's own index_add backward); contributions with zero weight are dropped.
constexpr int BIN_THREADS = 256;
constexpr int BIN_MAX_TEXELS = 4096; // dim <= 64

struct BinSpace {
    int *offsets;   // per (mesh, level): dim*dim + 1 ints, level-major inside a mesh
    int *ent_v;     // per (mesh, level): 4*nv vertex ids
    float *ent_w;   // per (mesh, level): 4*nv weights
    int off_stride; // ints per mesh in `offsets`
    int level_off[GEOM_POOL_MAX_LEVELS]; // start of level l inside a mesh's offsets block
    float *vert_partial; // [PL_MAX_CHUNKS][b][nv][2]
};

struct BinLds {
    int counts[BIN_MAX_TEXELS + 1];
    int wave_sum[BIN_THREADS / GEOM_WAVE];
};

__device__ __forceinline__ void pool_bin_body(const PoolArgs &a, const BinSpace &ws, int l, int mesh, BinLds &L)
{
    auto &counts = L.counts;
    auto &wave_sum = L.wave_sum;
    const int dim = a.dims[l], texels = dim * dim;
    for (int i = threadIdx.x; i <= texels; i += BIN_THREADS) counts[i] = 0;
    __syncthreads();
    for (int v = threadIdx.x; v < a.nv; v += BIN_THREADS) {
        const Projection pr = project(a, mesh, v);
        const Texel t = texel(pr.xs, pr.ys, dim);
        if (t.G * t.A != 0.f) atomicAdd(&counts[t.i11], 1);
        if (t.A * t.H != 0.f) atomicAdd(&counts[t.i12], 1);
        if (t.B * t.G != 0.f) atomicAdd(&counts[t.i21], 1);
        if (t.H * t.B != 0.f) atomicAdd(&counts[t.i22], 1);
    }
    __syncthreads();
    // exclusive scan of counts[0..texels): thread owns a contiguous run, wave shuffle scan, wave totals
    const int per = (texels + BIN_THREADS - 1) / BIN_THREADS;
    const int i0 = threadIdx.x * per;
    int run = 0;
    for (int i = i0; i < min(i0 + per, texels); ++i) run += counts[i];
    const int lane = threadIdx.x & (GEOM_WAVE - 1), wave = threadIdx.x >> 6;
    int incl = run;
    for (int o = 1; o < GEOM_WAVE; o <<= 1) {
        const int t = __shfl_up(incl, o, GEOM_WAVE);
        if (lane >= o) incl += t;
    }
    if (lane == GEOM_WAVE - 1) wave_sum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_sum[w];
    int excl = base + incl - run;
    int *offs = ws.offsets + (size_t)mesh * ws.off_stride + ws.level_off[l];
    for (int i = i0; i < min(i0 + per, texels); ++i) {
        const int c = counts[i];
        counts[i] = excl; // becomes the fill cursor
        offs[i] = excl;
        excl += c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < BIN_THREADS / GEOM_WAVE; ++w) total += wave_sum[w];
        offs[texels] = total;
    }
    int *ev = ws.ent_v + ((size_t)mesh * a.levels + l) * 4 * a.nv;
    float *ew = ws.ent_w + ((size_t)mesh * a.levels + l) * 4 * a.nv;
    for (int v = threadIdx.x; v < a.nv; v += BIN_THREADS) {
        const Projection pr = project(a, mesh, v);
        const Texel t = texel(pr.xs, pr.ys, dim);
        const float w11 = t.G * t.A, w12 = t.A * t.H, w21 = t.B * t.G, w22 = t.H * t.B;
        if (w11 != 0.f) { const int p = atomicAdd(&counts[t.i11], 1); ev[p] = v; ew[p] = w11; }
        if (w12 != 0.f) { const int p = atomicAdd(&counts[t.i12], 1); ev[p] = v; ew[p] = w12; }
        if (w21 != 0.f) { const int p = atomicAdd(&counts[t.i21], 1); ev[p] = v; ew[p] = w21; }
        if (w22 != 0.f) { const int p = atomicAdd(&counts[t.i22], 1); ev[p] = v; ew[p] = w22; }
    }
}

// The backward pass is TWO launches of two roles each (the four bodies were launches of their own until round 6: at the
// training shape the 64-workgroup binning pass and the 32-workgroup finish pass each held the whole chip for 6-8 us):
//   1. pool_bwd_lists_kernel: the texel lists (one workgroup per mesh and map) BESIDE the vertex tiles' per-chunk sums of
//      d loss / d (xs, ys) -- neither reads what the other writes;
//   2. pool_bwd_grads_kernel: the map gradient from the lists BESIDE the vertex gradient from the per-chunk sums.
static_assert(BIN_THREADS == PL_THREADS, "the two roles of a launch share its workgroup size");

__global__ __launch_bounds__(PL_THREADS) void pool_bwd_lists_kernel(PoolArgs a, BinSpace ws, const float *grad_out, int zc,
                                                                    int bin_blocks, int verts_blocks)
{
    __shared__ union Lds {
        BinLds bin;
        VertsLds verts;
        __device__ Lds() {}
    } lds;
    if ((int)blockIdx.x < bin_blocks) pool_bin_body(a, ws, (int)blockIdx.x % a.levels, (int)blockIdx.x / a.levels, lds.bin);
    else pool_bwd_verts_body(a, grad_out, ws.vert_partial, zc, (int)blockIdx.x - bin_blocks, verts_blocks, lds.verts);
}

// What this kernel has to avoid is not traffic but INSTRUCTIONS and TAILS.  The lists are short and skewed: at the training
// shape 0.6 entries per texel at 56 x 56 and 40 at 7 x 7, two thirds of the texels empty, the fullest 7 x 7 texel 135 entries
// (tools/time_pool_gather.py).  Round 5's form gave every (texel, 64-channel chunk) its own wave -- 94 080 waves per 16 meshes,
// each mostly index arithmetic: 59 us per call, 55 us with the gradient-row loads removed (tools/probe/pg_variants.sh); with
// waves <-> chunks instead, the fullest texel's wave walked 135 rows x 2 chunks alone (42 us).  Here
//   * a workgroup owns a RUN of T consecutive texels of one map (gather_plan: 64 texels at 56 x 56 down to one at 7 x 7: about
//     40 entries) x up to 256 channels: a lane holds VEC = 1 / 2 / 4 consecutive channels, a gradient row arrives as ONE
//     load of up to 16 bytes per lane;
//   * the lists of a run are ONE contiguous range of the workspace (they are texel-major): the four waves take a quarter
//     of the ENTRIES each -- whatever the texels --, up to 64 (vertex, weight) pairs per load, eight rows in flight;
//   * a wave's sums for the texels strictly inside its quarter go to the LDS tile as they are; its first and its last texel
//     may continue in a neighbour's quarter: those two partial sums are added wave after wave (list order: the result does
//     not depend on how the entries were split);
//   * every per-map constant comes precomputed in the launch arguments: no integer division anywhere;
//   * the tile leaves as runs of T floats per channel (256 bytes at 56 x 56).
constexpr int PG_TILE = 4096; // floats of a run's sums: T texels x 64 VEC channels (+ VEC per texel row of padding)

struct GatherPlan {
    int first_task[GEOM_POOL_MAX_LEVELS + 1]; // tasks (= runs) of level l: [first_task[l], first_task[l + 1])
    int log_t[GEOM_POOL_MAX_LEVELS];          // T = 1 << log_t texels per run
    int vec[GEOM_POOL_MAX_LEVELS];            // channels per lane
    int parts[GEOM_POOL_MAX_LEVELS];          // 64 * vec-channel slices of the map (grid.z walks them)
    int col0[GEOM_POOL_MAX_LEVELS];           // first column of the map inside the pooled features
    int max_parts;
};

static void gather_plan(const PoolArgs &a, GatherPlan &p)
{
    int task = 0, col = 0;
    p.max_parts = 1;
    for (int l = 0; l < a.levels; ++l) {
        const int texels = a.dims[l] * a.dims[l], C = a.channels[l];
        // (rows of the pitched gradient are only 4-byte aligned: any vector width is legal for the buffer loads; a lane's
        // channels must not straddle the map's last channel)
        const int vec = (C >= 256 && C % 4 == 0) ? 4 : (C >= 128 && C % 2 == 0) ? 2 : 1;
        const int W = GEOM_WAVE * vec;
        int log_t = 0;
        while ((2 << log_t) * W <= PG_TILE && (2 << log_t) * 48 <= texels) ++log_t;
        p.first_task[l] = task, p.log_t[l] = log_t, p.vec[l] = vec, p.parts[l] = (C + W - 1) / W, p.col0[l] = col;
        if (p.parts[l] > p.max_parts) p.max_parts = p.parts[l];
        task += (texels + (1 << log_t) - 1) >> log_t;
        col += C;
    }
    p.first_task[a.levels] = task;
}

typedef unsigned pg_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pg_u32x2 __attribute__((ext_vector_type(2)));

template <int VEC> struct PgRow {
    float x[VEC];
};

template <int VEC> __device__ __forceinline__ PgRow<VEC> pg_load(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    PgRow<VEC> o;
    if constexpr (VEC == 4) {
        const pg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        o.x[0] = __uint_as_float(v.x), o.x[1] = __uint_as_float(v.y), o.x[2] = __uint_as_float(v.z), o.x[3] = __uint_as_float(v.w);
    } else if constexpr (VEC == 2) {
        const pg_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        o.x[0] = __uint_as_float(v.x), o.x[1] = __uint_as_float(v.y);
    } else {
        o.x[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    }
    return o;
}

template <int VEC>
__device__ __forceinline__ void pg_run(const PoolArgs &a, const BinSpace &ws, const GatherPlan &p, const float *grad_out, int l,
                                       int run, int part, float *tile)
{
    constexpr int W = GEOM_WAVE * VEC, ST = W + VEC; // channels of the slice, floats per texel row of the tile
    const int lane = threadIdx.x & (GEOM_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mesh = blockIdx.y;
    const int dim = a.dims[l], C = a.channels[l], texels = dim * dim;
    const int log_t = p.log_t[l], T = 1 << log_t;
    const int tx0 = run << log_t, nt = min(T, texels - tx0); // texels of this run
    const int *offs = ws.offsets + (size_t)mesh * ws.off_stride + ws.level_off[l] + tx0;
    const int *ev = ws.ent_v + ((size_t)mesh * a.levels + l) * 4 * a.nv;
    const float *ew = ws.ent_w + ((size_t)mesh * a.levels + l) * 4 * a.nv;
    // where the list of texel `lane` ends (lanes past the run: the end of the run's range)
    const int ends = offs[min(lane + 1, nt)];
    const int E0 = offs[0], E1 = __builtin_amdgcn_readlane(ends, GEOM_WAVE - 1);
    for (int i = threadIdx.x; i < (ST << log_t); i += PL_THREADS) tile[i] = 0.f; // (texels nothing projects into stay zero)
    __syncthreads();
    const int n = E1 - E0;
    const int ea = E0 + ((n * wave) >> 2), ez = E0 + ((n * (wave + 1)) >> 2); // this wave's quarter of the entries
    const int c = part * W + lane * VEC;                                          // the lane's first channel
    const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(grad_out + (size_t)mesh * a.nv * a.ld), 0, (int)((((size_t)a.nv - 1) * a.ld + a.ctot) * 4), 0x00020000);
    const unsigned col_off = c < C ? (unsigned)(p.col0[l] + c) * 4u : 0x80000000u; // (out of range: the load returns zeros)
    // the texel the quarter starts in = the number of lists that end at or before its first entry
    const int t_first = __builtin_popcountll(__ballot(lane < nt && ends <= ea));
    int cur = t_first, next = __builtin_amdgcn_readlane(ends, min(cur, GEOM_WAVE - 1));
    PgRow<VEC> acc, first;
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc.x[q] = 0.f, first.x[q] = 0.f;
    for (int eb = ea; eb < ez; eb += GEOM_WAVE) {
        const int nb = min(GEOM_WAVE, ez - eb);
        const int my = eb + min(lane, nb - 1);
        const int vr = ev[my];
        const float wr = ew[my];
        for (int k0 = 0; k0 < nb; k0 += 8) {
            PgRow<VEC> gv[8];
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = min(k0 + j, nb - 1);
                const int v = __builtin_amdgcn_readlane(vr, k);
                wv[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wr), k));
#ifdef PG_PROBE_NO_ROWS
                for (int q = 0; q < VEC; ++q) gv[j].x[q] = (float)v;
#else
                gv[j] = pg_load<VEC>(r_g, col_off + (unsigned)v * (unsigned)a.ld * 4u);
#endif
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + j < nb) {
                    const int e = eb + k0 + j;
                    while (e >= next) { // (wave-uniform) the list of `cur` is done
                        if (cur == t_first) {
                            first = acc;
                        } else {
#pragma unroll
                            for (int q = 0; q < VEC; ++q) tile[cur * ST + lane * VEC + q] = acc.x[q];
                        }
#pragma unroll
                        for (int q = 0; q < VEC; ++q) acc.x[q] = 0.f;
                        ++cur;
                        next = __builtin_amdgcn_readlane(ends, cur);
                    }
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc.x[q] += gv[j].x[q] * wv[j];
                }
            }
        }
    }
    // the quarter's first and last texel: added in wave (= list) order
    for (int w = 0; w < PL_WAVES; ++w) {
        if (w == wave && ea < ez) {
            if (cur != t_first) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) tile[t_first * ST + lane * VEC + q] += first.x[q];
            }
#pragma unroll
            for (int q = 0; q < VEC; ++q) tile[cur * ST + lane * VEC + q] += acc.x[q];
        }
        __syncthreads();
    }
    // out: per channel a run of nt consecutive texels
    float *out = a.grad_blocks[l] + ((size_t)mesh * C + part * W) * texels + tx0;
    const int nch = min(W, C - part * W);
    for (int i = threadIdx.x; i < (W << log_t); i += PL_THREADS) {
        const int t = i & (T - 1), ch = i >> log_t;
#ifdef PG_PROBE_NO_STORE
        if (tile[t * ST + ch] == 1.2345f)
#endif
        if (t < nt && ch < nch) out[(size_t)ch * texels + t] = tile[t * ST + ch];
    }
}

// finish_z >= 0: the workgroups of that grid.z slice chain the vertex tiles' sums to grad_verts (256 vertices each); gather_tasks
// = the runs of all maps (0: no map wants a gradient)
__global__ __launch_bounds__(PL_THREADS) void pool_bwd_grads_kernel(PoolArgs a, BinSpace ws, GatherPlan p, const float *grad_out,
                                                                    int gather_tasks, int finish_z, int finish_chunks,
                                                                    float *grad_verts)
{
    __shared__ __attribute__((aligned(16))) float tile[PG_TILE + GEOM_WAVE];
    if ((int)blockIdx.z == finish_z) {
        pool_bwd_verts_finish_body(a, ws.vert_partial, finish_chunks, grad_verts, blockIdx.y, (int)blockIdx.x * PL_THREADS + (int)threadIdx.x);
        return;
    }
    // workgroups go to the eight XCDs round robin (gridDim.x is a multiple of 8: XCD = blockIdx.x % 8): give an XCD CONSECUTIVE
    // runs -- the 2 x 2 texels a vertex touches then meet in one L2 (its gradient row is fetched from HBM once, not per XCD),
    // and so do the short runs of a small map that share a 128-byte line of the output
    const int task = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (task >= gather_tasks) return;
    int l = 0;
    while (l + 1 < a.levels && task >= p.first_task[l + 1]) ++l;
    const int part = blockIdx.z;
    if (!a.grad_blocks[l] || part >= p.parts[l]) return; // workgroup-uniform
#ifdef PG_PROBE_BARE
    return;
#endif
    const int run = task - p.first_task[l];
    switch (p.vec[l]) {
    case 4: pg_run<4>(a, ws, p, grad_out, l, run, part, tile); break;
    case 2: pg_run<2>(a, ws, p, grad_out, l, run, part, tile); break;
    default: pg_run<1>(a, ws, p, grad_out, l, run, part, tile); break;
    }
}

int fill_args(PoolArgs &a, int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos, int levels,
              const float *const *blocks, const int *channels, const int *dims)
{
    if (b < 0 || nv < 0 || levels < 0 || levels > GEOM_POOL_MAX_LEVELS) return GEOM_EINVAL;
    if (!verts || !cam_mat || !cam_pos || (levels > 0 && (!blocks || !channels || !dims))) return GEOM_EINVAL;
    if (b > 65535 || (size_t)((nv + PL_VERTS - 1) / PL_VERTS) * b * PL_MAX_CHUNKS > (size_t)INT_MAX - 8) return GEOM_ETOOBIG;
    a.verts = verts, a.cam_mat = cam_mat, a.cam_pos = cam_pos, a.levels = levels, a.b = b, a.nv = nv, a.ctot = 0, a.chunks = 0;
    for (int l = 0; l < levels; ++l) {
        if (!blocks[l] || channels[l] <= 0 || dims[l] <= 0) return GEOM_EINVAL;
        if ((size_t)channels[l] * dims[l] * dims[l] * 4 > (size_t)INT_MAX) return GEOM_ETOOBIG; // (a mesh's planes as one buffer)
        a.blocks[l] = blocks[l], a.grad_blocks[l] = nullptr, a.channels[l] = channels[l], a.dims[l] = dims[l];
        a.ctot += channels[l];
        a.chunks += (channels[l] + GEOM_WAVE - 1) / GEOM_WAVE;
    }
    a.ld = a.ctot;
    return 0;
}

} // namespace

static int pool_fwd_launch(PoolArgs &a, int b, int nv, float *out, int64_t out_ld, int n_fronts, const float *const *fronts,
                           const int *widths, const int *cols, float *buf, void *stream)
{
    if (!out) return GEOM_EINVAL;
    if (out_ld) {
        if (out_ld < a.ctot || out_ld > INT_MAX) return GEOM_EINVAL;
        a.ld = (int)out_ld;
    }
    PoolFronts fr{};
    const int zc = a.chunks < PL_MAX_CHUNKS ? a.chunks : PL_MAX_CHUNKS;
    fr.work_blocks = (int)pool_grid(nv, b, zc);
    if (n_fronts < 0 || n_fronts > PL_MAX_FRONTS || (n_fronts && (!fronts || !widths || !cols || !buf))) return GEOM_EINVAL;
    int extra = 0;
    for (int f = 0; f < n_fronts; ++f) {
        if (!fronts[f] || widths[f] <= 0 || cols[f] < 0 || cols[f] + widths[f] > a.ld) return GEOM_EINVAL;
        fr.src[f] = fronts[f], fr.width[f] = widths[f], fr.col[f] = cols[f], fr.first_block[f] = extra;
        const int64_t blocks = ((int64_t)b * nv * widths[f] + 4 * PL_THREADS - 1) / (4 * PL_THREADS);
        if (blocks > (1 << 24)) return GEOM_ETOOBIG;
        extra += (int)blocks;
    }
    fr.first_block[n_fronts] = extra, fr.count = n_fronts, fr.dst = buf;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(fr.work_blocks + extra), dim3(PL_THREADS), 0, static_cast<hipStream_t>(stream), a, out, zc, fr);
    return geom::launch_status();
}

// out_ld: floats between two vertex rows of `out` (0 = the pooled width: a contiguous [b,nv,sum channels] tensor; larger: the
// features are written as a column slice of a wider row-major buffer -- the deformation block's concatenated input)
extern "C" int geom_pool_features_fwd_ld_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                                             int levels, const float *const *blocks, const int *channels, const int *dims,
                                             float *out, int64_t out_ld, void *stream)
{
    PoolArgs a;
    if (int rc = fill_args(a, b, nv, verts, cam_mat, cam_pos, levels, blocks, channels, dims)) return rc;
    if (b == 0 || nv == 0 || levels == 0) return 0;
    return pool_fwd_launch(a, b, nv, out, out_ld, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

// ... and up to two tensors fronts[f] [b, nv, widths[f]] (contiguous) copied into columns [cols[f], cols[f] + widths[f]) of the
// wide buffer `buf` (row pitch out_ld, `out` = buf + the pooled features' first column) by the same launch: what the caller
// concatenates in front of the pooled features (GEOMetrics.py:123,128; models.py:241)
extern "C" int geom_pool_features_fwd_fronts_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                                                 int levels, const float *const *blocks, const int *channels, const int *dims,
                                                 float *out, int64_t out_ld, int n_fronts, const float *const *fronts,
                                                 const int *widths, const int *cols, float *buf, void *stream)
{
    PoolArgs a;
    if (int rc = fill_args(a, b, nv, verts, cam_mat, cam_pos, levels, blocks, channels, dims)) return rc;
    if (b == 0 || nv == 0 || levels == 0) return 0;
    return pool_fwd_launch(a, b, nv, out, out_ld, n_fronts, fronts, widths, cols, buf, stream);
}

extern "C" int geom_pool_features_fwd_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                                          int levels, const float *const *blocks, const int *channels, const int *dims,
                                          float *out, void *stream)
{
    return geom_pool_features_fwd_ld_f32(b, nv, verts, cam_mat, cam_pos, levels, blocks, channels, dims, out, 0, stream);
}

static size_t pool_ws_layout(int b, int nv, int levels, const int *dims, BinSpace *ws, void *base)
{
    int per_mesh = 0;
    for (int l = 0; l < levels; ++l) {
        if (ws) ws->level_off[l] = per_mesh;
        per_mesh += dims[l] * dims[l] + 1;
    }
    const size_t off_bytes = ((size_t)b * per_mesh * sizeof(int) + 15) / 16 * 16;
    const size_t ent = (size_t)b * levels * 4 * nv;
    if (ws) {
        char *p = static_cast<char *>(base);
        ws->offsets = reinterpret_cast<int *>(p);
        ws->ent_v = reinterpret_cast<int *>(p + off_bytes);
        ws->ent_w = reinterpret_cast<float *>(p + off_bytes + ent * sizeof(int));
        ws->off_stride = per_mesh;
    }
    const size_t lists = (off_bytes + ent * (sizeof(int) + sizeof(float)) + 15) / 16 * 16;
    if (ws) ws->vert_partial = reinterpret_cast<float *>(static_cast<char *>(base) + lists);
    // + the per-chunk shares of d loss / d (xs, ys) of every vertex (pool_bwd_verts_body)
    return lists + (size_t)PL_MAX_CHUNKS * b * nv * 2 * sizeof(float);
}

extern "C" size_t geom_pool_features_bwd_workspace_bytes(int b, int nv, int levels, const int *dims)
{
    if (b <= 0 || nv <= 0 || levels <= 0 || levels > GEOM_POOL_MAX_LEVELS || !dims) return 0;
    return pool_ws_layout(b, nv, levels, dims, nullptr, nullptr);
}

extern "C" int geom_pool_features_bwd_ld_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                                             int levels, const float *const *blocks, const int *channels, const int *dims,
                                             const float *grad_out, int64_t grad_ld, float *const *grad_blocks, float *grad_verts,
                                             void *workspace, size_t workspace_bytes, void *stream);
extern "C" int geom_pool_features_bwd_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                                          int levels, const float *const *blocks, const int *channels, const int *dims,
                                          const float *grad_out, float *const *grad_blocks, float *grad_verts,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    return geom_pool_features_bwd_ld_f32(b, nv, verts, cam_mat, cam_pos, levels, blocks, channels, dims, grad_out, 0, grad_blocks,
                                         grad_verts, workspace, workspace_bytes, stream);
}

// grad_ld: floats between two vertex rows of `grad_out` (0 = the pooled width; larger: the gradient is read in place out of a
// column slice of a wider buffer -- the input gradient of the deformation block's first product)
extern "C" int geom_pool_features_bwd_ld_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                                             int levels, const float *const *blocks, const int *channels, const int *dims,
                                             const float *grad_out, int64_t grad_ld, float *const *grad_blocks, float *grad_verts,
                                             void *workspace, size_t workspace_bytes, void *stream)
{
    PoolArgs a;
    if (int rc = fill_args(a, b, nv, verts, cam_mat, cam_pos, levels, blocks, channels, dims)) return rc;
    if (b == 0 || nv == 0 || levels == 0) return 0;
    if (!grad_out) return GEOM_EINVAL;
    if (grad_ld) {
        if (grad_ld < a.ctot || grad_ld > INT_MAX) return GEOM_EINVAL;
        a.ld = (int)grad_ld;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool any_map = false;
    for (int l = 0; l < levels; ++l) {
        a.grad_blocks[l] = grad_blocks ? grad_blocks[l] : nullptr;
        any_map = any_map || a.grad_blocks[l];
        if (dims[l] * dims[l] > BIN_MAX_TEXELS) return GEOM_EUNSUPPORTED;

    }
    if (!any_map && !grad_verts) return 0;
    BinSpace ws;
    const size_t need = pool_ws_layout(b, nv, levels, dims, &ws, workspace);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return GEOM_EINVAL;
    GatherPlan plan;
    gather_plan(a, plan);
    if (plan.max_parts > 65534 || (size_t)nv * a.ld * 4 > (size_t)INT_MAX) return GEOM_ETOOBIG;
    const int zc = a.chunks < PL_MAX_CHUNKS ? a.chunks : PL_MAX_CHUNKS;
    const int bin_blocks = any_map ? levels * b : 0;
    const unsigned verts_blocks = grad_verts ? pool_grid(nv, b, zc) : 0;
    hipLaunchKernelGGL(pool_bwd_lists_kernel, dim3(bin_blocks + verts_blocks), dim3(PL_THREADS), 0, s, a, ws, grad_out, zc, bin_blocks,
                       (int)verts_blocks);
    const int gather_tasks = any_map ? plan.first_task[levels] : 0;
    const int parts = any_map ? plan.max_parts : 0;
    int gx = (gather_tasks + 7) / 8 * 8;
    if (grad_verts && gx < (nv + PL_THREADS - 1) / PL_THREADS) gx = ((nv + PL_THREADS - 1) / PL_THREADS + 7) / 8 * 8;
    hipLaunchKernelGGL(pool_bwd_grads_kernel, dim3(gx, b, parts + (grad_verts ? 1 : 0)), dim3(PL_THREADS), 0, s, a, ws, plan, grad_out,
                       gather_tasks, grad_verts ? parts : -1, zc, grad_verts);
    return geom::launch_status();
}

// ---- camera from (azimuth deg, elevation deg, distance): reference utils.py:286-313 -----------------------------------------
// The reference (and this package's torch mirror of it) forms the camera of every image with ~35 tiny eager launches --
// scale, remainder, sin, cos, stack, two cross products, three normalisations -- three times per training step; here one
// thread per image evaluates the same fp32 expressions in the same order (-ffp-contract=off: no fused multiply-adds).
namespace {
// torch.remainder (the `%` of the reference line): the result takes the divisor's sign
__device__ __forceinline__ float fmodf_floor(float a, float b)
{
    float r = fmodf(a, b);
    if (r != 0.f && ((r < 0.f) != (b < 0.f))) r += b;
    return r;
}
__global__ __launch_bounds__(64) void camera_info_kernel(int b, const float *param, float *cam_mat, float *cam_pos)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= b) return;
    const float pi = 3.14159265358979323846f;
    const float theta = fmodf_floor(pi * param[3 * i] / 180.0f, 360.0f), phi = fmodf_floor(pi * param[3 * i + 1] / 180.0f, 360.0f);
    const float dist = param[3 * i + 2];
    const float cam_y = dist * sinf(phi), flat = dist * cosf(phi);
    const float z[3] = {flat * cosf(theta), cam_y, flat * sinf(theta)};          // camera position = the z axis (unnormalised)
    // x = up x z with up = (0, 1, 0);  y = z x x   (torch.cross: a x b = (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0))
    const float x[3] = {1.0f * z[2] - 0.0f * z[1], 0.0f * z[0] - 0.0f * z[2], 0.0f * z[1] - 1.0f * z[0]};
    const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    const float *rows[3] = {x, y, z};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float *a = rows[r];
        const float n = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) cam_mat[9 * i + 3 * r + k] = a[k] / n;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) cam_pos[3 * i + k] = z[k];
}
} // namespace

// cam_mat [b,3,3] (rows = the camera's x, y, z axes, normalised), cam_pos [b,3] from param [b,3] = (azimuth deg, elevation deg,
// distance): what geom_pool_features_* take.  One launch.
extern "C" int geom_camera_info_f32(int b, const float *param, float *cam_mat, float *cam_pos, void *stream)
{
    if (b < 0) return GEOM_EINVAL;
    if (b == 0) return 0;
    if (!param || !cam_mat || !cam_pos) return GEOM_EINVAL;
    hipLaunchKernelGGL(camera_info_kernel, dim3((b + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), b, param, cam_mat, cam_pos);
    return geom::launch_status();
}
