// Per-mesh column maximum over a RAGGED batch of meshes.
//
// The reference encodes one mesh at a time (auto_encoder.py:71-76: a python loop of 17 layer calls per mesh,
// each with its own dense [V,V] adjacency) and ends every mesh with GCNMax's `torch.max(i_s, dim=0)`
// (layers.py:78).  Here the meshes of a batch are concatenated along the vertex axis (block-diagonal
// CSR, see geometrics_amd/ragged.py), so each layer is ONE GEMM + ONE aggregation launch for the whole
// batch, and the per-mesh max becomes a segmented column max over rows [offsets[s], offsets[s+1]).
//
//   pass 1: grid (splits, segments, column tiles); a workgroup = 4 row lanes x 64 columns scans its
//           slice of the segment's rows with coalesced 256-byte row reads and writes one partial
//           (value, row) per column;
//   pass 2: a thread per (segment, column) folds the partials in slice order.
// Ties keep the LOWEST row and the first NaN wins (what a sequential `>` scan with NaN propagation
// gives); the result does not depend on the split count.  The backward is a gather: every element of
// grad_x is written once (grad_out where its row is the arg-max, else 0) -- no zero-fill, no atomics.
#include "geom_common.h"

namespace {

constexpr int SG_THREADS = 256;
constexpr int SG_COLS = 64;
constexpr int SG_ROWLANES = SG_THREADS / SG_COLS; // 4
constexpr int SG_MAX_SPLITS = 64;

// is (v, r) a better maximum than (bv, br)?  br < 0 = nothing yet
__device__ __forceinline__ bool seg_better(float v, int r, float bv, int br)
{
    if (br < 0) return true;
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;        // a NaN beats any number
    if (!vn && v != bv) return v > bv;
    return r < br;                  // equal values (or both NaN): first row wins
}

__global__ __launch_bounds__(SG_THREADS) void segment_max_partial_kernel(const int64_t *offsets, int c, int splits,
                                                                          const float *x, float *part_v, int *part_r)
{
    __shared__ float sv[SG_ROWLANES][SG_COLS];
    __shared__ int sr[SG_ROWLANES][SG_COLS];
    const int split = blockIdx.x, seg = blockIdx.y;
    const int cl = threadIdx.x & (SG_COLS - 1), rl = threadIdx.x >> 6;
    const int col = blockIdx.z * SG_COLS + cl;
    const int64_t r0 = offsets[seg], r1 = offsets[seg + 1];
    const int64_t per = (r1 - r0 + splits - 1) / splits;
    const int64_t begin = r0 + (int64_t)split * per;
    const int64_t end = begin + per < r1 ? begin + per : r1;

    float bv = 0.f;
    int br = -1;
    if (col < c) {
        for (int64_t r = begin + rl; r < end; r += SG_ROWLANES) {
            const float v = x[r * c + col];
            if (seg_better(v, (int)(r - r0), bv, br)) {
                bv = v;
                br = (int)(r - r0);
            }
        }
    }
    sv[rl][cl] = bv;
    sr[rl][cl] = br;
    __syncthreads();
    if (rl == 0 && col < c) {
#pragma unroll
        for (int k = 1; k < SG_ROWLANES; ++k)
            if (sr[k][cl] >= 0 && seg_better(sv[k][cl], sr[k][cl], bv, br)) {
                bv = sv[k][cl];
                br = sr[k][cl];
            }
        const size_t o = ((size_t)seg * splits + split) * c + col;
        part_v[o] = bv;
        part_r[o] = br;
    }
}

__global__ __launch_bounds__(SG_THREADS) void segment_max_final_kernel(int nseg, int c, int splits, const float *part_v,
                                                                        const int *part_r, float *out, int *arg)
{
    const int64_t i = (int64_t)blockIdx.x * SG_THREADS + threadIdx.x;
    if (i >= (int64_t)nseg * c) return;
    const int seg = (int)(i / c), col = (int)(i - (int64_t)seg * c);
    float bv = -INFINITY;
    int br = -1;
    for (int s = 0; s < splits; ++s) {
        const size_t o = ((size_t)seg * splits + s) * c + col;
        const int r = part_r[o];
        if (r >= 0 && seg_better(part_v[o], r, bv, br)) {
            bv = part_v[o];
            br = r;
        }
    }
    out[i] = bv; // an empty segment yields -inf / -1
    arg[i] = br;
}

__global__ __launch_bounds__(SG_THREADS) void segment_max_bwd_kernel(int nseg, const int64_t *offsets, int c,
                                                                      const float *grad_out, const int *arg,
                                                                      float *grad_x)
{
    const int64_t total = offsets[nseg];
    const int64_t i = (int64_t)blockIdx.x * SG_THREADS + threadIdx.x;
    if (i >= total * c) return;
    const int64_t r = i / c;
    const int col = (int)(i - r * c);
    int lo = 0, hi = nseg - 1; // last segment whose first row is <= r
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= r) lo = mid;
        else hi = mid - 1;
    }
    const size_t o = (size_t)lo * c + col;
    grad_x[i] = (arg[o] == (int)(r - offsets[lo])) ? grad_out[o] : 0.f;
}

inline int seg_splits(int64_t max_len)
{
    const int64_t s = (max_len + 16 * SG_ROWLANES - 1) / (16 * SG_ROWLANES); // about 16 rows per thread
    return (int)(s < 1 ? 1 : (s > SG_MAX_SPLITS ? SG_MAX_SPLITS : s));
}

} // namespace

extern "C" int64_t geom_segment_max_workspace_bytes(int nseg, int c, int64_t max_len)
{
    if (nseg <= 0 || c <= 0) return 0;
    return (int64_t)nseg * seg_splits(max_len) * c * 8;
}

extern "C" int geom_segment_max_fwd_f32(int nseg, const int64_t *offsets, int64_t max_len, int c, const float *x,
                                        float *out, int *arg, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (nseg < 0 || c < 0 || max_len < 0) return GEOM_EINVAL;
    if (nseg == 0 || c == 0) return 0;
    if (!offsets || !out || !arg || (max_len > 0 && !x)) return GEOM_EINVAL;
    if (nseg > 65535) return GEOM_ETOOBIG;
    const int splits = seg_splits(max_len);
    if (!workspace || workspace_bytes < geom_segment_max_workspace_bytes(nseg, c, max_len)) return GEOM_EINVAL;
    float *part_v = static_cast<float *>(workspace);
    int *part_r = reinterpret_cast<int *>(part_v + (size_t)nseg * splits * c);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(segment_max_partial_kernel, dim3(splits, nseg, (c + SG_COLS - 1) / SG_COLS), dim3(SG_THREADS), 0, s,
                       offsets, c, splits, x, part_v, part_r);
    hipLaunchKernelGGL(segment_max_final_kernel, dim3((unsigned)(((int64_t)nseg * c + SG_THREADS - 1) / SG_THREADS)),
                       dim3(SG_THREADS), 0, s, nseg, c, splits, part_v, part_r, out, arg);
    return geom::launch_status();
}

extern "C" int geom_segment_max_bwd_f32(int nseg, const int64_t *offsets, int64_t total_rows, int c,
                                        const float *grad_out, const int *arg, float *grad_x, void *stream)
{
    if (nseg < 0 || c < 0 || total_rows < 0) return GEOM_EINVAL;
    if (nseg == 0 || c == 0 || total_rows == 0) return 0;
    if (!offsets || !grad_out || !arg || !grad_x) return GEOM_EINVAL;
    const int64_t blocks = (total_rows * c + SG_THREADS - 1) / SG_THREADS;
    if (blocks > 0x7fffffffLL) return GEOM_ETOOBIG;
    hipLaunchKernelGGL(segment_max_bwd_kernel, dim3((unsigned)blocks), dim3(SG_THREADS), 0,
                       static_cast<hipStream_t>(stream), nseg, offsets, c, grad_out, arg, grad_x);
    return geom::launch_status();
}
