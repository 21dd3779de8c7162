// Shared device/host helpers for libgeom_hip.so (gfx950 only; wave = 64 lanes).
//
// All kernels are compiled with -ffp-contract=off: every product and sum is a
// separately rounded binary32 operation in exactly the order written, which is
// the arithmetic the CPU oracle (oracle/geom_oracle.c) pins.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <math.h>

#include "geom_hip.h"

#define GEOM_WAVE 64

namespace geom {

// squared distance, reference operation order: (dx*dx + dy*dy) + dz*dz, d = target - query
// (chamfer_distance/chamfer_distance.cu:35-38, old_GEOMetrics/.../my_lib.c:13-16)
__device__ __forceinline__ float sqdist3(float tx, float ty, float tz, float qx, float qy, float qz)
{
    const float dx = tx - qx;
    const float dy = ty - qy;
    const float dz = tz - qz;
    const float xx = dx * dx;
    const float yy = dy * dy;
    const float zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

// The other admissible canonical form (GEOM_FLAG_NN_FMA): what a contracting compiler makes of the same source line
// -- fma(dz, dz, fma(dx, dx, dy*dy)), bit-identical to the reference's nnsearch built with gcc -mfma
// -ffp-contract=fast (oracle/_ref/libref_nnsearch_fma.so).  Explicit fmaf: the file is compiled -ffp-contract=off.
__device__ __forceinline__ float sqdist3_fma(float tx, float ty, float tz, float qx, float qy, float qz)
{
    const float dx = tx - qx;
    const float dy = ty - qy;
    const float dz = tz - qz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}

// lexicographic (distance, index) "is b better than a": what a strict-'<'
// first-wins sequential scan reduces to when partial scans are merged.
__device__ __forceinline__ bool lex_less(float bd, int bi, float ad, int ai)
{
    return (bd < ad) || (bd == ad && bi < ai);
}

inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// reference tile length whose tail is truncated under GEOM_FLAG_REF_TAIL_TRUNC
constexpr int REF_TILE = 512;

// true when target k of an m-long set is skipped by the shipped CUDA kernels'
// `end_ka = end_k - (end_k & 3)` loop bound
__host__ __device__ __forceinline__ bool ref_tail_skipped(int k, int m)
{
    const int t0 = (k / REF_TILE) * REF_TILE;
    const int len = (m - t0 < REF_TILE) ? (m - t0) : REF_TILE;
    return (k - t0) >= len - (len & 3);
}

// XCD-aware work assignment.  MI355X dispatches workgroup `bid` to XCD `bid % 8`, and every XCD
// has its own 4 MiB L2.  Jobs (a mesh, or a (direction, mesh) pair) whose workgroups all read the
// same records are therefore pinned to one XCD: slot = bid % 8 owns jobs slot, slot+8, ...; within a
// slot the `tiles` workgroups of a job are consecutive.  Grid size = 8 * tiles * ceil(jobs / 8);
// workgroups whose job index falls beyond `jobs` exit.  Placement only affects speed (L2 hits
// instead of 8x refetch through the fabric), never results.
// A store other workgroups of the SAME launch may read after a flag hand-off (the finalize tail of the fused surface scan):
// agent scope = written through the XCD's L2, so the producer needs no L2 write-back (a release fence costs a buffer_wbl2
// per wave: 9 000 of them made that launch 245 us instead of 45) -- only s_waitcnt vmcnt(0) before it signs off.
template <typename T>
__device__ __forceinline__ void store_agent(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int NUM_XCD = 8;
// With fewer than 8 jobs every job is spread over floor(8 / jobs) XCDs (its tiles interleaved), so
// a small batch still uses the whole chip.
__host__ __device__ __forceinline__ int xcd_spread(int jobs) { return jobs >= NUM_XCD ? 1 : NUM_XCD / jobs; }
__host__ __device__ __forceinline__ unsigned xcd_grid(int jobs, int tiles)
{
    const int spread = xcd_spread(jobs);
    if (spread > 1) return (unsigned)(NUM_XCD * ((tiles + spread - 1) / spread));
    return (unsigned)(NUM_XCD * tiles * ((jobs + NUM_XCD - 1) / NUM_XCD));
}
__device__ __forceinline__ bool xcd_assign(int bid, int jobs, int tiles, int &job, int &tile)
{
    const int slot = bid % NUM_XCD;
    const int j = bid / NUM_XCD;
    const int spread = xcd_spread(jobs);
    if (spread > 1) {
        job = slot / spread;
        tile = j * spread + slot % spread;
        return job < jobs && tile < tiles;
    }
    job = slot + NUM_XCD * (j / tiles);
    tile = j % tiles;
    return job < jobs;
}

// Job j of a two-directional NN scan over b meshes -> direction (0: cloud 1 queries cloud 2).  Direction 1 comes FIRST in the
// grid: in the surface loss it is the sampled points' scan, the one the loss (and its in-launch finalize roles) waits for.
__host__ __device__ __forceinline__ int nn_job_dir(int job, int b) { return 1 - job / b; }

} // namespace geom
