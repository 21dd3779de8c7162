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

} // namespace geom
