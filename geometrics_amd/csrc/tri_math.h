// Point-to-triangle closest-feature arithmetic shared by the tri_distance scan and the
// point-to-surface loss kernels.  Operation order follows the reference kernel
// (tri_distance/tri_distance.cu:6-91 helpers, :140-191 decision tree) so that the HIP
// result is bit-identical to oracle/geom_oracle.c under -ffp-contract=off.
#pragma once
#include "geom_common.h"

namespace geom {

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }

__device__ __forceinline__ float dot3(V3 a, V3 b)
{
    const float xx = a.x * b.x, yy = a.y * b.y, zz = a.z * b.z;
    const float s = xx + yy;
    return s + zz;
}

__device__ __forceinline__ V3 cross3(V3 a, V3 b)
{
    const float p0 = a.y * b.z, p1 = a.z * b.y;
    const float p2 = a.z * b.x, p3 = a.x * b.z;
    const float p4 = a.x * b.y, p5 = a.y * b.x;
    return V3{p0 - p1, p2 - p3, p4 - p5};
}

// foot of the perpendicular on the triangle plane (Project_Plane + normalize,
// tri_distance.cu:76-91): the reciprocal is a DOUBLE division narrowed to float.
__device__ __forceinline__ V3 plane_foot(V3 org, V3 nrm, V3 p)
{
    const V3 v = p - org;
    const float len = sqrtf(dot3(nrm, nrm));
    const float inv = (float)(1.0 / (double)len);
    const V3 unit = nrm * inv;
    const float h = dot3(v, unit);
    return p - unit * h;
}

// One (point, triangle) evaluation, literal form: every quantity is derived from the
// three corners on the spot, divisions included.  Returns the squared distance to the
// chosen closest point and its 0..6 region code.
template <bool FIX6>
__device__ __forceinline__ float tri_pair_literal(V3 p, V3 A, V3 B, V3 C, int &opt)
{
    const V3 dAB = B - A;
    const V3 dBC = C - B;
    const V3 dCA = A - C;
    const V3 nrm = cross3(A - B, A - C);
    const V3 vA = p - A;
    const V3 vC = p - C;
    const float uab = dot3(vA, dAB) / dot3(dAB, dAB);
    const float uca = dot3(vC, dCA) / dot3(dCA, dCA);
    V3 hit;
    if (uca > 1.f && uab < 0.f) {
        hit = A;
        opt = 1;
    } else {
        const V3 vB = p - B;
        const float ubc = dot3(vB, dBC) / dot3(dBC, dBC);
        if (uab > 1.f && ubc < 0.f) {
            hit = B;
            opt = 2;
        } else if (ubc > 1.f && uca < 0.f) {
            hit = C;
            opt = 3;
        } else if ((uab <= 1.f && uab >= 0.f) && !(dot3(cross3(nrm, dAB), vA) > 0.f)) {
            hit = A + dAB * uab;
            opt = 4;
        } else if ((ubc <= 1.f && ubc >= 0.f) && !(dot3(cross3(nrm, dBC), vB) > 0.f)) {
            hit = B + dBC * ubc;
            opt = 5;
        } else if ((uca <= 1.f && uca >= 0.f) && !(dot3(cross3(nrm, dCA), vC) > 0.f)) {
            hit = C + (FIX6 ? dCA : dAB) * uca; // Q2: the reference walks along AB (tri_distance.cu:180)
            opt = 6;
        } else {
            hit = plane_foot(A, nrm, p);
            opt = 0;
        }
    }
    const V3 diff = p - hit;
    return dot3(diff, diff);
}

// Closest point of calc_point_to_line (utils.py:506-550): the SELECTED candidate only,
// with its affine weights on (A,B,C).  Option 6 walks along CA here (utils.py:543) -- the
// kernel-side Q2 quirk only influences which triangle won.
__device__ __forceinline__ V3 closest_on_triangle(V3 p, V3 A, V3 B, V3 C, int opt, V3 &w)
{
    if (opt == 1) { w = mk(1.f, 0.f, 0.f); return A; }
    if (opt == 2) { w = mk(0.f, 1.f, 0.f); return B; }
    if (opt == 3) { w = mk(0.f, 0.f, 1.f); return C; }
    if (opt == 4) {
        const V3 d = B - A;
        const float t = dot3(p - A, d) / dot3(d, d);
        w = mk(1.f - t, t, 0.f);
        return A + d * t;
    }
    if (opt == 5) {
        const V3 d = C - B;
        const float t = dot3(p - B, d) / dot3(d, d);
        w = mk(0.f, 1.f - t, t);
        return B + d * t;
    }
    if (opt == 6) {
        const V3 d = A - C;
        const float t = dot3(p - C, d) / dot3(d, d);
        w = mk(t, 0.f, 1.f - t);
        return C + d * t;
    }
    // plane projection (Plane.Project, utils.py:573-587): n = N / sqrt(sum N^2)
    const V3 N = cross3(A - B, A - C);
    const float len = sqrtf(dot3(N, N));
    const V3 n = mk(N.x / len, N.y / len, N.z / len);
    const float h = dot3(p - A, n);
    const V3 q = p - n * h;
    // affine weights of q: q - A = s (B-A) + t (C-A), 2x2 normal equations
    const V3 e1 = B - A, e2 = C - A, r = q - A;
    const float a11 = dot3(e1, e1), a12 = dot3(e1, e2), a22 = dot3(e2, e2);
    const float b1 = dot3(r, e1), b2 = dot3(r, e2);
    const float det = a11 * a22 - a12 * a12;
    const float s = (b1 * a22 - b2 * a12) / det;
    const float t = (b2 * a11 - b1 * a12) / det;
    w = mk(1.f - s - t, s, t);
    return q;
}

} // namespace geom
