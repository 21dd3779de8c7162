// Fused per-vertex BatchNorm + ReLU + residual average for the mesh deformation block
// (SURVEY section 8f, row 2).
//
// The reference block (models.py:237-297) follows every 0N-GCN layer with
//     x = F.relu(self.bnK(x))            nn.BatchNorm1d(verts) on [B,V,C]: one statistic per VERTEX,
//                                         taken over the B*C values of that vertex
//     features = features + x ; features /= 2        (every second layer)
// i.e. a two-pass batch-norm, a ReLU, an add and a divide: five passes over [B,V,C] forward and as
// many backward, 14 + 13 times per block and three blocks per step.  Here one workgroup owns one
// vertex: it reads the vertex's B rows once (kept in registers), reduces mean / variance in LDS in a
// fixed order, and writes the final value once; the backward likewise reads x and grad_out once and
// writes grad_x (and grad_residual) once, with the two BN reductions done in the workgroup.
//
// Semantics = torch.nn.BatchNorm1d in training mode: biased variance for the normalisation,
// running_mean/var updated with `momentum` and the UNBIASED variance; eval mode uses the running
// statistics.
#include "geom_common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_PER_THREAD = 16; // register-resident path: B*C <= 4096

struct BnArgs {
    const float *x;        // [b, nv, c]
    const float *weight;   // [nv] or null (=1)
    const float *bias;     // [nv] or null (=0)
    const float *res;      // optional residual, row stride res_ld (>= c), [b, nv, res_ld]
    float *out;            // [b, nv, c]
    float *run_mean, *run_var;  // [nv], updated when training (may be null)
    float *save_mean, *save_invstd; // [nv] (training forward writes, backward reads)
    int b, nv, c, res_ld;
    float eps, momentum, scale; // out = (res + act(bn(x))) * scale   (scale = 1 without a residual)
    int relu, training;
};

// fixed-order block reduction of two values (wave shuffles, then the 4 wave partials in order)
__device__ __forceinline__ void block_sum2(float &a, float &b, float *lds)
{
    for (int off = GEOM_WAVE / 2; off > 0; off >>= 1) {
        a += __shfl_down(a, off, GEOM_WAVE);
        b += __shfl_down(b, off, GEOM_WAVE);
    }
    const int lane = threadIdx.x & (GEOM_WAVE - 1), wave = threadIdx.x >> 6;
    __syncthreads(); // lds may still be read from a previous call
    if (lane == 0) {
        lds[2 * wave] = a;
        lds[2 * wave + 1] = b;
    }
    __syncthreads();
    a = 0.f, b = 0.f;
    for (int w = 0; w < BN_THREADS / GEOM_WAVE; ++w) {
        a += lds[2 * w];
        b += lds[2 * w + 1];
    }
}

__global__ __launch_bounds__(BN_THREADS) void vertex_bn_fwd_kernel(BnArgs a)
{
    __shared__ float lds[2 * BN_THREADS / GEOM_WAVE];
    const int v = blockIdx.x;
    const int n = a.b * a.c;
    float xv[BN_MAX_PER_THREAD];
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int i = 0; i < BN_MAX_PER_THREAD; ++i) {
        const int e = threadIdx.x + i * BN_THREADS;
        xv[i] = 0.f;
        if (e < n) {
            const int bi = e / a.c, ci = e - bi * a.c;
            xv[i] = a.x[((size_t)bi * a.nv + v) * a.c + ci];
            s += xv[i];
        }
    }
    float mean, invstd;
    if (a.training) {
        block_sum2(s, dummy, lds);
        mean = s / n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < BN_MAX_PER_THREAD; ++i)
            if (threadIdx.x + i * BN_THREADS < n) {
                const float d = xv[i] - mean;
                q += d * d;
            }
        dummy = 0.f;
        block_sum2(q, dummy, lds);
        const float var = q / n; // biased, as used for normalisation
        invstd = 1.f / sqrtf(var + a.eps);
        if (threadIdx.x == 0) {
            a.save_mean[v] = mean;
            a.save_invstd[v] = invstd;
            if (a.run_mean) a.run_mean[v] = (1.f - a.momentum) * a.run_mean[v] + a.momentum * mean;
            if (a.run_var) a.run_var[v] = (1.f - a.momentum) * a.run_var[v] + a.momentum * (n > 1 ? q / (n - 1) : var);
        }
    } else {
        mean = a.run_mean[v];
        invstd = 1.f / sqrtf(a.run_var[v] + a.eps);
    }
    const float g = a.weight ? a.weight[v] : 1.f, be = a.bias ? a.bias[v] : 0.f;
#pragma unroll
    for (int i = 0; i < BN_MAX_PER_THREAD; ++i) {
        const int e = threadIdx.x + i * BN_THREADS;
        if (e < n) {
            const int bi = e / a.c, ci = e - bi * a.c;
            float y = (xv[i] - mean) * invstd * g + be;
            if (a.relu) y = y > 0.f ? y : 0.f;
            if (a.res) y = (a.res[((size_t)bi * a.nv + v) * a.res_ld + ci] + y) * a.scale;
            a.out[((size_t)bi * a.nv + v) * a.c + ci] = y;
        }
    }
}

// Vector path (c % 4 == 0, b <= 4 * (256 / (c/4)) rows -- every hidden layer of the deformation block): thread
// (r0, q) owns float4 column group q of rows r0, r0 + R, r0 + 2R, r0 + 3R (R = 256 / (c/4) rows per pass), so the
// vertex's B rows arrive as 16-byte loads of contiguous 4c-byte runs, with no per-element division, and the residual
// is fetched in the SAME round trip as x instead of after the statistics (the scalar kernel above: 19 us per launch at
// the reference's training shape B = 16, V = 482, C = 192 for 18 MB of traffic).
constexpr int BN_VEC_ITERS = 4;

__global__ __launch_bounds__(BN_THREADS) void vertex_bn_fwd_vec_kernel(BnArgs a)
{
    __shared__ float lds[2 * BN_THREADS / GEOM_WAVE];
    const int v = blockIdx.x;
    const int c4 = a.c >> 2, rows = BN_THREADS / c4;
    const int r0 = threadIdx.x / c4, q = threadIdx.x - r0 * c4;
    const bool lane_on = r0 < rows;
    const int n = a.b * a.c;
    // the vertex's parameters and running statistics are requested up front, in the round trip of the rows: read where
    // they are used they add a dependent trip after the reductions (the launch is a latency chain, not a bandwidth job)
    const float g = a.weight ? a.weight[v] : 1.f, be = a.bias ? a.bias[v] : 0.f;
    const bool updates = a.training && threadIdx.x == 0;
    const float old_mean = (updates && a.run_mean) ? a.run_mean[v] : 0.f;
    const float old_var = (updates && a.run_var) ? a.run_var[v] : 0.f;
    float4 xv[BN_VEC_ITERS], rv[BN_VEC_ITERS];
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int i = 0; i < BN_VEC_ITERS; ++i) {
        const int bi = r0 + i * rows;
        xv[i] = rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_on && bi < a.b) {
            xv[i] = *reinterpret_cast<const float4 *>(a.x + ((size_t)bi * a.nv + v) * a.c + 4 * q);
            if (a.res) rv[i] = *reinterpret_cast<const float4 *>(a.res + ((size_t)bi * a.nv + v) * a.res_ld + 4 * q);
        }
    }
#pragma unroll
    for (int i = 0; i < BN_VEC_ITERS; ++i) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w); // absent rows hold zeros
    float mean, invstd;
    if (a.training) {
        block_sum2(s, dummy, lds);
        mean = s / n;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < BN_VEC_ITERS; ++i)
            if (lane_on && r0 + i * rows < a.b) {
                const float d0 = xv[i].x - mean, d1 = xv[i].y - mean, d2 = xv[i].z - mean, d3 = xv[i].w - mean;
                qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        dummy = 0.f;
        block_sum2(qq, dummy, lds);
        const float var = qq / n; // biased, as used for normalisation
        invstd = 1.f / sqrtf(var + a.eps);
        if (threadIdx.x == 0) {
            a.save_mean[v] = mean;
            a.save_invstd[v] = invstd;
            if (a.run_mean) a.run_mean[v] = (1.f - a.momentum) * old_mean + a.momentum * mean;
            if (a.run_var) a.run_var[v] = (1.f - a.momentum) * old_var + a.momentum * (n > 1 ? qq / (n - 1) : var);
        }
    } else {
        mean = a.run_mean[v];
        invstd = 1.f / sqrtf(a.run_var[v] + a.eps);
    }
    auto finish = [&](float x, float r) {
        float y = (x - mean) * invstd * g + be;
        if (a.relu) y = y > 0.f ? y : 0.f;
        if (a.res) y = (r + y) * a.scale;
        return y;
    };
#pragma unroll
    for (int i = 0; i < BN_VEC_ITERS; ++i) {
        const int bi = r0 + i * rows;
        if (lane_on && bi < a.b)
            *reinterpret_cast<float4 *>(a.out + ((size_t)bi * a.nv + v) * a.c + 4 * q) =
                make_float4(finish(xv[i].x, rv[i].x), finish(xv[i].y, rv[i].y), finish(xv[i].z, rv[i].z), finish(xv[i].w, rv[i].w));
    }
}

struct BnBwdArgs {
    const float *x, *grad_out, *grad_out2, *weight, *bias, *save_mean, *save_invstd; // grad_out2: optional second upstream
    //   gradient of the same output (it fed a layer AND a residual average): summed on the fly, one add as autograd's
    float *grad_x, *grad_res, *grad_weight, *grad_bias; // grad_res [b,nv,c] optional
    int b, nv, c;
    float scale;
    int relu, has_res;
};

__global__ __launch_bounds__(BN_THREADS) void vertex_bn_bwd_kernel(BnBwdArgs a)
{
    __shared__ float lds[2 * BN_THREADS / GEOM_WAVE];
    const int v = blockIdx.x;
    const int n = a.b * a.c;
    const float mean = a.save_mean[v], invstd = a.save_invstd[v];
    const float g = a.weight ? a.weight[v] : 1.f, be = a.bias ? a.bias[v] : 0.f;
    float xh[BN_MAX_PER_THREAD], gy[BN_MAX_PER_THREAD];
    float sum_g = 0.f, sum_gx = 0.f;
#pragma unroll
    for (int i = 0; i < BN_MAX_PER_THREAD; ++i) {
        const int e = threadIdx.x + i * BN_THREADS;
        xh[i] = 0.f, gy[i] = 0.f;
        if (e < n) {
            const int bi = e / a.c, ci = e - bi * a.c;
            const size_t o = ((size_t)bi * a.nv + v) * a.c + ci;
            xh[i] = (a.x[o] - mean) * invstd;
            float go = a.grad_out[o];
            if (a.grad_out2) go += a.grad_out2[o];
            if (a.has_res) {
                go *= a.scale;
                if (a.grad_res) a.grad_res[o] = go;
            }
            if (a.relu && !(xh[i] * g + be > 0.f)) go = 0.f;
            gy[i] = go;
            sum_g += go;
            sum_gx += go * xh[i];
        }
    }
    block_sum2(sum_g, sum_gx, lds);
    if (threadIdx.x == 0) {
        if (a.grad_bias) a.grad_bias[v] = sum_g;
        if (a.grad_weight) a.grad_weight[v] = sum_gx;
    }
    const float k = g * invstd, mg = sum_g / n, mgx = sum_gx / n;
#pragma unroll
    for (int i = 0; i < BN_MAX_PER_THREAD; ++i) {
        const int e = threadIdx.x + i * BN_THREADS;
        if (e < n) {
            const int bi = e / a.c, ci = e - bi * a.c;
            a.grad_x[((size_t)bi * a.nv + v) * a.c + ci] = k * (gy[i] - mg - xh[i] * mgx);
        }
    }
}

__global__ __launch_bounds__(BN_THREADS) void vertex_bn_bwd_vec_kernel(BnBwdArgs a)
{
    __shared__ float lds[2 * BN_THREADS / GEOM_WAVE];
    const int v = blockIdx.x;
    const int c4 = a.c >> 2, rows = BN_THREADS / c4;
    const int r0 = threadIdx.x / c4, q = threadIdx.x - r0 * c4;
    const bool lane_on = r0 < rows;
    const int n = a.b * a.c;
    const float mean = a.save_mean[v], invstd = a.save_invstd[v];
    const float g = a.weight ? a.weight[v] : 1.f, be = a.bias ? a.bias[v] : 0.f;
    float4 xh[BN_VEC_ITERS], gy[BN_VEC_ITERS], g2[BN_VEC_ITERS];
    float sum_g = 0.f, sum_gx = 0.f;
#pragma unroll
    for (int i = 0; i < BN_VEC_ITERS; ++i) { // all operands of every row in one round trip
        const int bi = r0 + i * rows;
        xh[i] = gy[i] = g2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_on && bi < a.b) {
            const size_t o = ((size_t)bi * a.nv + v) * a.c + 4 * q;
            xh[i] = *reinterpret_cast<const float4 *>(a.x + o);
            gy[i] = *reinterpret_cast<const float4 *>(a.grad_out + o);
            if (a.grad_out2) g2[i] = *reinterpret_cast<const float4 *>(a.grad_out2 + o);
        }
    }
    if (a.grad_out2) {
#pragma unroll
        for (int i = 0; i < BN_VEC_ITERS; ++i) gy[i].x += g2[i].x, gy[i].y += g2[i].y, gy[i].z += g2[i].z, gy[i].w += g2[i].w;
    }
    auto one = [&](float &x, float &go) {
        x = (x - mean) * invstd;
        if (a.has_res) go *= a.scale;
        const float pass = go;
        if (a.relu && !(x * g + be > 0.f)) go = 0.f;
        sum_g += go;
        sum_gx += go * x;
        return pass;
    };
#pragma unroll
    for (int i = 0; i < BN_VEC_ITERS; ++i) {
        const int bi = r0 + i * rows;
        if (lane_on && bi < a.b) {
            const float4 r = make_float4(one(xh[i].x, gy[i].x), one(xh[i].y, gy[i].y), one(xh[i].z, gy[i].z), one(xh[i].w, gy[i].w));
            if (a.has_res && a.grad_res) *reinterpret_cast<float4 *>(a.grad_res + ((size_t)bi * a.nv + v) * a.c + 4 * q) = r;
        }
    }
    block_sum2(sum_g, sum_gx, lds);
    if (threadIdx.x == 0) {
        if (a.grad_bias) a.grad_bias[v] = sum_g;
        if (a.grad_weight) a.grad_weight[v] = sum_gx;
    }
    const float k = g * invstd, mg = sum_g / n, mgx = sum_gx / n;
#pragma unroll
    for (int i = 0; i < BN_VEC_ITERS; ++i) {
        const int bi = r0 + i * rows;
        if (lane_on && bi < a.b)
            *reinterpret_cast<float4 *>(a.grad_x + ((size_t)bi * a.nv + v) * a.c + 4 * q) =
                make_float4(k * (gy[i].x - mg - xh[i].x * mgx), k * (gy[i].y - mg - xh[i].y * mgx),
                            k * (gy[i].z - mg - xh[i].z * mgx), k * (gy[i].w - mg - xh[i].w * mgx));
    }
}

inline bool bn_vec_ok(int b, int c, uintptr_t ptr_bits, int extra_ld)
{
    if (c % 4 != 0 || c / 4 > BN_THREADS || (ptr_bits & 15) || (extra_ld & 3)) return false;
    const int rows = BN_THREADS / (c / 4);
    return b <= BN_VEC_ITERS * rows;
}

} // namespace

extern "C" int geom_vertex_bn_fwd_f32(int b, int nv, int c, const float *x, const float *weight, const float *bias,
                                      float *running_mean, float *running_var, int training, float momentum, float eps,
                                      int relu, const float *residual, int residual_ld, float scale,
                                      float *out, float *save_mean, float *save_invstd, void *stream)
{
    if (b < 0 || nv < 0 || c < 0) return GEOM_EINVAL;
    if ((int64_t)b * c > BN_THREADS * BN_MAX_PER_THREAD) return GEOM_EUNSUPPORTED;
    if (b == 0 || nv == 0 || c == 0) return 0;
    if (!x || !out) return GEOM_EINVAL;
    if (training ? (!save_mean || !save_invstd) : (!running_mean || !running_var)) return GEOM_EINVAL;
    if (residual && residual_ld < c) return GEOM_EINVAL;
    BnArgs a{x, weight, bias, residual, out, running_mean, running_var, save_mean, save_invstd,
             b, nv, c, residual_ld, eps, momentum, residual ? scale : 1.f, relu, training};
    if (bn_vec_ok(b, c, (uintptr_t)x | (uintptr_t)out | (uintptr_t)residual, residual ? residual_ld : 0))
        hipLaunchKernelGGL(vertex_bn_fwd_vec_kernel, dim3(nv), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(vertex_bn_fwd_kernel, dim3(nv), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return geom::launch_status();
}

extern "C" int geom_vertex_bn_bwd_f32(int b, int nv, int c, const float *x, const float *grad_out, const float *weight,
                                      const float *bias, const float *save_mean, const float *save_invstd, int relu,
                                      int has_residual, float scale, float *grad_x, float *grad_residual,
                                      float *grad_weight, float *grad_bias, const float *grad_out2, void *stream)
{
    if (b < 0 || nv < 0 || c < 0) return GEOM_EINVAL;
    if ((int64_t)b * c > BN_THREADS * BN_MAX_PER_THREAD) return GEOM_EUNSUPPORTED;
    if (b == 0 || nv == 0 || c == 0) return 0;
    if (!x || !grad_out || !save_mean || !save_invstd || !grad_x) return GEOM_EINVAL;
    BnBwdArgs a{x, grad_out, grad_out2, weight, bias, save_mean, save_invstd, grad_x, grad_residual, grad_weight, grad_bias,
                b, nv, c, has_residual ? scale : 1.f, relu, has_residual};
    if (bn_vec_ok(b, c, (uintptr_t)x | (uintptr_t)grad_out | (uintptr_t)grad_out2 | (uintptr_t)grad_x | (uintptr_t)grad_residual, 0))
        hipLaunchKernelGGL(vertex_bn_bwd_vec_kernel, dim3(nv), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(vertex_bn_bwd_kernel, dim3(nv), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return geom::launch_status();
}
