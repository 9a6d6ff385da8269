// tsp_colormap.hip -- the colormap / log-scale post-pass (kernels B and B').
//
// Reference: src/topsy/shaders/colormap.wgsl fragment_main non-bivariate branch (:113-127),
// log10 (:75-77), fragment_main_tri + gamma_map (:131-159); the LUT is the 1000 x rgba32float
// 1-D texture of Colormap._setup_map_texture (src/topsy/colormap/implementation.py:205-238),
// sampled with a linear filter and clamp-to-edge; the target is rgba8unorm.
//
// HBM-bound elementwise map: 8 (or 16) B in, 4 B out per pixel, one pixel per lane, the 16 KB LUT
// stays in L1/L2.  Arithmetic is the canonical order of tsp_math.h (no FMA contraction) so the
// uint8 result is bit-reproducible against the CPU oracle.
#include "tsp_internal.h"

namespace tsp {

__global__ __launch_bounds__(256) void colormap_scalar_kernel(const float *__restrict__ img, int64_t npix, int C,
                                                              const float4 *__restrict__ lut, int n_lut, float vmin,
                                                              float vmax, int log_scale, int weighted,
                                                              uint32_t *__restrict__ out) {
    const float range = vmax - vmin;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        float r, g;
        if (C == 2) {
            const float2 v2 = *reinterpret_cast<const float2 *>(img + p * 2);
            r = v2.x; g = v2.y;
        } else {
            r = img[p * C]; g = img[p * C + 1];
        }
        float v = weighted ? g / r : r;
        if (log_scale) v = canon_log10f(v);
        float t = (v - vmin) / range;
        t = (t != t) ? 0.0f : t;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        const float c = t * (float)n_lut - 0.5f;
        const float c0 = __builtin_floorf(c);
        const float f = c - c0;
        const int i0 = clampi((int)c0, 0, n_lut - 1), i1 = clampi((int)c0 + 1, 0, n_lut - 1);
        const float gq = 1.0f - f;
        const float4 a = lut[i0], b = lut[i1];
        const uint32_t R8 = unorm8(a.x * gq + b.x * f);
        const uint32_t G8 = unorm8(a.y * gq + b.y * f);
        const uint32_t B8 = unorm8(a.z * gq + b.z * f);
        const uint32_t A8 = unorm8(a.w * gq + b.w * f);
        out[p] = R8 | (G8 << 8) | (B8 << 16) | (A8 << 24);
    }
}

__global__ __launch_bounds__(256) void colormap_rgb_kernel(const float *__restrict__ img, int64_t npix, int C,
                                                           float vmin, float vmax, float gamma,
                                                           uint32_t *__restrict__ out8, float4 *__restrict__ outf) {
    const float range = vmax - vmin;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = canon_log10f(img[p * C + k]);
            float x = (v - vmin) / range;
            x = (x != x) ? 0.0f : x;
            x = x < 0.0f ? 0.0f : x;
            c[k] = canon_powf(x, gamma);
        }
        if (out8) out8[p] = unorm8(c[0]) | (unorm8(c[1]) << 8) | (unorm8(c[2]) << 16) | (255u << 24);
        if (outf) outf[p] = make_float4(c[0], c[1], c[2], 1.0f);
    }
}

// BIVARIATE branch of fragment_main (colormap.wgsl:91-111): x = normalised log10 density, y = normalised
// (weighted) value; 2-D LUT [y][x] with a linear filter and clamp-to-edge.
__global__ __launch_bounds__(256) void colormap_bivariate_kernel(const float *__restrict__ img, int64_t npix, int C,
                                                                 const float4 *__restrict__ lut, int n, float vmin,
                                                                 float vmax, float dvmin, float dvmax, int log_scale,
                                                                 int weighted, uint32_t *__restrict__ out) {
    const float range = vmax - vmin, drange = dvmax - dvmin;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const float r = img[p * C], g = img[p * C + 1];
        float x = (canon_log10f(r) - dvmin) / drange;
        float y = weighted ? g / r : r;
        if (log_scale) y = canon_log10f(y);
        y = (y - vmin) / range;
        x = (x != x || x < 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x);
        y = (y != y || y < 0.0f) ? 0.0f : (y > 1.0f ? 1.0f : y);
        const float cx = x * (float)n - 0.5f, cy = y * (float)n - 0.5f;
        const float x0 = __builtin_floorf(cx), y0 = __builtin_floorf(cy);
        const float fx = cx - x0, fy = cy - y0, gx = 1.0f - fx, gy = 1.0f - fy;
        const int i0 = clampi((int)x0, 0, n - 1), i1 = clampi((int)x0 + 1, 0, n - 1);
        const int j0 = clampi((int)y0, 0, n - 1), j1 = clampi((int)y0 + 1, 0, n - 1);
        const float4 a = lut[(size_t)j0 * n + i0], b = lut[(size_t)j0 * n + i1];
        const float4 c = lut[(size_t)j1 * n + i0], d = lut[(size_t)j1 * n + i1];
        const uint32_t R8 = unorm8((a.x * gx + b.x * fx) * gy + (c.x * gx + d.x * fx) * fy);
        const uint32_t G8 = unorm8((a.y * gx + b.y * fx) * gy + (c.y * gx + d.y * fx) * fy);
        const uint32_t B8 = unorm8((a.z * gx + b.z * fx) * gy + (c.z * gx + d.z * fx) * fy);
        const uint32_t A8 = unorm8((a.w * gx + b.w * fx) * gy + (c.w * gx + d.w * fx) * fy);
        out[p] = R8 | (G8 << 8) | (B8 << 16) | (A8 << 24);
    }
}

static inline unsigned grid_for(int64_t npix, int cu) {
    int64_t b = (npix + 255) / 256;
    const int64_t cap = (int64_t)cu * 8;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

int launch_colormap_scalar(tsp_context *ctx, const float *d_img, int64_t npix, int C, const float *d_lut, int n_lut,
                           float vmin, float vmax, int log_scale, int weighted, uint8_t *d_out) {
    hipLaunchKernelGGL(colormap_scalar_kernel, dim3(grid_for(npix, ctx->cu_count)), dim3(256), 0, ctx->stream, d_img,
                       npix, C, reinterpret_cast<const float4 *>(d_lut), n_lut, vmin, vmax, log_scale, weighted,
                       reinterpret_cast<uint32_t *>(d_out));
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

int launch_colormap_bivariate(tsp_context *ctx, const float *d_img, int64_t npix, int C, float vmin, float vmax, float dvmin,
                             float dvmax, int log_scale, int weighted, uint8_t *d_out) {
    hipLaunchKernelGGL(colormap_bivariate_kernel, dim3(grid_for(npix, ctx->cu_count)), dim3(256), 0, ctx->stream, d_img, npix,
                       C, reinterpret_cast<const float4 *>(ctx->lut2d), ctx->lut2d_n, vmin, vmax, dvmin, dvmax, log_scale,
                       weighted, reinterpret_cast<uint32_t *>(d_out));
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

int launch_colormap_rgb(tsp_context *ctx, const float *d_img, int64_t npix, int C, float vmin, float vmax, float gamma,
                        uint8_t *d_out8, float *d_outf) {
    hipLaunchKernelGGL(colormap_rgb_kernel, dim3(grid_for(npix, ctx->cu_count)), dim3(256), 0, ctx->stream, d_img, npix,
                       C, vmin, vmax, gamma, reinterpret_cast<uint32_t *>(d_out8), reinterpret_cast<float4 *>(d_outf));
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

}  // namespace tsp
