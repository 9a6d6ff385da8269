// tsp_postpass.hip -- image-space post-passes on the float32 render target.
//
// Periodic tiling: out(x) = sum_k w_k * src(x - o_k), the reference's PeriodicSPH accumulation overlay
// (src/topsy/periodic_sph.py:36-88, shaders/overlay.wgsl:18-51; sampler mag/min linear,
// src/topsy/overlay.py:65-68).  One output pixel per lane; instances are applied in sequence with
// float32 accumulation (the canonical order of oracle_np.periodic_tile, so results are bit-identical).
#include <algorithm>
#include <vector>

#include "tsp_internal.h"

namespace tsp {

struct AxisTap {      // bilinear taps of one output row/column for one instance
    int i0, i1;
    float f;
    bool inside;
};

__device__ __forceinline__ AxisTap tile_axis(float centre, float shift, int R) {
    AxisTap a;
    const float s = centre - shift;                   // source coordinate of this pixel centre
    a.inside = (s >= 0.0f) && (s < (float)R);
    const float t = s - 0.5f;
    const float t0 = __builtin_floorf(t);
    a.f = t - t0;
    a.i0 = clampi((int)t0, 0, R - 1);
    a.i1 = clampi((int)t0 + 1, 0, R - 1);
    return a;
}

template <int C>
__global__ __launch_bounds__(256) void tile_periodic_kernel(const float *__restrict__ src, float *__restrict__ dst, int R, int n,
                                                            const float *__restrict__ offsets, const float *__restrict__ weights) {
    const float halfR = 0.5f * (float)R;
    const int64_t npix = (int64_t)R * R;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const int j = (int)(p / R), i = (int)(p % R);
        const float cx = (float)i + 0.5f, cy = (float)j + 0.5f;
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0f;
        for (int k = 0; k < n; ++k) {
            const AxisTap ax = tile_axis(cx, offsets[2 * k] * halfR, R);            // +x clip -> +column
            const AxisTap ay = tile_axis(cy, -(offsets[2 * k + 1] * halfR), R);     // +y clip -> -row
            if (!(ax.inside && ay.inside)) continue;
            const float w = weights[k];
            const float gx = 1.0f - ax.f, gy = 1.0f - ay.f;
            const float *r0 = src + (size_t)ay.i0 * R * C, *r1 = src + (size_t)ay.i1 * R * C;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float top = r0[ax.i0 * C + c] * gx + r0[ax.i1 * C + c] * ax.f;
                const float bot = r1[ax.i0 * C + c] * gx + r1[ax.i1 * C + c] * ax.f;
                acc[c] = acc[c] + (top * gy + bot * ay.f) * w;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dst[p * C + c] = acc[c];
    }
}

int tile_periodic(tsp_context *ctx, int n, const float *h_offsets, const float *h_weights) {
    const size_t img_bytes = (size_t)ctx->R * ctx->R * ctx->C * sizeof(float);
    const size_t tab_bytes = (size_t)n * 3 * sizeof(float);
    if (ctx->scratch_bytes < img_bytes + tab_bytes + 256) {
        if (ctx->scratch) TSP_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        TSP_HIP(hipMalloc(&ctx->scratch, img_bytes + tab_bytes + 256));
        ctx->scratch_bytes = img_bytes + tab_bytes + 256;
    }
    float *d_out = (float *)ctx->scratch;
    float *d_off = (float *)((char *)ctx->scratch + ((img_bytes + 255) & ~(size_t)255));
    float *d_w = d_off + 2 * n;
    TSP_HIP(hipMemcpyAsync(d_off, h_offsets, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TSP_HIP(hipMemcpyAsync(d_w, h_weights, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = (unsigned)std::min<int64_t>(((int64_t)ctx->R * ctx->R + 255) / 256, (int64_t)ctx->cu_count * 16);
    if (ctx->C == 2)
        hipLaunchKernelGGL(tile_periodic_kernel<2>, dim3(grid), dim3(256), 0, ctx->stream, ctx->image, d_out, ctx->R, n, d_off, d_w);
    else
        hipLaunchKernelGGL(tile_periodic_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, ctx->image, d_out, ctx->R, n, d_off, d_w);
    TSP_HIP(hipGetLastError());
    TSP_HIP(hipMemcpyAsync(ctx->image, d_out, img_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

}  // namespace tsp
