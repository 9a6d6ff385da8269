// tsp_splat_generic.hip -- the generic splat kernel: any particle order, any ranges, global
// atomics only (float32 terms added into the float64 master image).  It is the cross-check pipeline (TSP_PIPE_GENERIC) and the fallback the
// three-class pipeline is validated against; arithmetic is the same canonical order (tsp_math.h).
//
// Reference semantics: vertex_weighting / vertex_depth / vertex_rgb + fragment_weighting /
// fragment_rgb with additive blending (src/topsy/shaders/sph.wgsl:54-91,139-165; blend
// src/topsy/sph.py:31-42).
//
// Mapping: one lane per particle for footprints <= 16 px; larger footprints are broadcast
// (ballot + shuffle) and rasterised by all 64 lanes of the wave, 64 pixels per step.
#include "tsp_internal.h"

namespace tsp {

__device__ __forceinline__ void atomic_add_f32(double *addr, float v) {   // float32 term into the float64 master image
    __hip_atomic_fetch_add(addr, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ranges layout on device: [0,n) starts, [n,2n) lens, [2n,3n+1) prefix of lens
__device__ __forceinline__ int64_t work_to_particle(const int64_t *ranges, int n_ranges, int64_t w) {
    if (n_ranges == 1) return ranges[0] + w;
    const int64_t *prefix = ranges + 2 * n_ranges;
    int lo = 0, hi = n_ranges - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= w) lo = mid; else hi = mid - 1;
    }
    return ranges[lo] + (w - prefix[lo]);
}

template <int MODE>
__global__ __launch_bounds__(256) void splat_generic_kernel(Particles p, const int64_t *ranges, int n_ranges,
                                                            int64_t total, Camera cam, const float *mips_g,
                                                            double *img, Counters *cnt, int count_frag, int rule) {
    __shared__ float T[MIP_TOTAL];
    for (int i = threadIdx.x; i < MIP_TOTAL; i += 256) T[i] = mips_g[i];
    __syncthreads();
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int R = cam.R;
    unsigned long long nfrag = 0, nculled = 0;

    for (int64_t base = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); base < total;
         base += (int64_t)gridDim.x * 256) {
        const int64_t w = base + lane;
        Proj pr = {};
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        int ilo = 0, ihi = -1, jlo = 0, jhi = -1;
        bool active = false;
        if (w < total) {
            const int64_t i = work_to_particle(ranges, n_ranges, w);
            const float h = p.h[i];
            pr = project(cam, p.x[i], p.y[i], p.z[i], h);
            if (pr.keep) {
                cover_range(pr.pcx, pr.half, R, ilo, ihi);
                cover_range(pr.pcy, pr.half, R, jlo, jhi);
                active = (ilo <= ihi) && (jlo <= jhi);
            }
            if (active) {
                const float hh = h * h;
                if (MODE == TSP_MODE_RGB) {
                    w0 = p.r[i] / hh; w1 = p.g[i] / hh; w2 = p.b[i] / hh;
                } else {
                    w0 = p.m[i] / hh;
                    w1 = (MODE == TSP_MODE_DEPTH) ? pr.cz : (p.q ? p.q[i] : 0.0f);
                }
            } else {
                ++nculled;
            }
        }
        const int nx = ihi - ilo + 1;
        const int npx = active ? nx * (jhi - jlo + 1) : 0;
        const int lvl = level_for(pr.P);
        const bool small = active && npx <= 16;
        if (small) {
            for (int j = jlo; j <= jhi; ++j) {
                const float dy = ((float)j + 0.5f) - pr.pcy;
                for (int i = ilo; i <= ihi; ++i) {
                    const float dx = ((float)i + 0.5f) - pr.pcx;
                    const float k = rule ? sample_kernel_rule(T, pr, dx, dy, rule) : sample_kernel(T, pr, lvl, dx, dy);
                    double *px = img + ((size_t)j * R + i) * C;
                    if (MODE == TSP_MODE_RGB) {
                        atomic_add_f32(px + 0, k * w0);
                        atomic_add_f32(px + 1, k * w1);
                        atomic_add_f32(px + 2, k * w2);
                        atomic_add_f32(px + 3, 1.0f);
                    } else {
                        const float val = k * w0;
                        atomic_add_f32(px + 0, val);
                        atomic_add_f32(px + 1, val * w1);
                    }
                }
            }
            nfrag += npx;
        }
        unsigned long long big = __ballot(active && !small);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            Proj q;
            q.pcx = __shfl(pr.pcx, src); q.pcy = __shfl(pr.pcy, src);
            q.P = __shfl(pr.P, src); q.half = __shfl(pr.half, src); q.invP = __shfl(pr.invP, src);
            const float a0 = __shfl(w0, src), a1 = __shfl(w1, src), a2 = __shfl(w2, src);
            const int bi = __shfl(ilo, src), bj = __shfl(jlo, src), bnx = __shfl(nx, src);
            const int bn = __shfl(npx, src);
            const int blvl = level_for(q.P);
            for (int idx = lane; idx < bn; idx += 64) {
                const int jj = idx / bnx;
                const int j = bj + jj, i = bi + (idx - jj * bnx);
                const float dy = ((float)j + 0.5f) - q.pcy;
                const float dx = ((float)i + 0.5f) - q.pcx;
                const float k = rule ? sample_kernel_rule(T, q, dx, dy, rule) : sample_kernel(T, q, blvl, dx, dy);
                double *px = img + ((size_t)j * R + i) * C;
                if (MODE == TSP_MODE_RGB) {
                    atomic_add_f32(px + 0, k * a0);
                    atomic_add_f32(px + 1, k * a1);
                    atomic_add_f32(px + 2, k * a2);
                    atomic_add_f32(px + 3, 1.0f);
                } else {
                    const float val = k * a0;
                    atomic_add_f32(px + 0, val);
                    atomic_add_f32(px + 1, val * a1);
                }
            }
            if (lane == 0) nfrag += bn;
        }
    }
    if (count_frag) {
        if (nfrag) atomicAdd(&cnt->n_fragments, nfrag);
    }
    if (nculled) atomicAdd(&cnt->n_culled, nculled);
}

int launch_generic(tsp_context *ctx, const Camera &cam, const int64_t *d_ranges, int n_ranges, int64_t total,
                   int mode, int rule) {
    if (total <= 0) return TSP_OK;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)ctx->cu_count * 8;
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), block(256);
    const int cf = ctx->count_fragments ? 1 : 0;
    Particles parts = ctx->p;
    if (!ctx->use_quantity) parts.q = nullptr;
    switch (mode) {
        case TSP_MODE_WEIGHTED:
            hipLaunchKernelGGL(splat_generic_kernel<TSP_MODE_WEIGHTED>, grid, block, 0, ctx->stream, parts, d_ranges,
                               n_ranges, total, cam, ctx->mips, ctx->image64, ctx->counters, cf, rule);
            break;
        case TSP_MODE_DEPTH:
            hipLaunchKernelGGL(splat_generic_kernel<TSP_MODE_DEPTH>, grid, block, 0, ctx->stream, parts, d_ranges,
                               n_ranges, total, cam, ctx->mips, ctx->image64, ctx->counters, cf, rule);
            break;
        case TSP_MODE_RGB:
            hipLaunchKernelGGL(splat_generic_kernel<TSP_MODE_RGB>, grid, block, 0, ctx->stream, parts, d_ranges,
                               n_ranges, total, cam, ctx->mips, ctx->image64, ctx->counters, cf, rule);
            break;
        default:
            set_error("bad mode %d", mode);
            return TSP_EINVAL;
    }
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

}  // namespace tsp
