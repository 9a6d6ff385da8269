// tsp_data.hip -- load-time data services (none of this runs per frame):
//   * on-device synthetic snapshot with the distribution of topsy.loader.TestDataLoader
//     (reference src/topsy/loader.py:241-332) from a counter-based generator;
//   * load-time spatial ordering (stratified Morton), the analogue of the reference's cell sort
//     at load (src/topsy/loader.py:88-97, src/topsy/cell_layout.py:63-113);
//   * the float4 read-sum microbenchmark that measures the HBM streaming-read peak.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "tsp_internal.h"

namespace tsp {

// ------------------------------------------------------------------------------------------------
// synthetic snapshot
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ double u01(uint64_t r) {  // (0, 1]
    return ((double)(r >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}

struct SynthParams {
    int64_t n_total, first, count;
    uint64_t seed, mul, add;    // index bijection j = (mul * i + add) mod n_total
    int64_t c0, c1;             // component boundaries: [0,c0) comp 0, [c0,c0+c1) comp 1, rest comp 2
    float h_cap;
};

__global__ __launch_bounds__(256) void synth_kernel(SynthParams sp, float *x, float *y, float *z, float *h, float *m,
                                                    float *q, float *r, float *g, float *b) {
    // TestDataLoader: weights (0.5, 0.4, 0.1), means, stds (loader.py:245-247)
    const double W[3] = {0.5, 0.4, 0.1};
    const double MU[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {6.0, 10.0, 0.0}};
    const double SD[3][3] = {{20.0, 20.0, 20.0}, {4.0, 0.2, 4.0}, {2.0, 2.0, 3.0}};
    const double TWO_PI = 6.283185307179586;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < sp.count; t += (int64_t)gridDim.x * 256) {
        const uint64_t i = (uint64_t)(sp.first + t);
        const uint64_t j = (uint64_t)(((unsigned __int128)sp.mul * i + sp.add) % (uint64_t)sp.n_total);
        const int comp = (int64_t)j < sp.c0 ? 0 : ((int64_t)j < sp.c0 + sp.c1 ? 1 : 2);
        // 3 standard normals from 4 uniforms (Box-Muller), keyed by (seed, j)
        const uint64_t k0 = splitmix64(sp.seed ^ (j * 4 + 0)), k1 = splitmix64(sp.seed ^ (j * 4 + 1));
        const uint64_t k2 = splitmix64(sp.seed ^ (j * 4 + 2)), k3 = splitmix64(sp.seed ^ (j * 4 + 3));
        const double r0 = sqrt(-2.0 * log(u01(k0))), r1 = sqrt(-2.0 * log(u01(k2)));
        const double a0 = TWO_PI * u01(k1), a1 = TWO_PI * u01(k3);
        const double n0 = r0 * cos(a0), n1 = r0 * sin(a0), n2 = r1 * cos(a1);
        // pos = normal.astype(f32) * std + mean, stored as float32 (loader.py:283-287)
        const float px = (float)((double)(float)n0 * SD[comp][0] + MU[comp][0]);
        const float py = (float)((double)(float)n1 * SD[comp][1] + MU[comp][1]);
        const float pz = (float)((double)(float)n2 * SD[comp][2] + MU[comp][2]);
        // density (loader.py:265-272; note: no 1/2 in the exponent), h = 2 / den^0.333333 (:294-296)
        double den = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double dx = (double)px - MU[c][0], dy = (double)py - MU[c][1], dz = (double)pz - MU[c][2];
            const double e = dx * dx / (SD[c][0] * SD[c][0]) + dy * dy / (SD[c][1] * SD[c][1]) +
                             dz * dz / (SD[c][2] * SD[c][2]);
            den += W[c] * exp(-e) / (15.749609945722419 /* (2 pi)^1.5 */ * SD[c][0] * SD[c][1] * SD[c][2]);
        }
        den *= (double)sp.n_total;
        float hh = (float)(2.0 / pow(den, 0.333333));
        if (sp.h_cap > 0.0f && !(hh < sp.h_cap)) hh = sp.h_cap;
        x[t] = px; y[t] = py; z[t] = pz; h[t] = hh;
        m[t] = 1e-8f;                                                    // loader.py:298-299
        if (q) q[t] = sinf(px) * cosf(py) * cosf(pz) * 1e-4f;            // loader.py:301-303
        if (r) {                                                         // loader.py:327-332
            r[t] = fabsf(sinf(px / 10.0f));
            g[t] = fabsf(cosf(py / 10.0f));
            b[t] = fabsf(cosf(pz / 10.0f));
        }
    }
}

static uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b) { uint64_t t = a % b; a = b; b = t; }
    return a;
}

int generate_synthetic(tsp_context *ctx, int64_t n_total, int64_t first, int64_t count, uint64_t seed, float h_cap,
                       int with_quantity, int with_rgb) {
    Particles &p = ctx->p;
    p.n = count;
    ctx->ws.bounds_valid = false;
    if (count == 0) return TSP_OK;
    float **need[] = {&p.x, &p.y, &p.z, &p.h, &p.m};
    for (float **a : need) TSP_HIP(hipMalloc((void **)a, (size_t)count * sizeof(float)));
    if (with_quantity) TSP_HIP(hipMalloc((void **)&p.q, (size_t)count * sizeof(float)));
    if (with_rgb) {
        TSP_HIP(hipMalloc((void **)&p.r, (size_t)count * sizeof(float)));
        TSP_HIP(hipMalloc((void **)&p.g, (size_t)count * sizeof(float)));
        TSP_HIP(hipMalloc((void **)&p.b, (size_t)count * sizeof(float)));
    }
    SynthParams sp;
    sp.n_total = n_total; sp.first = first; sp.count = count; sp.seed = seed; sp.h_cap = h_cap;
    sp.c0 = (int64_t)((double)n_total * 0.5);   // int(N * w), loader.py:281
    sp.c1 = (int64_t)((double)n_total * 0.4);
    uint64_t mul = 1;
    if (n_total > 2) {
        mul = ((uint64_t)((double)n_total * 0.6180339887498949)) | 1ull;
        while (gcd64(mul, (uint64_t)n_total) != 1) mul += 2;
        mul %= (uint64_t)n_total;
        if (mul == 0) mul = 1;
    }
    sp.mul = mul;
    sp.add = splitmix64(seed) % (uint64_t)n_total;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, sp, p.x, p.y, p.z, p.h, p.m, p.q,
                       p.r, p.g, p.b);
    TSP_HIP(hipGetLastError());
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

// ------------------------------------------------------------------------------------------------
// bounds of every block of BOUNDS_BLOCK consecutive particles (kernel S skips chunks that cannot reach the view)
// ------------------------------------------------------------------------------------------------
// One wave per block, eight particles per lane.  fminf / fmaxf drop NaN operands: a particle with a NaN coordinate draws
// nothing and does not widen the box; infinite coordinates or smoothing lengths make the box infinite (never culled).
__global__ __launch_bounds__(64) void block_bounds_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                                          const float *__restrict__ h, int64_t n, float4 *__restrict__ bounds) {
    const int64_t b = blockIdx.x;
    const int64_t i0 = b * BOUNDS_BLOCK;
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}, hm = -inf;
    for (int k = 0; k < BOUNDS_BLOCK / 64; ++k) {
        const int64_t i = i0 + k * 64 + threadIdx.x;
        if (i < n) {
            const float v[3] = {x[i], y[i], z[i]};
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], v[a]); hi[a] = fmaxf(hi[a], v[a]); }
            hm = fmaxf(hm, h[i]);
        }
    }
    for (int o = 32; o; o >>= 1) {
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
        hm = fmaxf(hm, __shfl_xor(hm, o));
    }
    if (threadIdx.x == 0) {
        bounds[2 * b] = make_float4(lo[0], lo[1], lo[2], hm);
        bounds[2 * b + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
}

int ensure_block_bounds(tsp_context *ctx) {
    Workspace &ws = ctx->ws;
    if (ws.bounds_valid) return TSP_OK;
    const int64_t n = ctx->p.n, blocks = (n + BOUNDS_BLOCK - 1) / BOUNDS_BLOCK;
    if (blocks == 0) return TSP_OK;
    if (ws.bounds_capacity < blocks) {
        if (ws.block_bounds) TSP_HIP(hipFree(ws.block_bounds));
        ws.block_bounds = nullptr;
        ws.bounds_capacity = blocks;
        TSP_HIP(hipMalloc((void **)&ws.block_bounds, (size_t)blocks * 2 * sizeof(float4)));
    }
    hipLaunchKernelGGL(block_bounds_kernel, dim3((unsigned)blocks), dim3(64), 0, ctx->stream, ctx->p.x, ctx->p.y, ctx->p.z, ctx->p.h, n,
                       ws.block_bounds);
    TSP_HIP(hipGetLastError());
    ws.bounds_valid = true;
    return TSP_OK;
}

// ------------------------------------------------------------------------------------------------
// load-time spatial ordering: key = (stratum << 48) | morton48(x, y, z)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ordered_u32(float f) {   // monotone float -> uint map
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float unordered_f32(unsigned u) {
    const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

__global__ void bbox_kernel(const float *x, const float *y, const float *z, int64_t n, unsigned *mm /*6*/) {
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v[3] = {x[i], y[i], z[i]};
        for (int k = 0; k < 3; ++k) {
            if (v[k] != v[k] || __builtin_fabsf(v[k]) == __builtin_inff()) continue;
            const unsigned o = ordered_u32(v[k]);
            lo[k] = min(lo[k], o);
            hi[k] = max(hi[k], o);
        }
    }
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off; off >>= 1) {
            lo[k] = min(lo[k], (unsigned)__shfl_xor((int)lo[k], off));
            hi[k] = max(hi[k], (unsigned)__shfl_xor((int)hi[k], off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&mm[k], lo[k]);
            atomicMax(&mm[3 + k], hi[k]);
        }
    }
}

__device__ __forceinline__ uint64_t part1by2(uint64_t x) {   // classic 21-bit spreader, used for 16 bits
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void morton_key_kernel(const float *x, const float *y, const float *z, int64_t n, float3 lo, float3 inv,
                                  int n_strata, uint64_t seed, uint64_t *keys, uint32_t *vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float fx = (x[i] - lo.x) * inv.x, fy = (y[i] - lo.y) * inv.y, fz = (z[i] - lo.z) * inv.z;
        fx = fx != fx ? 0.f : fminf(fmaxf(fx, 0.f), 65535.f);
        fy = fy != fy ? 0.f : fminf(fmaxf(fy, 0.f), 65535.f);
        fz = fz != fz ? 0.f : fminf(fmaxf(fz, 0.f), 65535.f);
        const uint64_t mk = part1by2((uint64_t)fx) | (part1by2((uint64_t)fy) << 1) | (part1by2((uint64_t)fz) << 2);
        const uint64_t stratum = n_strata > 1 ? splitmix64(seed ^ (uint64_t)i) % (uint64_t)n_strata : 0;
        keys[i] = (stratum << 48) | mk;
        vals[i] = (uint32_t)i;
    }
}

__global__ void gather_f32_kernel(const float *__restrict__ src, const uint32_t *__restrict__ idx,
                                  float *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[idx[i]];
}
__global__ void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                  uint32_t *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[idx[i]];
}

// offsets[e] = first index whose key prefix (key >> shift) is >= e, for e = 0 .. n_entries - 1 (offsets[n_entries - 1] = n when
// the last entry is one past the largest prefix).  shift = 48: prefix = stratum; shift = 48 - 3 k: prefix = (stratum, cell
// code of a (2^k)^3 grid) -- the leading 3 k bits of a Morton key are the key of the particle's cell.
__global__ void key_prefix_offsets_kernel(const uint64_t *__restrict__ keys, int64_t n, int64_t n_entries, int shift, int64_t *__restrict__ offsets) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> shift) < e) lo = mid + 1; else hi = mid;
    }
    offsets[e] = lo;
}

// ------------------------------------------------------------------------------------------------
// vertex weights, once per upload
// ------------------------------------------------------------------------------------------------
// vertex_weighting / vertex_rgb (sph.wgsl:69-83) divide the mass (the three band masses) by h*h for every vertex of every
// frame; the quotient does not depend on the camera.  Same float32 operations (one multiply, one IEEE division), once.
__global__ __launch_bounds__(256) void weights_kernel(const float *__restrict__ h, const float *__restrict__ a, const float *__restrict__ b,
                                                      const float *__restrict__ c, int64_t n, float *__restrict__ wa,
                                                      float *__restrict__ wb, float *__restrict__ wc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float hh = h[i] * h[i];
        wa[i] = a[i] / hh;
        if (b) { wb[i] = b[i] / hh; wc[i] = c[i] / hh; }
    }
}

int ensure_weights(tsp_context *ctx, bool rgb) {
    Particles &p = ctx->p;
    if (p.n == 0 || (rgb ? p.wrgb_valid : p.wm_valid)) return TSP_OK;
    int rc;
    const unsigned grid = (unsigned)std::min<int64_t>((p.n + 255) / 256, (int64_t)ctx->cu_count * 32);
    if (rgb) {
        if ((rc = ensure_array(&p.wr, p.n)) || (rc = ensure_array(&p.wg, p.n)) || (rc = ensure_array(&p.wb, p.n))) return rc;
        hipLaunchKernelGGL(weights_kernel, dim3(grid), dim3(256), 0, ctx->stream, p.h, p.r, p.g, p.b, p.n, p.wr, p.wg, p.wb);
        p.wrgb_valid = true;
    } else {
        if ((rc = ensure_array(&p.wm, p.n))) return rc;
        hipLaunchKernelGGL(weights_kernel, dim3(grid), dim3(256), 0, ctx->stream, p.h, p.m, (const float *)nullptr, (const float *)nullptr, p.n,
                           p.wm, (float *)nullptr, (float *)nullptr);
        p.wm_valid = true;
    }
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// order_out[a + t(r, L)] = order_in[a + r] for every segment [a, a + L) = (aligned 512-block) x (cell run): t transposes the
// segment's ranks 8-way, r -> (r mod 8) * ceil-ish(L / 8) + r / 8 (rows r mod 8 < L mod 8 hold one element more): a bijection
__global__ __launch_bounds__(256) void interleave_order_kernel(const uint32_t *__restrict__ order_in, uint32_t *__restrict__ order_out,
                                                               const uint64_t *__restrict__ sorted_keys, const int64_t *__restrict__ cell_start,
                                                               int shift, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = (int64_t)(sorted_keys[i] >> shift);
        const int64_t b0 = i & ~(int64_t)(BOUNDS_BLOCK - 1);
        const int64_t a = max(b0, cell_start[cell]), e = min(min(b0 + BOUNDS_BLOCK, n), cell_start[cell + 1]);
        const int64_t L = e - a, r = i - a;
        int64_t j = i;
        if (L >= 16 && r >= 0 && r < L) {
            const int64_t q = L >> 3, rem = L & 7, row = r & 7;
            j = a + row * q + min(row, rem) + (r >> 3);
        }
        order_out[j] = order_in[i];
    }
}

// Variant 2 of the in-block arrangement: inside every segment (aligned 512-block x cell run) the particles are ordered by
// DESCENDING smoothing length, so the 64 particles of a kernel-S wave step have nearly the same footprint width at any camera
// (h does not depend on the view): the rasteriser's loops run to the widest footprint of the wave, and a wave of mixed widths
// idles the narrow lanes (measured lane utilisation 48-82 %).  One workgroup per block, bitonic sort of (segment, -h) keys in LDS.
__global__ __launch_bounds__(256) void sort_blocks_by_h_kernel(const uint32_t *__restrict__ order_in, uint32_t *__restrict__ order_out,
                                                               const uint64_t *__restrict__ sorted_keys, const int64_t *__restrict__ cell_start,
                                                               int shift, int64_t n, const float *__restrict__ h_old) {
    __shared__ unsigned long long s_key[BOUNDS_BLOCK];
    __shared__ uint32_t s_val[BOUNDS_BLOCK];
    const int64_t b0 = (int64_t)blockIdx.x * BOUNDS_BLOCK;
    for (int t = threadIdx.x; t < BOUNDS_BLOCK; t += 256) {
        const int64_t i = b0 + t;
        unsigned long long key = ~0ull;
        uint32_t val = 0;
        if (i < n) {
            val = order_in[i];
            const int64_t cell = (int64_t)(sorted_keys[i] >> shift);
            const int64_t a = max(b0, cell_start[cell]);
            const uint32_t hb = __float_as_uint(h_old[val]);                 // h > 0: the bit pattern orders like the value
            key = ((unsigned long long)(a - b0) << 32) | (unsigned long long)(0xFFFFFFFFu - hb);
        }
        s_key[t] = key; s_val[t] = val;
    }
    __syncthreads();
    for (int k = 2; k <= BOUNDS_BLOCK; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < BOUNDS_BLOCK; t += 256) {
                const int p = t ^ j;
                if (p > t) {
                    const bool up = (t & k) == 0;
                    const unsigned long long ka = s_key[t], kb = s_key[p];
                    if ((ka > kb) == up) {
                        s_key[t] = kb; s_key[p] = ka;
                        const uint32_t va = s_val[t]; s_val[t] = s_val[p]; s_val[p] = va;
                    }
                }
            }
            __syncthreads();
        }
    for (int t = threadIdx.x; t < BOUNDS_BLOCK; t += 256)
        if (b0 + t < n) order_out[b0 + t] = s_val[t];
}

int reorder_spatial(tsp_context *ctx, int n_strata, uint64_t seed, int64_t *perm_out) {
    Particles &p = ctx->p;
    const int64_t n = p.n;
    ctx->ws.bounds_valid = false;
    p.wm_valid = p.wrgb_valid = false;       // (recomputed in the new order by the next render)
    hipStream_t st = ctx->stream;
    float lo[3], inv[3];
    {   // bounding box of the positions
        DeviceScratch mm;
        TSP_HIP(mm.alloc(6 * sizeof(unsigned)));
        const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0};
        TSP_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(bbox_kernel, dim3(1024), dim3(256), 0, st, p.x, p.y, p.z, n, mm.as<unsigned>());
        unsigned hmm[6];
        TSP_HIP(hipMemcpyAsync(hmm, mm.p, sizeof(hmm), hipMemcpyDeviceToHost, st));
        TSP_HIP(hipStreamSynchronize(st));
        for (int k = 0; k < 3; ++k) {
            const float a = unordered_f32(hmm[k]), b = unordered_f32(hmm[3 + k]);
            lo[k] = a;
            inv[k] = (b > a) ? 65535.0f / (b - a) : 0.0f;
        }
    }
    DeviceScratch order;       // order[new] = old index (relative to the current order)
    {
        DeviceScratch keys, keys2, vals, tmp;
        TSP_HIP(keys.alloc((size_t)n * 8));
        TSP_HIP(keys2.alloc((size_t)n * 8));
        TSP_HIP(vals.alloc((size_t)n * 4));
        TSP_HIP(order.alloc((size_t)n * 4));
        hipLaunchKernelGGL(morton_key_kernel, dim3(4096), dim3(256), 0, st, p.x, p.y, p.z, n,
                           make_float3(lo[0], lo[1], lo[2]), make_float3(inv[0], inv[1], inv[2]), n_strata, seed,
                           keys.as<uint64_t>(), vals.as<uint32_t>());
        TSP_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        TSP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(),
                                                   vals.as<uint32_t>(), order.as<uint32_t>(), n, 0, 60, st));
        TSP_HIP(tmp.alloc(tmp_bytes));
        TSP_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(),
                                                   vals.as<uint32_t>(), order.as<uint32_t>(), n, 0, 60, st));
        // stratum boundaries in the new order (keys2 holds the sorted keys)
        DeviceScratch d_off;
        TSP_HIP(d_off.alloc((size_t)(n_strata + 1) * sizeof(int64_t)));
        hipLaunchKernelGGL(key_prefix_offsets_kernel, dim3((n_strata + 256) / 256), dim3(256), 0, st, keys2.as<uint64_t>(), n,
                           (int64_t)n_strata + 1, 48, d_off.as<int64_t>());
        TSP_HIP(hipGetLastError());
        ctx->strata_offsets.assign((size_t)n_strata + 1, 0);
        TSP_HIP(hipMemcpyAsync(ctx->strata_offsets.data(), d_off.p, (size_t)(n_strata + 1) * sizeof(int64_t),
                               hipMemcpyDeviceToHost, st));
        // Cells for view culling (the role of the reference's CellLayout, src/topsy/cell_layout.py): inside a stratum the
        // Morton order stores every cell of a (2^k)^3 grid over the bounding box as ONE contiguous run; k <= 4 (the
        // reference's 16^3 cells, config.py:27), fewer for small snapshots (>= 16 particles per (stratum, cell) on average).
        // The host merges runs separated by short gaps, so fine cells do not fragment kernel S's 512-particle chunks.
        int k = 0;
        while (k < 4 && n / ((int64_t)n_strata << (3 * (k + 1))) >= 16) ++k;
        const int64_t n_entries = ((int64_t)n_strata << (3 * k)) + 1;
        DeviceScratch d_cell;
        TSP_HIP(d_cell.alloc((size_t)n_entries * sizeof(int64_t)));
        hipLaunchKernelGGL(key_prefix_offsets_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, st, keys2.as<uint64_t>(), n,
                           n_entries, 48 - 3 * k, d_cell.as<int64_t>());
        TSP_HIP(hipGetLastError());
        ctx->cell_offsets.assign((size_t)n_entries, 0);
        TSP_HIP(hipMemcpyAsync(ctx->cell_offsets.data(), d_cell.p, (size_t)n_entries * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        ctx->cell_bits = k;
        for (int a = 0; a < 3; ++a) {
            ctx->cell_lo[a] = lo[a];
            // a cell spans 2^(16 - k) quantisation steps of 1 / inv world units each (inv = 0: a degenerate axis, one cell)
            ctx->cell_width[a] = inv[a] > 0.0f ? (float)(1 << (16 - k)) / inv[a] : 0.0f;
        }
        // Lane decorrelation (round 5): kernel S gives one lane per particle and 64 consecutive particles per wave step.  Morton
        // neighbours are neighbours on screen, so the lanes of a step scatter their footprints into the same few pixels of the LDS
        // window -- same-address ds_add_f64 (a quarter of the LDS pipe's busy cycles at 1e9 particles).  Inside every aligned
        // block of 512 particles (kernel S's chunk) the Morton order is therefore transposed 64 x 8 -> 8 x 64: a wave step
        // then holds every eighth particle of the block -- twice the spread on screen per axis -- while the block, its bounding
        // box and its window stay what they were.  Blocks are cut at cell boundaries, so every (stratum, cell) run keeps
        // exactly its own particles (view culling by cell runs is unaffected).
        if (ctx->reorder_interleave == 2) {
            hipLaunchKernelGGL(sort_blocks_by_h_kernel, dim3((unsigned)((n + BOUNDS_BLOCK - 1) / BOUNDS_BLOCK)), dim3(256), 0, st, order.as<uint32_t>(),
                               vals.as<uint32_t>(), keys2.as<uint64_t>(), d_cell.as<int64_t>(), 48 - 3 * k, n, p.h);
            TSP_HIP(hipGetLastError());
            void *t = order.p; order.p = vals.p; vals.p = t;
        } else if (ctx->reorder_interleave) {
            hipLaunchKernelGGL(interleave_order_kernel, dim3(4096), dim3(256), 0, st, order.as<uint32_t>(), vals.as<uint32_t>(), keys2.as<uint64_t>(),
                               d_cell.as<int64_t>(), 48 - 3 * k, n);
            TSP_HIP(hipGetLastError());
            void *t = order.p; order.p = vals.p; vals.p = t;
        }
        TSP_HIP(hipStreamSynchronize(st));
    }
    // permute every resident attribute through one spare buffer
    DeviceScratch spare;
    TSP_HIP(spare.alloc((size_t)n * 4));
    float **arrs[] = {&p.x, &p.y, &p.z, &p.h, &p.m, &p.q, &p.r, &p.g, &p.b};
    for (float **a : arrs) {
        if (!*a) continue;
        hipLaunchKernelGGL(gather_f32_kernel, dim3(4096), dim3(256), 0, st, *a, order.as<uint32_t>(), spare.as<float>(), n);
        TSP_HIP(hipGetLastError());
        TSP_HIP(hipStreamSynchronize(st));
        float *t = *a; *a = spare.as<float>(); spare.p = t;
    }
    if (p.perm) {   // compose with an earlier reordering: perm_new[i] = perm_old[order[i]]
        hipLaunchKernelGGL(gather_u32_kernel, dim3(4096), dim3(256), 0, st, p.perm, order.as<uint32_t>(), spare.as<uint32_t>(), n);
        TSP_HIP(hipGetLastError());
        TSP_HIP(hipStreamSynchronize(st));
        uint32_t *t = p.perm; p.perm = spare.as<uint32_t>(); spare.p = t;
    } else {
        p.perm = static_cast<uint32_t *>(order.release());
    }
    if (perm_out) {
        std::vector<uint32_t> hp((size_t)n);
        TSP_HIP(hipMemcpy(hp.data(), p.perm, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) perm_out[i] = (int64_t)hp[(size_t)i];
    }
    return TSP_OK;
}

// ------------------------------------------------------------------------------------------------
// on-device autorange: sort the finite content values of the render target
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void content_key_kernel(const float *__restrict__ img, int64_t npix, int C, int kind,
                                                          float scale, uint32_t *__restrict__ keys,
                                                          unsigned long long *counts /* [finite, nonpositive] */) {
    unsigned long long nf = 0, nnp = 0;
    const int per_px = (kind == 2) ? 3 : (kind == 3 ? C : 1);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < npix * per_px; t += (int64_t)gridDim.x * 256) {
        float v;
        if (kind == 2) {
            v = img[(t / 3) * C + (t % 3)] * scale;
        } else if (kind == 3) {
            v = img[t] * scale;                                 // every channel, as vals.ravel() of the raw image
        } else {
            const float a = img[t * C] * scale;                 // get_image(): raw * mass_scale (float32)
            v = (kind == 1) ? (img[t * C + 1] * scale) / a : a; // weighted content: ch1 / ch0
        }
        const bool fin = (v == v) && (__builtin_fabsf(v) != __builtin_inff());
        keys[t] = fin ? ordered_u32(v) : 0xFFFFFFFFu;
        nf += fin;
        nnp += fin && (v <= 0.0f);
    }
    for (int o = 32; o; o >>= 1) {
        nf += __shfl_xor((long long)nf, o);
        nnp += __shfl_xor((long long)nnp, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nf) atomicAdd(&counts[0], nf);
        if (nnp) atomicAdd(&counts[1], nnp);
    }
}

int content_sort(tsp_context *ctx, int kind, float scale, int64_t *n_finite, int64_t *n_nonpositive) {
    const int64_t npix = (int64_t)ctx->R * ctx->R;
    const int64_t n = npix * (kind == 2 ? 3 : (kind == 3 ? ctx->C : 1));
    hipStream_t st = ctx->stream;
    if (ctx->sort_capacity < n) {
        if (ctx->sort_keys) TSP_HIP(hipFree(ctx->sort_keys));
        if (ctx->sort_keys_alt) TSP_HIP(hipFree(ctx->sort_keys_alt));
        if (ctx->sort_tmp) TSP_HIP(hipFree(ctx->sort_tmp));
        ctx->sort_tmp = nullptr;
        TSP_HIP(hipMalloc((void **)&ctx->sort_keys, (size_t)n * 4));
        TSP_HIP(hipMalloc((void **)&ctx->sort_keys_alt, (size_t)n * 4));
        ctx->sort_capacity = n;
        ctx->sort_tmp_bytes = 0;
        TSP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, ctx->sort_tmp_bytes, ctx->sort_keys, ctx->sort_keys_alt, n, 0, 32, st));
        TSP_HIP(hipMalloc(&ctx->sort_tmp, ctx->sort_tmp_bytes ? ctx->sort_tmp_bytes : 16));
    }
    unsigned long long *counts = reinterpret_cast<unsigned long long *>(ctx->counters);   // scratch: reuse the counter block
    TSP_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), st));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8);
    hipLaunchKernelGGL(content_key_kernel, dim3(grid), dim3(256), 0, st, ctx->image, npix, ctx->C, kind, scale, ctx->sort_keys, counts);
    TSP_HIP(hipGetLastError());
    size_t tmp_bytes = ctx->sort_tmp_bytes;
    TSP_HIP(hipcub::DeviceRadixSort::SortKeys(ctx->sort_tmp, tmp_bytes, ctx->sort_keys, ctx->sort_keys_alt, n, 0, 32, st));
    unsigned long long hc[2];
    TSP_HIP(hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, st));
    TSP_HIP(hipStreamSynchronize(st));
    ctx->sorted_count = (int64_t)hc[0];
    *n_finite = (int64_t)hc[0];
    *n_nonpositive = (int64_t)hc[1];
    return TSP_OK;
}

// ------------------------------------------------------------------------------------------------
// HBM streaming-read microbenchmark: float4 read-sum
// ------------------------------------------------------------------------------------------------
// U = independent 16-byte loads in flight per lane; NT = non-temporal loads (the stream is read once)
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_sum_kernel(const float4 *__restrict__ src, int64_t n4, float *sink) {
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) {
                const float *q = reinterpret_cast<const float *>(src + i + u * stride);
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(q));
                v[u] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                v[u] = src[i + u * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < n4; i += stride) {
        const float4 a = src[i];
        acc += a.x + a.y + a.z + a.w;
    }
    if (acc == 123.456f) *sink = acc;   // keep the loads alive
}

// Streaming-read peak of this GPU as this process can reach it: the best of a few launch shapes (loads in flight per
// lane x workgroups per CU x cache policy) over a buffer larger than the 256 MiB Infinity Cache.
int measure_read_bandwidth(tsp_context *ctx, int64_t bytes, int iters, double *gbps_out) {
    bytes &= ~(int64_t)4095;
    if (bytes < 4096) bytes = 4096;
    DeviceScratch buf_s, sink_s;
    TSP_HIP(buf_s.alloc((size_t)bytes));
    TSP_HIP(sink_s.alloc(4));
    float4 *buf = buf_s.as<float4>();
    float *sink = sink_s.as<float>();
    TSP_HIP(hipMemsetAsync(buf, 0x11, (size_t)bytes, ctx->stream));
    double best = 0.0;
    for (int variant = 0; variant < 6; ++variant) {
        const unsigned grid = (unsigned)ctx->cu_count * (variant % 3 == 0 ? 8u : (variant % 3 == 1 ? 16u : 32u));
        auto launch = [&]() {
            if (variant < 3) hipLaunchKernelGGL((read_sum_kernel<4, false>), dim3(grid), dim3(256), 0, ctx->stream, buf, bytes / 16, sink);
            else hipLaunchKernelGGL((read_sum_kernel<8, true>), dim3(grid), dim3(256), 0, ctx->stream, buf, bytes / 16, sink);
        };
        launch();
        TSP_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
        for (int it = 0; it < iters; ++it) launch();
        TSP_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
        TSP_HIP(hipStreamSynchronize(ctx->stream));
        TSP_HIP(hipGetLastError());
        float ms = 0.f;
        TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
        best = std::max(best, (double)bytes * iters / (ms * 1e-3) / 1e9);
    }
    *gbps_out = best;
    return TSP_OK;
}

}  // namespace tsp
