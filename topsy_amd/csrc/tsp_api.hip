// tsp_api.hip -- the extern "C" boundary of libtopsy_splat (see include/topsy_splat.h).
//
// Each entry point cites the reference interface it replaces in the header.  Everything here is
// host-side plumbing: argument validation, HBM allocation, uploads, kernel dispatch, hipEvent
// timing (the TimeGpuOperation hook of reference src/topsy/util.py:76-115) and read-back.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tsp_internal.h"

namespace tsp {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_array(float **p, int64_t n) {
    if (*p) {
        TSP_HIP(hipFree(*p));
        *p = nullptr;
    }
    if (n > 0) TSP_HIP(hipMalloc((void **)p, (size_t)n * sizeof(float)));
    return TSP_OK;
}

static int upload_array(tsp_context *ctx, float **dst, const float *src, int64_t n) {
    if (!*dst) TSP_HIP(hipMalloc((void **)dst, (size_t)n * sizeof(float)));
    TSP_HIP(hipMemcpyAsync(*dst, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return TSP_OK;
}

__global__ void image_to_float_kernel(const double *__restrict__ src, float *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}
__global__ void image_to_double_kernel(const float *__restrict__ src, double *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (double)src[i];
}

int launch_image_convert(tsp_context *ctx, bool to_float) {
    const int64_t n = (int64_t)ctx->R * ctx->R * ctx->C;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8);
    if (to_float) hipLaunchKernelGGL(image_to_float_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->image64, ctx->image, n);
    else hipLaunchKernelGGL(image_to_double_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->image, ctx->image64, n);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

__global__ void gather_kernel(const float *__restrict__ src, const uint32_t *__restrict__ perm, float *__restrict__ dst,
                              int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[perm[i]];
}

// upload in caller order, then apply the load-time permutation if one is active
static int upload_permuted(tsp_context *ctx, float **dst, const float *src, int64_t n) {
    if (!ctx->p.perm) return upload_array(ctx, dst, src, n);
    float *tmp = nullptr;
    TSP_HIP(hipMalloc((void **)&tmp, (size_t)n * sizeof(float)));
    TSP_HIP(hipMemcpyAsync(tmp, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (!*dst) TSP_HIP(hipMalloc((void **)dst, (size_t)n * sizeof(float)));
    hipLaunchKernelGGL(gather_kernel, dim3(2048), dim3(256), 0, ctx->stream, tmp, ctx->p.perm, *dst, n);
    TSP_HIP(hipGetLastError());
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    TSP_HIP(hipFree(tmp));
    return TSP_OK;
}

static void free_particles(tsp_context *ctx) {
    Particles &p = ctx->p;
    float **arrs[] = {&p.x, &p.y, &p.z, &p.h, &p.m, &p.q, &p.r, &p.g, &p.b};
    for (float **a : arrs) {
        if (*a) (void)hipFree(*a);
        *a = nullptr;
    }
    float **derived[] = {&p.wm, &p.wr, &p.wg, &p.wb};
    for (float **a : derived) {
        if (*a) (void)hipFree(*a);
        *a = nullptr;
    }
    p.wm_valid = p.wrgb_valid = false;
    if (p.perm) (void)hipFree(p.perm);
    p.perm = nullptr;
    p.n = 0;
    ctx->ws.bounds_valid = false;
    ctx->strata_offsets.clear();
    ctx->cell_offsets.clear();
    ctx->cell_bits = 0;
}

}  // namespace tsp

using namespace tsp;

extern "C" {

const char *tsp_last_error(void) { return g_err; }
int tsp_version(void) { return 105; }     // 101: tsp_stats gained ms_mega, n_mega (16 bytes); 102: the per-kernel fragment counts (32 bytes); 103: n_chunk_culled (8 bytes); 104: matrix-core / kernel-I options removed; 105: kernel M's options removed (kernel G draws the mid footprints)
int tsp_stats_size(void) { return (int)sizeof(tsp_stats); }

int tsp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        set_error("hipGetDeviceCount failed");
        return TSP_ENODEV;
    }
    return n;
}

static int create_resources(tsp_context *ctx, int device_id, int resolution, int n_channels) {
    hipDeviceProp_t prop;
    TSP_HIP(hipGetDeviceProperties(&prop, device_id));
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    TSP_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    TSP_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    for (auto &e : ctx->ev) TSP_HIP(hipEventCreate(&e));
    const size_t npx = (size_t)resolution * resolution;
    TSP_HIP(hipMalloc((void **)&ctx->image, npx * n_channels * sizeof(float)));
    TSP_HIP(hipMemsetAsync(ctx->image, 0, npx * n_channels * sizeof(float), ctx->stream));
    TSP_HIP(hipMalloc((void **)&ctx->image64, npx * n_channels * sizeof(double)));
    TSP_HIP(hipMemsetAsync(ctx->image64, 0, npx * n_channels * sizeof(double), ctx->stream));
    TSP_HIP(hipMalloc((void **)&ctx->mips, MIP_TOTAL * sizeof(float)));
    TSP_HIP(hipMalloc((void **)&ctx->counters, sizeof(Counters)));
    TSP_HIP(hipMemsetAsync(ctx->counters, 0, sizeof(Counters), ctx->stream));
    TSP_HIP(hipMalloc((void **)&ctx->out8, npx * 4));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_create(int device_id, int resolution, int n_channels, tsp_context **out) {
    TSP_REQUIRE(out != nullptr, TSP_EINVAL, "out is NULL");
    *out = nullptr;
    TSP_REQUIRE(resolution > 0 && resolution <= 16384, TSP_EINVAL, "resolution %d out of range", resolution);
    TSP_REQUIRE(n_channels == 2 || n_channels == 4, TSP_EINVAL, "n_channels must be 2 (SPH) or 4 (RGBSPH), got %d",
                n_channels);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available");
        return TSP_ENODEV;
    }
    TSP_REQUIRE(device_id >= 0 && device_id < ndev, TSP_ENODEV, "device %d not present (%d devices)", device_id, ndev);
    TSP_HIP(hipSetDevice(device_id));
    tsp_context *ctx = new tsp_context();
    ctx->device = device_id;
    ctx->R = resolution;
    ctx->C = n_channels;
    ctx->Ccap = n_channels;
    // a failed allocation (the float64 master image is R*R*C*8 bytes: 8.6 GB at 16384^2 x 4) must not leak the
    // streams, events and buffers created before it: tsp_destroy releases whatever exists (the error text stays)
    const int rc = create_resources(ctx, device_id, resolution, n_channels);
    if (rc != TSP_OK) {
        tsp_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return TSP_OK;
}

void tsp_destroy(tsp_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    tsp_comm_destroy(ctx);
    free_particles(ctx);
    void *ptrs[] = {ctx->image, ctx->image64, ctx->image64_entry, ctx->mips, ctx->counters, ctx->out8, ctx->outf, ctx->lut, ctx->lut2d, ctx->scratch,
                    ctx->ws.mid_geom, ctx->ws.mid_w, ctx->ws.huge_geom, ctx->ws.huge_w, ctx->ws.hband_geom, ctx->ws.hband_w, ctx->ws.hband_count, ctx->ws.mband_geom, ctx->ws.mband_w, ctx->ws.mband_count, ctx->ws.mband_base, ctx->ws.mitem_tile, ctx->ws.mitem_base, 
                    ctx->ws.block_bounds, ctx->ws.alive_list, ctx->ws.cull_info, ctx->ws.range_prefix, ctx->ws.count_diff, ctx->ws.count_band, ctx->sort_keys, ctx->sort_keys_alt, ctx->sort_tmp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int tsp_set_kernel_mips(tsp_context *ctx, const float *lut, int n0, int n_levels) {
    TSP_REQUIRE(ctx && lut, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(n0 == 64 && n_levels == 4, TSP_EINVAL,
                "kernel texture must be 64^2 with 4 mip levels (reference sph.py:396), got n0=%d levels=%d", n0, n_levels);
    TSP_HIP(hipSetDevice(ctx->device));
    TSP_HIP(hipMemcpy(ctx->mips, lut, MIP_TOTAL * sizeof(float), hipMemcpyHostToDevice));
    ctx->have_mips = true;
    // does the kernel vanish outside the inscribed disc, on every mip level?  level l has n = 64 >> l
    // texels per side, texel (j, i) centre = -2 + (k + 0.5) * 4 / n
    bool zero = true;
    for (int l = 0, off = 0; l < 4 && zero; off += (64 >> l) * (64 >> l), ++l) {
        const int n = 64 >> l;
        for (int j = 0; j < n && zero; ++j)
            for (int i = 0; i < n; ++i) {
                const double x = -2.0 + (i + 0.5) * 4.0 / n, y = -2.0 + (j + 0.5) * 4.0 / n;
                if (x * x + y * y >= 4.0 && lut[off + j * n + i] != 0.0f) { zero = false; break; }
            }
    }
    ctx->lut_zero_outside_disc = zero;
    // mirror symmetry, bit for bit (kernel G then keeps one quadrant of every level in LDS)
    bool sym = true;
    for (int l = 0, off = 0; l < 4 && sym; off += (64 >> l) * (64 >> l), ++l) {
        const int n = 64 >> l;
        for (int j = 0; j < n && sym; ++j)
            for (int i = 0; i < n; ++i)
                if (memcmp(&lut[off + j * n + i], &lut[off + j * n + (n - 1 - i)], 4) || memcmp(&lut[off + j * n + i], &lut[off + (n - 1 - j) * n + i], 4)) { sym = false; break; }
    }
    ctx->lut_mirror_symmetric = sym;
    return TSP_OK;
}

int tsp_upload_particles(tsp_context *ctx, int64_t n, const float *x, const float *y, const float *z, const float *h,
                         const float *mass) {
    TSP_REQUIRE(ctx, TSP_EINVAL, "NULL context");
    TSP_REQUIRE(n >= 0 && n < ((int64_t)1 << 32), TSP_EINVAL, "particle count %lld out of range", (long long)n);
    TSP_REQUIRE(n == 0 || (x && y && z && h), TSP_EINVAL, "NULL position/smoothing array");
    TSP_HIP(hipSetDevice(ctx->device));
    free_particles(ctx);
    ctx->p.n = n;
    if (n == 0) return TSP_OK;
    int rc;
    if ((rc = upload_array(ctx, &ctx->p.x, x, n))) return rc;
    if ((rc = upload_array(ctx, &ctx->p.y, y, n))) return rc;
    if ((rc = upload_array(ctx, &ctx->p.z, z, n))) return rc;
    if ((rc = upload_array(ctx, &ctx->p.h, h, n))) return rc;
    if (mass && (rc = upload_array(ctx, &ctx->p.m, mass, n))) return rc;
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_upload_quantity(tsp_context *ctx, const float *q) {
    TSP_REQUIRE(ctx, TSP_EINVAL, "NULL context");
    TSP_HIP(hipSetDevice(ctx->device));
    if (!q) {
        if (ctx->p.q) TSP_HIP(hipFree(ctx->p.q));
        ctx->p.q = nullptr;
        return TSP_OK;
    }
    TSP_REQUIRE(ctx->p.n > 0, TSP_ESTATE, "upload particles before the quantity");
    int rc = upload_permuted(ctx, &ctx->p.q, q, ctx->p.n);
    if (rc) return rc;
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_upload_rgb(tsp_context *ctx, const float *r, const float *g, const float *b) {
    TSP_REQUIRE(ctx && r && g && b, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(ctx->p.n > 0, TSP_ESTATE, "upload particles before rgb");
    TSP_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload_permuted(ctx, &ctx->p.r, r, ctx->p.n))) return rc;
    if ((rc = upload_permuted(ctx, &ctx->p.g, g, ctx->p.n))) return rc;
    if ((rc = upload_permuted(ctx, &ctx->p.b, b, ctx->p.n))) return rc;
    ctx->p.wrgb_valid = false;
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

// one lane per particle; the magnitudes are read in the caller's order (through the load-time permutation, if any)
__global__ __launch_bounds__(256) void band_contraction_kernel(const double *__restrict__ mags, const double *__restrict__ weights,
                                                               int n_bands, int64_t n, const uint32_t *__restrict__ perm,
                                                               float *__restrict__ r, float *__restrict__ g, float *__restrict__ b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = perm ? (int64_t)perm[i] : i;
        double acc[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < n_bands; ++k) {
            const double lum = pow(10.0, -0.4 * mags[(int64_t)k * n + src]);      // _effective_mass_for_band (loader.py:112-113)
            for (int c = 0; c < 3; ++c) {
                const double w = weights[c * n_bands + k];
                if (w != 0.0) acc[c] += w * lum;
            }
        }
        const float fr = (float)acc[0], fg = (float)acc[1], fb = (float)acc[2];
        r[i] = (fr != fr) ? 0.0f : fr;                                            // rgb[np.isnan(rgb)] = 0 (loader.py:120)
        g[i] = (fg != fg) ? 0.0f : fg;
        b[i] = (fb != fb) ? 0.0f : fb;
    }
}

int tsp_upload_band_magnitudes(tsp_context *ctx, int n_bands, const double *mags, const double *weights) {
    TSP_REQUIRE(ctx && mags && weights, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(n_bands >= 1 && n_bands <= 64, TSP_EINVAL, "n_bands %d out of range", n_bands);
    TSP_REQUIRE(ctx->p.n > 0, TSP_ESTATE, "upload particles before the band magnitudes");
    TSP_HIP(hipSetDevice(ctx->device));
    const int64_t n = ctx->p.n;
    DeviceScratch d_mags, d_w;
    TSP_HIP(d_mags.alloc((size_t)n_bands * n * sizeof(double)));
    TSP_HIP(d_w.alloc((size_t)3 * n_bands * sizeof(double)));
    TSP_HIP(hipMemcpyAsync(d_mags.p, mags, (size_t)n_bands * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TSP_HIP(hipMemcpyAsync(d_w.p, weights, (size_t)3 * n_bands * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc;
    if ((rc = ensure_array(&ctx->p.r, n)) || (rc = ensure_array(&ctx->p.g, n)) || (rc = ensure_array(&ctx->p.b, n))) return rc;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 16);
    hipLaunchKernelGGL(band_contraction_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_mags.as<double>(), d_w.as<double>(), n_bands, n,
                       ctx->p.perm, ctx->p.r, ctx->p.g, ctx->p.b);
    ctx->p.wrgb_valid = false;
    TSP_HIP(hipGetLastError());
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_generate_synthetic(tsp_context *ctx, int64_t n_total, int64_t first, int64_t count, uint64_t seed, float h_cap,
                           int with_quantity, int with_rgb) {
    TSP_REQUIRE(ctx, TSP_EINVAL, "NULL context");
    TSP_REQUIRE(n_total > 0 && first >= 0 && count >= 0 && first + count <= n_total, TSP_EINVAL,
                "bad shard [%lld, +%lld) of %lld", (long long)first, (long long)count, (long long)n_total);
    TSP_REQUIRE(count < ((int64_t)1 << 32), TSP_EINVAL, "shard too large");
    TSP_HIP(hipSetDevice(ctx->device));
    free_particles(ctx);
    return generate_synthetic(ctx, n_total, first, count, seed, h_cap, with_quantity, with_rgb);
}

int tsp_get_strata_offsets(tsp_context *ctx, int64_t *offsets_out, int capacity) {
    if (!ctx || !offsets_out || capacity <= 0) return 0;
    const int n = (int)std::min<size_t>(ctx->strata_offsets.size(), (size_t)capacity);
    for (int i = 0; i < n; ++i) offsets_out[i] = ctx->strata_offsets[(size_t)i];
    return n;
}

int tsp_get_cell_layout(tsp_context *ctx, int *n_strata_out, int *cells_per_axis_out, float *box_lo_out, float *cell_width_out) {
    TSP_REQUIRE(ctx && n_strata_out && cells_per_axis_out && box_lo_out && cell_width_out, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(!ctx->cell_offsets.empty(), TSP_ESTATE, "the particles were never reordered (tsp_reorder_spatial)");
    *n_strata_out = (int)ctx->strata_offsets.size() - 1;
    *cells_per_axis_out = 1 << ctx->cell_bits;
    for (int a = 0; a < 3; ++a) {
        box_lo_out[a] = ctx->cell_lo[a];
        cell_width_out[a] = ctx->cell_width[a];
    }
    return TSP_OK;
}

int64_t tsp_get_cell_offsets(tsp_context *ctx, int64_t *offsets_out, int64_t capacity) {
    if (!ctx || !offsets_out || capacity <= 0) return 0;
    const int64_t n = std::min<int64_t>((int64_t)ctx->cell_offsets.size(), capacity);
    for (int64_t i = 0; i < n; ++i) offsets_out[i] = ctx->cell_offsets[(size_t)i];
    return n;
}

int tsp_reorder_spatial(tsp_context *ctx, int n_strata, uint64_t seed, int64_t *perm_out) {
    TSP_REQUIRE(ctx, TSP_EINVAL, "NULL context");
    TSP_REQUIRE(n_strata >= 1 && n_strata <= 4096, TSP_EINVAL, "n_strata %d out of range", n_strata);
    TSP_REQUIRE(ctx->p.n > 0, TSP_ESTATE, "no particles resident");
    TSP_HIP(hipSetDevice(ctx->device));
    return reorder_spatial(ctx, n_strata, seed, perm_out);
}

int64_t tsp_num_particles(tsp_context *ctx) { return ctx ? ctx->p.n : 0; }

int tsp_download_particles(tsp_context *ctx, float *x, float *y, float *z, float *h, float *mass, float *q, float *r,
                           float *g, float *b) {
    TSP_REQUIRE(ctx, TSP_EINVAL, "NULL context");
    TSP_HIP(hipSetDevice(ctx->device));
    const Particles &p = ctx->p;
    struct { float *dst; const float *src; const char *name; } items[] = {
        {x, p.x, "x"}, {y, p.y, "y"}, {z, p.z, "z"}, {h, p.h, "h"}, {mass, p.m, "mass"},
        {q, p.q, "q"}, {r, p.r, "r"}, {g, p.g, "g"}, {b, p.b, "b"}};
    for (auto &it : items) {
        if (!it.dst) continue;
        TSP_REQUIRE(it.src, TSP_ESTATE, "array '%s' is not resident", it.name);
        TSP_HIP(hipMemcpy(it.dst, it.src, (size_t)p.n * sizeof(float), hipMemcpyDeviceToHost));
    }
    return TSP_OK;
}

static int render_block(tsp_context *ctx, const float *M, float scale_factor, const int64_t *starts, const int64_t *lens,
                        int n_ranges, int clear, int mode, int flags, double *gpu_ms_out);

// A block draws whole or not at all (the reference's render loop, sph.py:306-332, cannot fail half-way): the float64 accumulator is
// copied aside on entry (one device-to-device copy of the image: ~10 us at 1024^2) and a block that fails after its first kernel
// has added to it -- an allocation for its record lists or bins, an injected test failure -- puts accumulator, channel layout and
// statistics back as the call found them; the float32 presentation image is only written by the last step of a successful block.
int tsp_render(tsp_context *ctx, const float *M, float scale_factor, const int64_t *starts, const int64_t *lens,
               int n_ranges, int clear, int mode, int flags, double *gpu_ms_out) {
    TSP_REQUIRE(ctx && M, TSP_EINVAL, "NULL argument");
    TSP_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->R * ctx->R * ctx->Ccap * sizeof(double);
    if (!ctx->image64_entry) TSP_HIP(hipMalloc((void **)&ctx->image64_entry, bytes));
    TSP_HIP(hipMemcpyAsync(ctx->image64_entry, ctx->image64, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    const int C_entry = ctx->C;
    const tsp_stats stats_entry = ctx->stats;
    const int64_t culled_entry = ctx->chunk_culled_particles;
    const int rc = render_block(ctx, M, scale_factor, starts, lens, n_ranges, clear, mode, flags, gpu_ms_out);
    if (rc != TSP_OK) {
        const std::string why = tsp_last_error();        // (the restore below must not replace the reason of the failure)
        ctx->C = C_entry; ctx->stats = stats_entry; ctx->chunk_culled_particles = culled_entry;
        const hipError_t e1 = hipMemcpyAsync(ctx->image64, ctx->image64_entry, bytes, hipMemcpyDeviceToDevice, ctx->stream);
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);
        if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
        if (e1 != hipSuccess || e2 != hipSuccess) set_error("%s; and the accumulator could not be restored (%s)", why.c_str(), hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        else set_error("%s", why.c_str());
    }
    return rc;
}

static int render_block(tsp_context *ctx, const float *M, float scale_factor, const int64_t *starts, const int64_t *lens,
                        int n_ranges, int clear, int mode, int flags, double *gpu_ms_out) {
    TSP_REQUIRE(ctx && M, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(ctx->have_mips, TSP_ESTATE, "tsp_set_kernel_mips must be called before tsp_render");
    TSP_REQUIRE(mode == TSP_MODE_WEIGHTED || mode == TSP_MODE_DEPTH || mode == TSP_MODE_RGB, TSP_EINVAL, "bad mode %d",
                mode);
    if (mode == TSP_MODE_RGB) {
        TSP_REQUIRE(ctx->Ccap == 4, TSP_EINVAL, "TSP_MODE_RGB needs a 4-channel context");
        TSP_REQUIRE(ctx->p.n == 0 || (ctx->p.r && ctx->p.g && ctx->p.b), TSP_ESTATE, "rgb arrays not uploaded");
    } else {
        TSP_REQUIRE(ctx->p.n == 0 || ctx->p.m, TSP_ESTATE, "mass array not uploaded");
    }
    // the active channel count follows the mode (a 4-channel context can also hold 2-channel renders);
    // a block that does not clear must continue in the layout of the image it adds to
    const int newC = (mode == TSP_MODE_RGB) ? 4 : 2;
    TSP_REQUIRE(clear || newC == ctx->C, TSP_ESTATE, "cannot accumulate a %d-channel block onto a %d-channel image",
                newC, ctx->C);
    ctx->C = newC;
    TSP_REQUIRE(n_ranges >= 0 && (n_ranges == 0 || (starts && lens) || (!starts && !lens)), TSP_EINVAL, "bad ranges");
    TSP_HIP(hipSetDevice(ctx->device));

    // clip the ranges to [0, n) and drop empties (an indirect draw with instance_count 0 draws nothing)
    std::vector<int64_t> s, l;
    if (!starts) {
        if (ctx->p.n > 0) { s.push_back(0); l.push_back(ctx->p.n); }
    } else {
        for (int i = 0; i < n_ranges; ++i) {
            TSP_REQUIRE(lens[i] >= 0, TSP_EINVAL, "range %d has negative length %lld", i, (long long)lens[i]);
            // clip without forming starts + lens (a caller may pass INT64_MAX for "to the end")
            int64_t b = starts[i], len = lens[i];
            if (b < 0) {
                len = (len > -b) ? len + b : 0;      // b > INT64_MIN + len, no overflow
                b = 0;
            }
            if (b >= ctx->p.n || len == 0) continue;
            if (len > ctx->p.n - b) len = ctx->p.n - b;
            s.push_back(b);
            l.push_back(len);
        }
    }
    const int nr = (int)s.size();
    int64_t total = 0;
    for (int i = 0; i < nr; ++i) total += l[i];

    Camera cam;
    for (int i = 0; i < 12; ++i) cam.m[i] = M[i];
    cam.sf = scale_factor;
    cam.R = ctx->R;
    cam.Rf = (float)ctx->R;
    cam.halfR = 0.5f * cam.Rf;

    TSP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    if (clear)
        TSP_HIP(hipMemsetAsync(ctx->image64, 0, (size_t)ctx->R * ctx->R * ctx->C * sizeof(double), ctx->stream));
    TSP_HIP(hipMemsetAsync(ctx->counters, 0, sizeof(Counters), ctx->stream));
    ctx->stats = tsp_stats{};
    ctx->stats.n_particles = total;
    ctx->chunk_culled_particles = 0;
    int rc = TSP_OK;
    if (total > 0) {
        const int rule = (flags & TSP_SAMPLE_BILINEAR_MIP0) ? 1 : ((flags & TSP_SAMPLE_BILINEAR_MIP) ? 2 : 0);
        if ((flags & TSP_PIPE_GENERIC) || rule != 0) {     // the alternative sampling rules exist in the generic kernel only
            // device copy of the ranges: starts | lens | prefix
            std::vector<int64_t> pack(3 * (size_t)nr + 1);
            int64_t acc = 0;
            for (int i = 0; i < nr; ++i) {
                pack[i] = s[i];
                pack[nr + i] = l[i];
                pack[2 * nr + i] = acc;
                acc += l[i];
            }
            pack[3 * nr] = acc;
            if (ctx->ws.range_capacity < (int64_t)pack.size()) {
                if (ctx->ws.range_prefix) TSP_HIP(hipFree(ctx->ws.range_prefix));
                ctx->ws.range_capacity = (int64_t)pack.size() * 2 + 64;
                TSP_HIP(hipMalloc((void **)&ctx->ws.range_prefix, ctx->ws.range_capacity * sizeof(int64_t)));
            }
            TSP_HIP(hipMemcpyAsync(ctx->ws.range_prefix, pack.data(), pack.size() * sizeof(int64_t),
                                   hipMemcpyHostToDevice, ctx->stream));
            rc = launch_generic(ctx, cam, ctx->ws.range_prefix, nr, total, mode, rule);
            // pack must outlive the async copy
            TSP_HIP(hipStreamSynchronize(ctx->stream));
        } else {
            rc = launch_pipeline(ctx, cam, s.data(), l.data(), nr, total, mode);
        }
    }
    if (rc) return rc;
    if ((rc = launch_image_convert(ctx, true))) return rc;     // round the float64 master image once
    ctx->image_is_reduced = false;                             // `image` is this rank's partial image again
    TSP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->stats.ms_total = ms;
    Counters hc;
    TSP_HIP(hipMemcpy(&hc, ctx->counters, sizeof(hc), hipMemcpyDeviceToHost));
    ctx->stats.n_small = (int64_t)hc.n_small;
    ctx->stats.n_mid = (int64_t)hc.n_mid;
    ctx->stats.n_huge = (int64_t)hc.n_huge;
    ctx->stats.n_mega = 0;
    ctx->stats.n_culled = (int64_t)hc.n_culled + ctx->chunk_culled_particles;
    ctx->stats.n_chunk_culled = ctx->chunk_culled_particles;
    ctx->stats.n_fragments = (int64_t)hc.n_fragments;
    ctx->stats.n_fragments_stream = (int64_t)hc.n_frag_class[0];
    ctx->stats.n_fragments_mid = (int64_t)hc.n_frag_class[1];
    ctx->stats.n_fragments_huge = (int64_t)hc.n_frag_class[2];
    ctx->stats.n_fragments_mega = (int64_t)hc.n_frag_class[3];      // (0 in the product build; the TSP_H2_DEBUG analysis build counts here)
    if (gpu_ms_out) *gpu_ms_out = ms;
    return TSP_OK;
}

int tsp_read_image(tsp_context *ctx, float *out) {
    TSP_REQUIRE(ctx && out, TSP_EINVAL, "NULL argument");
    TSP_HIP(hipSetDevice(ctx->device));
    TSP_HIP(hipMemcpy(out, ctx->image, (size_t)ctx->R * ctx->R * ctx->C * sizeof(float), hipMemcpyDeviceToHost));
    return TSP_OK;
}

int tsp_write_image(tsp_context *ctx, const float *in) {
    TSP_REQUIRE(ctx && in, TSP_EINVAL, "NULL argument");
    TSP_HIP(hipSetDevice(ctx->device));
    TSP_HIP(hipMemcpy(ctx->image, in, (size_t)ctx->R * ctx->R * ctx->C * sizeof(float), hipMemcpyHostToDevice));
    int rc = launch_image_convert(ctx, false);                 // keep the master copy consistent
    if (rc) return rc;
    ctx->image_is_reduced = false;
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

// The cross-shard SUM becomes this context's float32 presentation image (what read-back, colormap and autorange see); the
// float64 accumulator keeps the context's own partial sums, as after an in-place RCCL reduce (tsp_comm_reduce_image)
int tsp_set_reduced_image(tsp_context *ctx, const float *sum) {
    TSP_REQUIRE(ctx && sum, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(!ctx->image_is_reduced, TSP_ESTATE, "the render target was already reduced for this frame (call tsp_render before reducing again)");
    TSP_HIP(hipSetDevice(ctx->device));
    TSP_HIP(hipMemcpy(ctx->image, sum, (size_t)ctx->R * ctx->R * ctx->C * sizeof(float), hipMemcpyHostToDevice));
    ctx->image_is_reduced = true;
    return TSP_OK;
}

static int ensure_lut(tsp_context *ctx, const float *lut_rgba, int n_lut) {
    TSP_REQUIRE(lut_rgba && n_lut >= 2 && n_lut <= 65536, TSP_EINVAL, "bad colormap LUT (n=%d)", n_lut);
    if (ctx->lut_capacity < n_lut) {
        if (ctx->lut) TSP_HIP(hipFree(ctx->lut));
        TSP_HIP(hipMalloc((void **)&ctx->lut, (size_t)n_lut * 4 * sizeof(float)));
        ctx->lut_capacity = n_lut;
    }
    TSP_HIP(hipMemcpyAsync(ctx->lut, lut_rgba, (size_t)n_lut * 4 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return TSP_OK;
}

int tsp_colormap_scalar(tsp_context *ctx, const float *lut_rgba, int n_lut, float vmin, float vmax, int log_scale,
                        int weighted, uint8_t *out_rgba) {
    TSP_REQUIRE(ctx && out_rgba, TSP_EINVAL, "NULL argument");
    TSP_HIP(hipSetDevice(ctx->device));
    int rc = ensure_lut(ctx, lut_rgba, n_lut);
    if (rc) return rc;
    const int64_t npix = (int64_t)ctx->R * ctx->R;
    rc = launch_colormap_scalar(ctx, ctx->image, npix, ctx->C, ctx->lut, n_lut, vmin, vmax, log_scale, weighted, ctx->out8);
    if (rc) return rc;
    TSP_HIP(hipMemcpyAsync(out_rgba, ctx->out8, (size_t)npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_colormap_rgb(tsp_context *ctx, float vmin, float vmax, float gamma, uint8_t *out_rgba8, float *out_rgba_f32) {
    TSP_REQUIRE(ctx && (out_rgba8 || out_rgba_f32), TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(ctx->C == 4, TSP_EINVAL, "rgb colormap needs a 4-channel image");
    TSP_HIP(hipSetDevice(ctx->device));
    const int64_t npix = (int64_t)ctx->R * ctx->R;
    if (out_rgba_f32 && !ctx->outf) TSP_HIP(hipMalloc((void **)&ctx->outf, (size_t)npix * 4 * sizeof(float)));
    int rc = launch_colormap_rgb(ctx, ctx->image, npix, ctx->C, vmin, vmax, gamma, out_rgba8 ? ctx->out8 : nullptr,
                                 out_rgba_f32 ? ctx->outf : nullptr);
    if (rc) return rc;
    if (out_rgba8) TSP_HIP(hipMemcpyAsync(out_rgba8, ctx->out8, (size_t)npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_rgba_f32)
        TSP_HIP(hipMemcpyAsync(out_rgba_f32, ctx->outf, (size_t)npix * 16, hipMemcpyDeviceToHost, ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_colormap_set_lut2d(tsp_context *ctx, const float *lut_rgba, int n) {
    TSP_REQUIRE(ctx && lut_rgba && n >= 2 && n <= 4096, TSP_EINVAL, "bad 2-D colormap LUT (n=%d)", n);
    TSP_HIP(hipSetDevice(ctx->device));
    if (ctx->lut2d_n != n || !ctx->lut2d) {
        // the size is recorded only once the new allocation exists: a failed hipMalloc leaves "no LUT"
        ctx->lut2d_n = 0;
        if (ctx->lut2d) (void)hipFree(ctx->lut2d);
        ctx->lut2d = nullptr;
        TSP_HIP(hipMalloc((void **)&ctx->lut2d, (size_t)n * n * 4 * sizeof(float)));
        ctx->lut2d_n = n;
    }
    TSP_HIP(hipMemcpy(ctx->lut2d, lut_rgba, (size_t)n * n * 4 * sizeof(float), hipMemcpyHostToDevice));
    return TSP_OK;
}

int tsp_colormap_bivariate(tsp_context *ctx, float vmin, float vmax, float density_vmin, float density_vmax, int log_scale,
                           int weighted, uint8_t *out_rgba) {
    TSP_REQUIRE(ctx && out_rgba, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(ctx->lut2d, TSP_ESTATE, "tsp_colormap_set_lut2d must be called first");
    TSP_HIP(hipSetDevice(ctx->device));
    const int64_t npix = (int64_t)ctx->R * ctx->R;
    int rc = launch_colormap_bivariate(ctx, ctx->image, npix, ctx->C, vmin, vmax, density_vmin, density_vmax, log_scale, weighted,
                                       ctx->out8);
    if (rc) return rc;
    TSP_HIP(hipMemcpyAsync(out_rgba, ctx->out8, (size_t)npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

static int ensure_scratch(tsp_context *ctx, size_t bytes) {
    if (ctx->scratch_bytes < bytes) {
        if (ctx->scratch) TSP_HIP(hipFree(ctx->scratch));
        TSP_HIP(hipMalloc(&ctx->scratch, bytes));
        ctx->scratch_bytes = bytes;
    }
    return TSP_OK;
}

int tsp_colormap_scalar_host(tsp_context *ctx, const float *img, int H, int W, int C, const float *lut_rgba, int n_lut,
                             float vmin, float vmax, int log_scale, int weighted, uint8_t *out_rgba) {
    TSP_REQUIRE(ctx && img && out_rgba, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(H > 0 && W > 0 && C >= 2, TSP_EINVAL, "bad image shape %dx%dx%d", H, W, C);
    TSP_HIP(hipSetDevice(ctx->device));
    const int64_t npix = (int64_t)H * W;
    const size_t in_bytes = (size_t)npix * C * sizeof(float), out_bytes = (size_t)npix * 4;
    int rc = ensure_scratch(ctx, in_bytes + out_bytes + 256);
    if (rc) return rc;
    if ((rc = ensure_lut(ctx, lut_rgba, n_lut))) return rc;
    float *d_in = (float *)ctx->scratch;
    uint8_t *d_out = (uint8_t *)ctx->scratch + ((in_bytes + 255) & ~(size_t)255);
    TSP_HIP(hipMemcpyAsync(d_in, img, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_colormap_scalar(ctx, d_in, npix, C, ctx->lut, n_lut, vmin, vmax, log_scale, weighted, d_out)))
        return rc;
    TSP_HIP(hipMemcpyAsync(out_rgba, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_colormap_bivariate_host(tsp_context *ctx, const float *img, int H, int W, int C, float vmin, float vmax,
                                float density_vmin, float density_vmax, int log_scale, int weighted, uint8_t *out_rgba) {
    TSP_REQUIRE(ctx && img && out_rgba, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(H > 0 && W > 0 && C >= 2, TSP_EINVAL, "bad image shape %dx%dx%d", H, W, C);
    TSP_REQUIRE(ctx->lut2d, TSP_ESTATE, "tsp_colormap_set_lut2d must be called first");
    TSP_HIP(hipSetDevice(ctx->device));
    const int64_t npix = (int64_t)H * W;
    const size_t in_bytes = (size_t)npix * C * sizeof(float), out_bytes = (size_t)npix * 4;
    int rc = ensure_scratch(ctx, in_bytes + out_bytes + 256);
    if (rc) return rc;
    float *d_in = (float *)ctx->scratch;
    uint8_t *d_out = (uint8_t *)ctx->scratch + ((in_bytes + 255) & ~(size_t)255);
    TSP_HIP(hipMemcpyAsync(d_in, img, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_colormap_bivariate(ctx, d_in, npix, C, vmin, vmax, density_vmin, density_vmax, log_scale, weighted, d_out)))
        return rc;
    TSP_HIP(hipMemcpyAsync(out_rgba, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_colormap_rgb_host(tsp_context *ctx, const float *img, int H, int W, int C, float vmin, float vmax, float gamma,
                          uint8_t *out_rgba8, float *out_rgba_f32) {
    TSP_REQUIRE(ctx && img && (out_rgba8 || out_rgba_f32), TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(H > 0 && W > 0 && C >= 3, TSP_EINVAL, "bad image shape %dx%dx%d", H, W, C);
    TSP_HIP(hipSetDevice(ctx->device));
    const int64_t npix = (int64_t)H * W;
    const size_t in_bytes = ((size_t)npix * C * sizeof(float) + 255) & ~(size_t)255;
    const size_t o8 = ((size_t)npix * 4 + 255) & ~(size_t)255, of = (size_t)npix * 16;
    int rc = ensure_scratch(ctx, in_bytes + o8 + of);
    if (rc) return rc;
    float *d_in = (float *)ctx->scratch;
    uint8_t *d_o8 = (uint8_t *)ctx->scratch + in_bytes;
    float *d_of = (float *)((uint8_t *)ctx->scratch + in_bytes + o8);
    TSP_HIP(hipMemcpyAsync(d_in, img, (size_t)npix * C * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_colormap_rgb(ctx, d_in, npix, C, vmin, vmax, gamma, out_rgba8 ? d_o8 : nullptr,
                                  out_rgba_f32 ? d_of : nullptr)))
        return rc;
    if (out_rgba8) TSP_HIP(hipMemcpyAsync(out_rgba8, d_o8, (size_t)npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_rgba_f32) TSP_HIP(hipMemcpyAsync(out_rgba_f32, d_of, of, hipMemcpyDeviceToHost, ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

int tsp_tile_periodic(tsp_context *ctx, int n, const float *offsets_xy, const float *weights) {
    TSP_REQUIRE(ctx && n >= 0 && n <= 4096 && (n == 0 || (offsets_xy && weights)), TSP_EINVAL, "bad argument");
    TSP_HIP(hipSetDevice(ctx->device));
    return tile_periodic(ctx, n, offsets_xy, weights);
}

int tsp_content_sort(tsp_context *ctx, int kind, float scale, int64_t *n_finite, int64_t *n_nonpositive) {
    TSP_REQUIRE(ctx && n_finite && n_nonpositive, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(kind >= 0 && kind <= 3, TSP_EINVAL, "bad content kind %d", kind);
    TSP_REQUIRE(kind != 2 || ctx->C == 4, TSP_EINVAL, "rgb content needs a 4-channel image");
    TSP_HIP(hipSetDevice(ctx->device));
    return content_sort(ctx, kind, scale, n_finite, n_nonpositive);
}

int tsp_content_values(tsp_context *ctx, const int64_t *ranks, int n_ranks, float *out) {
    TSP_REQUIRE(ctx && ranks && out && n_ranks >= 0, TSP_EINVAL, "bad argument");
    TSP_REQUIRE(ctx->sort_keys_alt, TSP_ESTATE, "tsp_content_sort has not been called");
    TSP_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < n_ranks; ++i) {
        TSP_REQUIRE(ranks[i] >= 0 && ranks[i] < ctx->sorted_count, TSP_EINVAL, "rank %lld outside [0, %lld)",
                    (long long)ranks[i], (long long)ctx->sorted_count);
        uint32_t key;
        TSP_HIP(hipMemcpy(&key, ctx->sort_keys_alt + ranks[i], 4, hipMemcpyDeviceToHost));
        const uint32_t bits = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;   // inverse of the monotone map
        memcpy(&out[i], &bits, 4);
    }
    return TSP_OK;
}

int tsp_get_stats(tsp_context *ctx, tsp_stats *out) {
    TSP_REQUIRE(ctx && out, TSP_EINVAL, "NULL argument");
    *out = ctx->stats;
    return TSP_OK;
}

int tsp_set_option(tsp_context *ctx, const char *name, int64_t value) {
    TSP_REQUIRE(ctx && name, TSP_EINVAL, "NULL argument");
    if (!strcmp(name, "count_fragments")) {
        ctx->count_fragments = value != 0;
        return TSP_OK;
    }
    if (!strcmp(name, "p_small_milli")) {     // class boundary small/mid in 1/1000 px (<= 16000: kernel S holds mips 2 and 3 and packs <= 16 texel columns)
        TSP_REQUIRE(value >= 0 && value <= 16000, TSP_EINVAL, "p_small out of range (kernel S packs <= 16 texel columns per footprint)");
        ctx->p_small = (float)value * 1e-3f;
        return TSP_OK;
    }
    if (!strcmp(name, "debug_no_raster")) {
        ctx->debug_no_raster = value != 0;
        return TSP_OK;
    }
    if (!strcmp(name, "huge_split") || !strcmp(name, "stream_blocks_per_cu")) {
        TSP_REQUIRE(value >= 0 && value <= 4096, TSP_EINVAL, "%s out of range", name);
        if (name[0] == 'h') ctx->huge_split = (int)value;
        else ctx->stream_blocks_per_cu = (int)value;      // 0 = as many as stay resident
        return TSP_OK;
    }
    if (!strcmp(name, "slice_records")) {     // deferred footprints per launch of kernels G / H2 (0 = default: 2^27 mid, 2^30 huge)
        TSP_REQUIRE(value == 0 || (value >= 64 && value <= (1ll << 27)), TSP_EINVAL, "%s: 0 or 64 .. 2^27", name);
        ctx->slice_records = value;
        return TSP_OK;
    }
    if (!strcmp(name, "debug_fail_stage")) {   // test aid: the next tsp_render fails after kernel S (1) or after kernel G (2)
        TSP_REQUIRE(value >= 0 && value <= 2, TSP_EINVAL, "%s out of range", name);
        ctx->debug_fail_stage = (int)value;
        return TSP_OK;
    }
    if (!strcmp(name, "debug_gather_full_lut")) { ctx->debug_gather_full_lut = value ? 1 : 0; return TSP_OK; }
    if (!strcmp(name, "mid_item_scale_milli")) {
        TSP_REQUIRE(value >= 1 && value <= 100000, TSP_EINVAL, "%s out of range", name);
        ctx->mid_item_scale = (double)value * 1e-3;
        return TSP_OK;
    }
    if (!strcmp(name, "mid_narrow_px_milli")) {    // mid footprints below this many 1/1000 px go to kernel N (0 = none)
        TSP_REQUIRE(value >= 0 && value <= 64000, TSP_EINVAL, "%s out of range", name);
        ctx->mid_narrow_px = (float)value * 1e-3f;
        return TSP_OK;
    }
    if (!strcmp(name, "mid_item_records")) {
        TSP_REQUIRE(value == 0 || (value >= 64 && value <= 8192 && (value & (value - 1)) == 0), TSP_EINVAL, "%s: 0 or a power of two from 64 to 8192", name);
        ctx->mid_item_records = (int)value;
        return TSP_OK;
    }
    if (!strcmp(name, "stream_batch_chunks")) {
        TSP_REQUIRE(value >= 4 && value <= 4096, TSP_EINVAL, "%s out of range", name);
        ctx->stream_batch_chunks = (int)value;
        return TSP_OK;
    }
    if (!strcmp(name, "huge_band_mib")) {     // memory the band bins of the huge records may take, MiB (0 = never bin: kernel H2 scans one list)
        TSP_REQUIRE(value >= 0 && value <= (1 << 20), TSP_EINVAL, "%s out of range", name);
        ctx->huge_band_budget = value << 20;
        return TSP_OK;
    }
    if (!strcmp(name, "huge_variant")) {
        // 1 = auto; 2, 4-7 force a strip shape / occupancy of kernel H2 (A/B and tests)
        TSP_REQUIRE(value >= 1 && value <= 7 && value != 3, TSP_EINVAL, "huge_variant out of range");
        ctx->huge_variant = (int)value;
        return TSP_OK;
    }
    if (!strcmp(name, "reorder_interleave")) {   // read by the next tsp_reorder_spatial
        TSP_REQUIRE(value >= 0 && value <= 2, TSP_EINVAL, "%s out of range", name);
        ctx->reorder_interleave = (int)value;      // 0: Morton order inside the blocks, 1: 64 x 8 transposition, 2: by descending h
        return TSP_OK;
    }
    if (!strcmp(name, "chunk_cull")) {        // 1 (default): kernel S skips the chunks whose bounds lie outside the view
        ctx->chunk_cull = value != 0;
        return TSP_OK;
    }
    if (!strcmp(name, "overlap_mid_huge")) {
        ctx->overlap_mid_huge = value != 0;
        return TSP_OK;
    }
    if (!strcmp(name, "use_quantity")) {      // 0: render density-only without dropping the resident q array
        ctx->use_quantity = value != 0;
        return TSP_OK;
    }
    if (!strcmp(name, "active_channels")) {   // layout tsp_write_image / the colormap calls assume
        TSP_REQUIRE((value == 2 || value == 4) && value <= ctx->Ccap, TSP_EINVAL, "bad channel count %lld", (long long)value);
        ctx->C = (int)value;
        return TSP_OK;
    }
    set_error("unknown option '%s'", name);
    return TSP_EINVAL;
}

int tsp_measure_read_bandwidth(tsp_context *ctx, int64_t bytes, int iters, double *gbps_out) {
    TSP_REQUIRE(ctx && gbps_out && bytes > 0 && iters > 0, TSP_EINVAL, "bad argument");
    TSP_HIP(hipSetDevice(ctx->device));
    return measure_read_bandwidth(ctx, bytes, iters, gbps_out);
}

}  // extern "C"
