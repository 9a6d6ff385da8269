// tsp_internal.h -- context layout and helpers shared by the translation units of libtopsy_splat.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>
#include <stdio.h>
#include <string>

#include "../../include/topsy_splat.h"
#include "tsp_math.h"

namespace tsp {

void set_error(const char *fmt, ...);

#define TSP_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            tsp::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return TSP_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

#define TSP_REQUIRE(cond, code, ...)       \
    do {                                   \
        if (!(cond)) {                     \
            tsp::set_error(__VA_ARGS__);   \
            return (code);                 \
        }                                  \
    } while (0)

// Resident particle data: struct-of-arrays float32 in HBM, one allocation per attribute so each
// streams as an independent fully-coalesced sequence (20 B/particle density, 24 B with q, 28 B rgb).
struct Particles {
    int64_t n = 0;
    float *x = nullptr, *y = nullptr, *z = nullptr, *h = nullptr, *m = nullptr;
    float *q = nullptr;                      // nullptr -> density render (q = 0)
    float *r = nullptr, *g = nullptr, *b = nullptr;
    uint32_t *perm = nullptr;                // new -> old index after tsp_reorder_spatial (else nullptr)
    // camera-independent vertex weights (sph.wgsl:76-83 `mass / (h*h)`, :69-73 `rgb / (h*h)`), formed once per upload by
    // ensure_weights() with the float32 operations the shader performs per vertex and frame: what kernel S streams
    // instead of m (r, g, b) -- the same bytes per particle, no division per particle and frame
    float *wm = nullptr;
    float *wr = nullptr, *wg = nullptr, *wb = nullptr;
    bool wm_valid = false, wrgb_valid = false;   // cleared whenever h / m / rgb change (upload, generate, reorder)
};

// Bounds of every block of BOUNDS_BLOCK consecutive particles (view culling of whole chunks, kernel S): two float4 per block,
// (xmin, ymin, zmin, hmax) and (xmax, ymax, zmax, -).  NaN coordinates are ignored (such a particle draws nothing).
constexpr int BOUNDS_BLOCK = 512;            // = CHUNK of tsp_pipeline.hip (static_assert there)

// Deferred-footprint record written by the streaming kernel for the tile kernels (20 B).
struct Record {
    float pcx, pcy, P, w0, w1;
};
struct Record4 {   // rgb variant (24 B)
    float pcx, pcy, P, w0, w1, w2;
};

struct Counters {      // device-side, zeroed per render call
    unsigned long long n_small, n_mid, n_huge, n_culled, n_fragments, huge_count, next_chunk, mid_odd_weights;     // next_chunk: kernel S's batch counter; mid_odd_weights: the mid list holds a weight that is not finite (kernel N's fill pass)
    unsigned long long n_frag_class[4];   // n_fragments by the kernel that drew them: S, G, H2, (unused)
};

struct Workspace {     // per-context scratch of the three-class pipeline (grown on demand)
    void *mid_geom = nullptr, *mid_w = nullptr;     // deferred mid footprints: float4 geometry + weights
    int64_t mid_capacity = 0;
    void *huge_geom = nullptr, *huge_w = nullptr;   // deferred huge footprints
    int64_t huge_capacity = 0;
    void *hband_geom = nullptr, *hband_w = nullptr; // the huge records once more, binned by 64-row image band (n_bands regions of hband_stride records)
    int *hband_count = nullptr;                     // records per band
    int64_t hband_stride = 0; int hband_bands = 0;
    void *mband_geom = nullptr, *mband_w = nullptr; // the mid records binned by tile for kernel G (exact-size bins)
    int *mband_count = nullptr;                     // records per tile | fill cursors
    long long *mband_base = nullptr;                // first record of each tile's bin
    int64_t mband_capacity = 0;                     // records the bins can hold
    int mtile_capacity = 0;                         // strips the per-strip arrays can hold
    int64_t mitem_capacity = 0;                     // work items the item table can hold
    int *mitem_tile = nullptr, *mitem_base = nullptr; // kernel G work items: item -> tile, tile -> first item
    int64_t chunk_capacity = 0;         // chunks alive_list can hold
    float4 *block_bounds = nullptr;     // chunk culling: bounds of every BOUNDS_BLOCK particles (valid while bounds_valid)
    int64_t bounds_capacity = 0;        // blocks
    bool bounds_valid = false;          // cleared whenever positions / smoothing lengths change (upload, generate, reorder)
    int *alive_list = nullptr;          // chunk culling: the chunks of this render call that may reach the view (seg_capacity entries)
    unsigned long long *cull_info = nullptr;   // [0] chunks alive, [1] particles in culled chunks
    int64_t *range_prefix = nullptr;    // device copy of the ranges of the current call
    int64_t range_capacity = 0;
    int *count_diff = nullptr;          // rgb: (R+1)^2 corner-difference image of the huge footprints' pixel rectangles
    int *count_band = nullptr;          // rgb: per (64-row band, column) sums of its row-scanned form
};

}  // namespace tsp

struct tsp_context {
    int device = 0;
    int R = 0, C = 0, Ccap = 0;       // C = active channels (2 or 4) <= Ccap
    bool use_quantity = true;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;    // option overlap_mid_huge: kernel G runs here, concurrently with kernel H2 on `stream`
    hipEvent_t ev[12] = {};
    float *image = nullptr;           // R*R*C float32 render target (what read-back, colormap and reduce see)
    double *image64 = nullptr;        // float64 master copy every kernel accumulates into (rounded once per render)
    double *image64_entry = nullptr;  // image64 as tsp_render found it: what a failed block puts back (the call draws all of a block or none of it)
    float *mips = nullptr;            // 5440 floats
    bool have_mips = false;
    bool lut_mirror_symmetric = false;    // every mip level equals its left-right and top-bottom mirror images bit for bit
    bool lut_zero_outside_disc = false;   // every level-0 texel whose centre is >= 2h from the centre is exactly 0
    tsp::Particles p;
    tsp::Counters *counters = nullptr;
    tsp::Workspace ws;
    uint8_t *out8 = nullptr;          // R*R*4 staging for colormap output
    float *outf = nullptr;            // R*R*4 float staging (HDR)
    float *lut = nullptr;             // colormap LUT on device
    int lut_capacity = 0;
    float *lut2d = nullptr;           // bivariate colormap LUT, n x n x RGBA
    int lut2d_n = 0;
    void *scratch = nullptr;          // host-image colormap staging
    size_t scratch_bytes = 0;
    uint32_t *sort_keys = nullptr, *sort_keys_alt = nullptr;   // content order statistics (autorange)
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int64_t sort_capacity = 0, sorted_count = 0;
    tsp_stats stats = {};
    std::vector<int64_t> cell_offsets;     // first index of every (stratum, Morton cell) run of the last reorder_spatial, then n
    int cell_bits = 0;                     // the cells form a (2^cell_bits)^3 grid over the bounding box
    float cell_lo[3] = {0, 0, 0}, cell_width[3] = {0, 0, 0};
    std::vector<int64_t> strata_offsets;   // first index of every stratum of the last reorder_spatial, then n
    uint32_t kernel_attr_done = 0;   // bit per kernel family whose dynamic-LDS limit was raised on this context's device
    bool count_fragments = false;
    // pipeline tuning (tsp_set_option)
    float p_small = 16.0f;             // footprints narrower than this many pixels are splatted by kernel S (mips 3 and 2; <= 16: its texel columns are packed 16 x 4 bits)
    int64_t huge_band_budget = 6ll << 30;   // bytes the band bins of the huge records may take (n_bands x n_huge records); above it kernel H2 scans one list
    int huge_variant = 1;             // kernel H2's strip shape / occupancy: 1 = auto (density: 64x32 strips at 8 waves/SIMD from 7e5 records, 64x16 below; two channels 64x16 at 7; rgb at 5), 2 / 4-7 = A/B builds
    int huge_split = 0;              // workgroups per image tile of kernel H2 (0 = auto)
    int reorder_interleave = 2;     // tsp_reorder_spatial's arrangement inside every 512-particle block: 0 Morton order, 1 transposed 64 x 8, 2 by descending smoothing length (tsp_data.hip)
    int stream_blocks_per_cu = 0;    // kernel S: persistent workgroups per CU (0 = what the occupancy query reports)
    int debug_gather_full_lut = 0;   // kernel G: 1 = the whole mip pyramid in LDS even when the kernel image is symmetric (measurement aid)
    double mid_item_scale = 0.35;    // kernel G: records per work item = this x sqrt(records), rounded to a power of two (option mid_item_scale_milli)
    float mid_narrow_px = 64.0f;     // mid footprints narrower than this are drawn by kernel N, four records per wave step (0: all by kernel G; option mid_narrow_px_milli)
    int mid_item_records = 0;        // kernel G: records per work item (0 = by list length; a power of two from 64 to 1024)
    int stream_batch_chunks = 8;     // kernel S: the largest batch of consecutive chunks a workgroup takes from the shared counter
    bool chunk_cull = true;           // chunks (512 consecutive particles) whose bounds lie outside the view are skipped by kernel S
                                      // unread: pays with a load-time spatial order (tsp_reorder_spatial); identical results
    int64_t chunk_culled_particles = 0;   // of the last render call
    bool overlap_mid_huge = false;    // option: kernels G and H2 on two streams (measured: no gain at 1.25e8, +6 % at 1e7)
    int64_t slice_records = 0;        // option: deferred records kernels G / H2 take per launch (0 = 2^27 mid / 2^30 huge); a block of any size draws in slices
    int debug_fail_stage = 0;         // test aid: the next render fails with TSP_ENOMEM after kernel S (1) / after kernel G (2); cleared by the failure
    int stream_occ[3][2] = {};        // kernel S: workgroups resident per CU by [mode][one-channel window | all channels] (occupancy query, once per context)
    bool debug_no_raster = false;    // measurement aid: kernel S classifies and emits records but rasterises nothing (the image is then incomplete)
    int cu_count = 256;
    // RCCL
    void *comm = nullptr;
    int n_ranks = 1, rank = 0;
    bool image_is_reduced = false;    // `image` already holds the cross-rank sum of the current frame (tsp_comm_reduce_image)
};

namespace tsp {
// hipMalloc'd scratch that is released on every exit path
struct DeviceScratch {
    void *p = nullptr;
    DeviceScratch() = default;
    DeviceScratch(const DeviceScratch &) = delete;
    DeviceScratch &operator=(const DeviceScratch &) = delete;
    ~DeviceScratch() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
    void *release() { void *q = p; p = nullptr; return q; }
    void reset(void *q) { if (p) (void)hipFree(p); p = q; }
};

// kernels / launchers implemented in the other translation units
int launch_generic(tsp_context *ctx, const Camera &cam, const int64_t *d_ranges, int n_ranges,
                   int64_t total, int mode, int rule);
int launch_pipeline(tsp_context *ctx, const Camera &cam, const int64_t *h_starts, const int64_t *h_lens,
                    int n_ranges, int64_t total, int mode);
int launch_colormap_scalar(tsp_context *ctx, const float *d_img, int64_t npix, int C, const float *d_lut,
                           int n_lut, float vmin, float vmax, int log_scale, int weighted, uint8_t *d_out);
int launch_colormap_rgb(tsp_context *ctx, const float *d_img, int64_t npix, int C, float vmin, float vmax,
                        float gamma, uint8_t *d_out8, float *d_outf);
int launch_colormap_bivariate(tsp_context *ctx, const float *d_img, int64_t npix, int C, float vmin, float vmax,
                             float dvmin, float dvmax, int log_scale, int weighted, uint8_t *d_out);
int generate_synthetic(tsp_context *ctx, int64_t n_total, int64_t first, int64_t count, uint64_t seed,
                       float h_cap, int with_quantity, int with_rgb);
int reorder_spatial(tsp_context *ctx, int n_strata, uint64_t seed, int64_t *perm_out);
int ensure_weights(tsp_context *ctx, bool rgb);    // (re)computes p.wm or p.wr / wg / wb on ctx->stream when the particles changed
int ensure_block_bounds(tsp_context *ctx);     // (re)computes ws.block_bounds on ctx->stream when the particles changed
int measure_read_bandwidth(tsp_context *ctx, int64_t bytes, int iters, double *gbps_out);
int launch_image_convert(tsp_context *ctx, bool to_float);
int tile_periodic(tsp_context *ctx, int n, const float *h_offsets, const float *h_weights);
int content_sort(tsp_context *ctx, int kind, float scale, int64_t *n_finite, int64_t *n_nonpositive);   // image64 -> image (true) or image -> image64 (false)
int ensure_array(float **p, int64_t n);
}  // namespace tsp
