/* topsy_splat.h -- C-ABI of the MI355X-native SPH particle-splatting backend for topsy.
 *
 * This is the drop-in boundary for ONE hot path of pynbody/topsy: per-particle smoothing-kernel
 * rasterisation into a float32 image (reference src/topsy/sph.py + shaders/sph.wgsl) and the
 * 1-D-LUT / log-scale colormap post-pass (reference src/topsy/colormap/implementation.py +
 * shaders/colormap.wgsl).  In the reference that path sits behind a Python object protocol
 * (Visualizer <-> SPH / ColormapHolder) whose device side is wgpu; here the device side is this
 * library (hand-written HIP for gfx950) and the Python side (topsy_amd/) binds it with ctypes.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success, a negative TSP_E* code on error;
 *     tsp_last_error() gives the text of the most recent failure on the calling thread.
 *   - the caller owns every host array; the library copies on upload and writes only into
 *     caller-allocated output buffers.  No callbacks, no exceptions cross the boundary.
 *   - one context = one GPU (one process per GPU for multi-GPU; see tsp_comm_*).  Calls on one
 *     context must be serialised by the caller.  All calls are synchronous (they return after
 *     the GPU work they issued has completed), mirroring the reference's
 *     submit + on_submitted_work_done_sync pairs (src/topsy/util.py:84-99).
 *   - images are row-major, row 0 = top (+y), channels interleaved: float32 [R][R][C].
 */
#ifndef TOPSY_SPLAT_H
#define TOPSY_SPLAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsp_context tsp_context;

enum {
    TSP_OK = 0,
    TSP_EINVAL = -1,   /* bad argument */
    TSP_EHIP = -2,     /* HIP runtime error (text in tsp_last_error) */
    TSP_ENODEV = -3,   /* no usable GPU */
    TSP_ESTATE = -4,   /* call order violated (e.g. render before upload / before kernel LUT) */
    TSP_ECOMM = -5,    /* RCCL error */
    TSP_ENOMEM = -6
};

/* Render modes: which per-particle channels feed the image (reference SPH subclasses). */
enum {
    TSP_MODE_WEIGHTED = 0, /* SPH: ch0 += k*m/h^2, ch1 += k*m/h^2*q      (sph.wgsl:76-83,139-146)  C=2 */
    TSP_MODE_DEPTH = 1,    /* DepthSPH: ch1 weights by clip-space z       (sph.wgsl:86-91)          C=2 */
    TSP_MODE_RGB = 2       /* RGBSPH: ch0..2 += k*(r,g,b)/h^2, ch3 += 1   (sph.wgsl:69-73,161-165)  C=4 */
};

/* Pipeline selection for tsp_render (flags argument). 0 = default (fast three-class pipeline). */
enum {
    TSP_PIPE_DEFAULT = 0,
    TSP_PIPE_GENERIC = 1,  /* single generic kernel, global atomics only (cross-check / debugging) */
    /* Kernel-texture sampling rule.  The default reproduces every golden vector of the
     * reference's tests: bilinear on mip 0 when the footprint is >= 64 px wide (LOD <= 0, mag filter linear,
     * src/topsy/sph.py:425-426), else the NEAREST texel of the mip the rounded LOD selects (min / mipmap filters
     * left at wgpu's default).  The two alternatives below are what a driver with other filter defaults would
     * do; they exist to diagnose such differences and run on the generic kernel only. */
    TSP_SAMPLE_BILINEAR_MIP0 = 0x10,  /* bilinear on mip 0 whatever the footprint width */
    TSP_SAMPLE_BILINEAR_MIP = 0x20    /* bilinear within the mip the rounded LOD selects */
};

const char *tsp_last_error(void);
/* ABI version.  100: first release.  101: tsp_stats grew by 16 bytes (ms_mega, n_mega appended) -- tsp_get_stats
 * writes sizeof(tsp_stats) bytes, so a client compiled against the version-100 header must not be run against a
 * version-101 library (check tsp_version() >= 101 and tsp_stats_size() == sizeof(tsp_stats) at start-up).  The same
 * release narrowed the option "p_small_milli" from <= 22627 to <= 16000 (kernel S packs at most 16 texel columns per
 * footprint; the default moved from 13.5 to 16 px): values 16001..22627 now return TSP_EINVAL.
 * 102: tsp_stats grew by 32 bytes (n_fragments_stream / _mid / _huge / _mega appended); same rule.
 * 103: tsp_stats grew by 8 bytes (n_chunk_culled appended); same rule.  Defaults changed without an ABI change: kernel H3
 * (matrix cores) is an option ("p_mega_px" / "p_mega2_px" default to 0), chunk culling ("chunk_cull") is on.
 * 104: the matrix-core kernels and the round-1 gather kernel are gone (no default rule selected them): the options
 * "p_mega_px", "p_mega2_px", "p_mega_rgb_px", "mega_variant", "rgb_mega_variant", "mega_split", "integrated_px" (kernel I, the
 * inexact option of round 3) and "huge_variant" = 0 / 3 now return TSP_EINVAL; tsp_stats keeps its layout (ms_mega, n_mega and
 * n_fragments_mega are reserved: always 0).  New entry points: tsp_set_reduced_image, tsp_group_shard_range,
 * tsp_group_upload_band_magnitudes.
 * 105: the LDS scatter kernel of the footprints below 64 px (kernel M) is gone -- kernel G, a register gather over per-strip bins of
 * the deferred records, draws them at every size: the options "mid_split" and "debug_extra_lds" return TSP_EINVAL; new options
 * "mid_item_records", "mid_item_scale_milli", "stream_batch_chunks", "debug_gather_full_lut"; "stream_blocks_per_cu" now counts the
 * persistent workgroups of kernel S per CU (0 = as many as stay resident).  No entry point or struct changed. */
int tsp_version(void);
int tsp_stats_size(void);

/* Number of visible GPUs (hipGetDeviceCount); <0 on error. Does not create a HIP context. */
int tsp_device_count(void);

/* Create a renderer for one GPU.  Mirrors SPH.__init__ (reference src/topsy/sph.py:50-88):
 * allocates the R x R x C float32 render target.  n_channels: 2 (SPH / DepthSPH, rg32float,
 * sph.py:23) or 4 (RGBSPH, rgba32float, sph.py:432-439). */
int tsp_create(int device_id, int resolution, int n_channels, tsp_context **out);
void tsp_destroy(tsp_context *ctx);

/* Kernel texture: n_levels mip levels of sizes n0, n0/2, ... concatenated, float32
 * (reference SPH._setup_kernel_texture, src/topsy/sph.py:396-426; n0 = 64, n_levels = 4).
 * Sampler semantics are fixed to the reference's: mag linear, min/mip nearest, clamp-to-edge. */
int tsp_set_kernel_mips(tsp_context *ctx, const float *lut, int n0, int n_levels);

/* Particle upload, SoA float32, caller's (global) index order is preserved.
 * Replaces ParticleBuffers.get_pos_smooth_buffers / get_mass_and_quantity_buffers /
 * get_rgb_buffers (reference src/topsy/particle_buffers.py:84-118).
 * mass may be NULL for a context that will only render TSP_MODE_RGB. */
int tsp_upload_particles(tsp_context *ctx, int64_t n, const float *x, const float *y, const float *z,
                         const float *h, const float *mass);
/* q == NULL selects the density render (reference uploads q = 0 then, particle_buffers.py:96-99;
 * quantity swap: src/topsy/visualizer.py:294-309). */
int tsp_upload_quantity(tsp_context *ctx, const float *q);
int tsp_upload_rgb(tsp_context *ctx, const float *r, const float *g, const float *b);
/* The same three channels computed ON the device from SSP band magnitudes -- the 3 x n_bands "band contraction" of the
 * rgb render mode: channel_c[i] = sum_b weights[c * n_bands + b] * 10^(-0.4 * mags[b * n + i]), NaN -> 0, evaluated in
 * float64 and rounded to float32 once.  With weights = diag(0.5, 1, 1) over the bands (I, V, U) this is the reference's
 * PynbodyDataInMemory.get_rgb_masses / _effective_mass_for_band (src/topsy/loader.py:112-121).  mags: n_bands arrays of n
 * float64 each, contiguous, caller's particle order.  An HBM-bound VALU kernel (8 n_bands bytes in, 12 bytes out and
 * 3 n_bands FMAs per particle): the matrix cores have nothing to gain here. */
int tsp_upload_band_magnitudes(tsp_context *ctx, int n_bands, const double *mags, const double *weights);

/* On-device synthetic snapshot = restatement of topsy.loader.TestDataLoader's distribution
 * (reference src/topsy/loader.py:241-332) with a counter-based generator, so that shard
 * [first, first+count) of an n_total-particle snapshot is reproducible on any GPU without
 * materialising n_total rows on the host.  h_cap > 0 caps the smoothing length (bandwidth-bound
 * variant of BASELINE.md section 3); with_quantity / with_rgb also fill q / rgb. */
int tsp_generate_synthetic(tsp_context *ctx, int64_t n_total, int64_t first, int64_t count,
                           uint64_t seed, float h_cap, int with_quantity, int with_rgb);

/* Load-time spatial ordering (the analogue of the reference's CellLayout sort at load,
 * src/topsy/loader.py:88-97): reorders the resident particles into n_strata uniform random
 * strata, each Morton-sorted, so that index prefixes remain unbiased samples (progressive
 * rendering) while consecutive indices are screen-coherent.  perm_out (optional, host, n
 * int64) receives new->old indices.  Later tsp_upload_quantity/rgb calls are given in the OLD
 * order and permuted by the library. */
int tsp_reorder_spatial(tsp_context *ctx, int n_strata, uint64_t seed, int64_t *perm_out);
/* First index of every stratum of the last tsp_reorder_spatial call, plus the particle count:
 * n_strata + 1 ascending int64 values.  A contiguous index range is an unbiased spatial sample only
 * when it is a union of whole strata, so the progressive renderer ends its blocks on these offsets
 * (the reference gets the same property from its per-cell random order, src/topsy/progressive_render.py:152-187).
 * Returns the number of values written (<= capacity), 0 when the particles were never reordered. */
int tsp_get_strata_offsets(tsp_context *ctx, int64_t *offsets_out, int capacity);

/* View culling on the library's own ordering (the role of the reference's CellLayout and
 * RenderProgressionWithCells._map_logical_range_to_actual_ranges, src/topsy/cell_layout.py:26-31,63-113,
 * src/topsy/progressive_render.py:152-187,207-220).  Inside a stratum the Morton order stores the particles of every cell
 * of a cells_per_axis^3 grid over the snapshot's bounding box as ONE contiguous run.  tsp_get_cell_layout reports the grid
 * (cells_per_axis = 2^k, the largest k <= 4 with >= 16 particles per (stratum, cell) on average; cell (cx, cy, cz) covers
 * box_lo + (cx, cy, cz) * cell_width ... + cell_width per axis), tsp_get_cell_offsets the first index of every run:
 * n_strata * cells_per_axis^3 + 1 ascending int64 values, entry s * cells^3 + code for stratum s and Morton cell code
 * (bit 3 j of the code = bit j of cx, bit 3 j + 1 = bit j of cy, bit 3 j + 2 = bit j of cz), the last entry = n.
 * The host selects the cells that meet the view sphere and hands tsp_render the (start, len) runs of those cells.
 * tsp_get_cell_offsets returns the number of values written (<= capacity), 0 when the particles were never reordered. */
int tsp_get_cell_layout(tsp_context *ctx, int *n_strata_out, int *cells_per_axis_out, float *box_lo_out /*3*/,
                        float *cell_width_out /*3*/);
int64_t tsp_get_cell_offsets(tsp_context *ctx, int64_t *offsets_out, int64_t capacity);

/* Copy resident particle arrays back (testing / fixtures). Any pointer may be NULL. */
int tsp_download_particles(tsp_context *ctx, float *x, float *y, float *z, float *h, float *mass,
                           float *q, float *r, float *g, float *b);
int64_t tsp_num_particles(tsp_context *ctx);

/* One render block.  Replaces the body of SPH.render's loop (reference src/topsy/sph.py:318-326:
 * encode_render_pass(clear) + update_particle_ranges(starts, lens) + timed queue.submit).
 *   M             row-major 4x4, clip = M * (x,y,z,1)  (= transpose of the reference's uploaded
 *                 "transform", src/topsy/sph.py:268-289)
 *   scale_factor  1/scale (sph.wgsl:58)
 *   starts/lens   n_ranges particle index ranges (first_instance, instance_count of the
 *                 reference's indirect draws, particle_buffers.py:76-82); NULL = all particles
 *   clear         1 = clear the target first (load_op clear, sph.py:346)
 *   mode          TSP_MODE_*
 *   flags         TSP_PIPE_*
 *   gpu_ms_out    optional: GPU time of this block (hipEvent pair) -- the TimeGpuOperation hook
 * A block of any size draws (the deferred-footprint lists go through the tile kernels in slices), and it draws whole or not at
 * all: on any error return the accumulator, the image, the channel layout and tsp_stats are as the call found them. */
int tsp_render(tsp_context *ctx, const float *M, float scale_factor, const int64_t *starts,
               const int64_t *lens, int n_ranges, int clear, int mode, int flags, double *gpu_ms_out);

/* Read the render target (R*R*C float32).  SPH._get_image_unscaled, src/topsy/sph.py:127-140. */
int tsp_read_image(tsp_context *ctx, float *out);
/* Overwrite the render target from the host (testing the colormap on a known buffer). */
int tsp_write_image(tsp_context *ctx, const float *in);

/* Colormap post-pass on the resident render target -> RGBA8 (R*R*4 bytes, RGBA order).
 * scalar: colormap.wgsl fragment_main non-bivariate branch (:113-127); lut = n_lut x RGBA float32
 * (Colormap._generate_mapping_rgba_f32, implementation.py:235-238); vmin/vmax are the
 * already-scaled shader parameters (Colormap._update_parameter_buffer, implementation.py:427-453). */
int tsp_colormap_scalar(tsp_context *ctx, const float *lut_rgba, int n_lut, float vmin, float vmax,
                        int log_scale, int weighted, uint8_t *out_rgba);
/* rgb: colormap.wgsl fragment_main_tri + gamma_map (:131-159).  out_rgba8 and/or out_rgba_f32
 * (unclamped, the HDR canvas value) may be NULL. */
int tsp_colormap_rgb(tsp_context *ctx, float vmin, float vmax, float gamma, uint8_t *out_rgba8,
                     float *out_rgba_f32);

/* Bivariate map: colormap.wgsl fragment_main BIVARIATE branch (:91-111) with
 * the 2-D LUT of BivariateColormap._generate_mapping_rgba_f32 (implementation.py:585-605), n x n x RGBA
 * float32, first axis = normalised (weighted) value, second axis = normalised log10 density.  The LUT
 * (16 MB at n = 1000) is uploaded once with tsp_colormap_set_lut2d and stays resident. */
int tsp_colormap_set_lut2d(tsp_context *ctx, const float *lut_rgba, int n);
int tsp_colormap_bivariate(tsp_context *ctx, float vmin, float vmax, float density_vmin, float density_vmax,
                           int log_scale, int weighted, uint8_t *out_rgba);
int tsp_colormap_bivariate_host(tsp_context *ctx, const float *img, int H, int W, int C, float vmin, float vmax,
                                float density_vmin, float density_vmax, int log_scale, int weighted,
                                uint8_t *out_rgba);

/* Same maps applied to an arbitrary host image (H x W x C float32), the entry
 * Colormap.sph_raw_output_to_image drives (implementation.py:132-201). */
int tsp_colormap_scalar_host(tsp_context *ctx, const float *img, int H, int W, int C,
                             const float *lut_rgba, int n_lut, float vmin, float vmax, int log_scale,
                             int weighted, uint8_t *out_rgba);
int tsp_colormap_rgb_host(tsp_context *ctx, const float *img, int H, int W, int C, float vmin,
                          float vmax, float gamma, uint8_t *out_rgba8, float *out_rgba_f32);

/* Periodic tiling post-pass: replaces the render target by the weighted sum
 * of n shifted copies of itself -- PeriodicSPH.render + PeriodicSPHAccumulationOverlay (reference
 * src/topsy/periodic_sph.py:36-88, shaders/overlay.wgsl:18-51): one full-viewport quad per instance
 * displaced by offsets_xy[k] (clip units, +y up), sampled with a linear filter and clamp-to-edge,
 * additively blended into a cleared target.  The float64 accumulator keeps the untiled render, so a
 * later tsp_render(clear = 0) continues from the raw image and the tiling is re-applied afterwards. */
int tsp_tile_periodic(tsp_context *ctx, int n, const float *offsets_xy, const float *weights);

/* On-device autorange support (replaces the image read-back + host
 * np.percentile of Colormap.autorange_vmin_vmax / _autorange_using_values, reference
 * src/topsy/colormap/implementation.py:381-425, and RGBColormap.autorange_vmin_vmax :512-531).
 * tsp_content_sort computes the logical content of the render target scaled by `scale` in the same
 * float32 arithmetic numpy uses on the host (kind 0: ch0*scale; 1: (ch1*scale)/(ch0*scale);
 * 2: the three colour channels of an rgb image, flattened; 3: every channel of the image, flattened --
 * what the reference's RGBColormap.autorange_vmin_vmax sees, since it ravel()s the raw 4-channel image
 * including the fragment-count channel), sorts the FINITE values on the device
 * and reports how many there are and how many of them are <= 0.  tsp_content_values then returns
 * the values at the given ranks of that ascending order (the host needs only a handful: min, max,
 * and the neighbours of each percentile's virtual index). */
int tsp_content_sort(tsp_context *ctx, int kind, float scale, int64_t *n_finite, int64_t *n_nonpositive);
int tsp_content_values(tsp_context *ctx, const int64_t *ranks, int n_ranks, float *out);

/* Counters of the last tsp_render call (measurement aid). */
typedef struct {
    int64_t n_particles;   /* particles visited (sum of range lengths) */
    int64_t n_small;       /* splatted by the streaming kernel */
    int64_t n_mid;         /* nearest-mip footprints deferred to kernel G (register gather over per-strip bins) */
    int64_t n_huge;        /* bilinear footprints (P >= 64 px) deferred to the tile-gather kernel */
    int64_t n_culled;      /* z-slab / off-screen / non-finite */
    int64_t n_fragments;   /* pixel updates (only counted when TSP_STATS is enabled) */
    double ms_stream, ms_mid, ms_huge, ms_total; /* per-kernel GPU time, hipEvents; ms_mid = kernel G + its binning passes; ms_huge = kernel H2 + its band fill */
    double ms_mega;        /* reserved (0): a second gather kernel existed in rounds 2-4 */
    int64_t n_mega;        /* reserved (0) */
    /* n_fragments by the kernel that drew them (counted like n_fragments; 0 on the generic pipeline): kernel S, kernel G,
     * kernel H2, reserved (0) -- what bench.py prices each kernel's fragment-rate roofline with */
    int64_t n_fragments_stream, n_fragments_mid, n_fragments_huge, n_fragments_mega;
    /* of n_culled: particles of chunks (512 consecutive particles) whose bounding box lay outside the view -- never read
     * (option "chunk_cull", on by default; needs >= 4096 chunks in the call, pays after tsp_reorder_spatial) */
    int64_t n_chunk_culled;
} tsp_stats;
int tsp_get_stats(tsp_context *ctx, tsp_stats *out);
/* Options by name.  "count_fragments" (0/1): fragment counting (adds atomics; off by default).  "use_quantity" (0/1): render
 * density-only without dropping the resident quantity.  "chunk_cull" (1/0), "reorder_interleave" (1/0, read by the next
 * tsp_reorder_spatial).  The remaining names are tuning and measurement aids of the pipeline ("p_small_milli", "huge_split",
 * "huge_variant", "huge_band_mib", "mid_item_records", "mid_item_scale_milli", "stream_blocks_per_cu", "stream_batch_chunks",
 * "overlap_mid_huge", "slice_records", "debug_*"; csrc/tsp_api.hip). */
int tsp_set_option(tsp_context *ctx, const char *name, int64_t value);

/* Streaming-read microbenchmark (float4 read-sum over `bytes` of device memory; best of a few launch shapes): returns GB/s.
 * The measured HBM peak BASELINE.md section 2 prices the roofline fraction against. */
int tsp_measure_read_bandwidth(tsp_context *ctx, int64_t bytes, int iters, double *gbps_out);

/* Multi-GPU: one process per GPU, particles sharded by index range, partial images summed
 * with ONE RCCL reduce over xGMI (the reference has no counterpart).
 * Rank 0 calls tsp_comm_unique_id and distributes the 128-byte id out-of-band. */
#define TSP_UNIQUE_ID_BYTES 128
int tsp_comm_unique_id(char *id_out);
int tsp_comm_init(tsp_context *ctx, int n_ranks, int rank, const char *id);
/* Sum-reduce the float32 render target to `root` (or to every rank when root < 0), in place.
 * Contract: every rank calls it exactly ONCE per frame, after the frame's last tsp_render.  The float64
 * accumulator stays rank-local; the reduced float32 image is a presentation copy (what tsp_read_image,
 * the colormap calls and autorange see), not an accumulator:
 *   - a second call without a tsp_render (or tsp_write_image) in between returns TSP_ESTATE instead of
 *     adding the other ranks' shares twice;
 *   - any later tsp_render -- including a REFINE block with clear = 0 -- rebuilds the float32 image from
 *     the rank's own accumulator, so the frame must be reduced again before it is presented.
 * On ranks other than `root` the image content after the call is unspecified (root >= 0). */
int tsp_comm_reduce_image(tsp_context *ctx, int root, double *gpu_ms_out);
int tsp_comm_destroy(tsp_context *ctx);
/* The same hand-over for a collective the CALLER performed (shards summed through the host: contexts that share a device,
 * which RCCL refuses, or no RCCL at all): `sum` (R * R * C float32) becomes the float32 presentation image of `ctx`; its
 * float64 accumulator keeps the context's own partial sums, so a later tsp_render with clear = 0 continues unrounded --
 * exactly the state tsp_comm_reduce_image leaves on the root.  Same once-per-frame rule (TSP_ESTATE on a second call). */
int tsp_set_reduced_image(tsp_context *ctx, const float *sum);

/* Several GPUs of one node behind ONE handle (SURVEY.md section 8b sketched `tsp_create(n_devices, device_ids, ...)`; the
 * reference has no counterpart -- its SplitBuffers, src/topsy/split_buffers.py:26-38,78-116, cuts one device's buffers the same
 * way).  A group is n_devices ordinary contexts plus the host-thread choreography: uploads are cut into the contiguous index
 * ranges [g N / G, (g + 1) N / G), tsp_group_render intersects the block's (start, len) ranges with every shard and runs the
 * G tsp_render calls concurrently (returns the slowest shard's GPU time), and tsp_group_end_frame is the frame's ONE sum-reduce
 * of the float32 image onto context 0 -- RCCL over xGMI when the device ids are distinct, a read-back / add of the float32
 * partial images through the host and tsp_set_reduced_image when two contexts share a device (RCCL refuses that; single-GPU test
 * boxes) or librccl cannot be loaded.  Everything that looks at the finished
 * frame (tsp_read_image, tsp_colormap_*, tsp_content_*, tsp_tile_periodic) is called on tsp_group_context(group, 0) after
 * tsp_group_end_frame; per-shard state can be inspected through tsp_group_context(group, g).  The reduce contract of
 * tsp_comm_reduce_image holds: end_frame reduces at most once per rendered frame, and a later tsp_group_render -- a REFINE
 * block with clear = 0 included -- continues from every shard's own float64 accumulator.  Calls on one group must be
 * serialised by the caller.  tsp_group_get_stats: counters summed over the shards, times of the slowest shard. */
typedef struct tsp_group tsp_group;
int tsp_group_create(int n_devices, const int *device_ids, int resolution, int n_channels, tsp_group **out);
void tsp_group_destroy(tsp_group *group);
int tsp_group_size(tsp_group *group);
tsp_context *tsp_group_context(tsp_group *group, int index);
int tsp_group_uses_rccl(tsp_group *group);
int tsp_group_set_kernel_mips(tsp_group *group, const float *lut, int n0, int n_levels);
int tsp_group_upload_particles(tsp_group *group, int64_t n, const float *x, const float *y, const float *z, const float *h,
                               const float *mass);
int tsp_group_upload_quantity(tsp_group *group, const float *q);
int tsp_group_upload_rgb(tsp_group *group, const float *r, const float *g, const float *b);
int tsp_group_generate_synthetic(tsp_group *group, int64_t n_total, int64_t first, int64_t count, uint64_t seed, float h_cap,
                                 int with_quantity, int with_rgb);
int tsp_group_reorder_spatial(tsp_group *group, int n_strata, uint64_t seed);
int64_t tsp_group_num_particles(tsp_group *group);
int tsp_group_set_option(tsp_group *group, const char *name, int64_t value);
int tsp_group_render(tsp_group *group, const float *M, float scale_factor, const int64_t *starts, const int64_t *lens,
                     int n_ranges, int clear, int mode, int flags, double *gpu_ms_out);
int tsp_group_end_frame(tsp_group *group, double *ms_out);
int tsp_group_get_stats(tsp_group *group, tsp_stats *out);
/* Shard `index` owns the global indices [*first_out, *first_out + *count_out) of the last upload / generate. */
int tsp_group_shard_range(tsp_group *group, int index, int64_t *first_out, int64_t *count_out);
/* tsp_upload_band_magnitudes for the group: mags is [n_bands][N] over the whole snapshot, every shard takes its columns. */
int tsp_group_upload_band_magnitudes(tsp_group *group, int n_bands, const double *mags, const double *weights);

#ifdef __cplusplus
}
#endif
#endif /* TOPSY_SPLAT_H */
