// tsp_pipeline.h -- what the translation units of the splat pipeline share: tile constants, the float64 atomic helpers,
// the argument block of the tile kernels, and the launcher of the tile-gather kernels (tsp_gather.hip).
#pragma once
#include "tsp_internal.h"

namespace tsp {


#ifndef TSP_S_BLOCK
#define TSP_S_BLOCK 256
#endif
#ifndef TSP_S_KPT
#define TSP_S_KPT 2
#endif
constexpr int SBLOCK = TSP_S_BLOCK;  // threads per workgroup of kernel S
constexpr int SWAVES = SBLOCK / 64;
constexpr int KPT = TSP_S_KPT;       // particles per thread per chunk
constexpr int CHUNK = SBLOCK * KPT;  // particles per chunk
// LDS accumulators are DOUBLE: on gfx950 a conflict-free ds_add_f64 costs ~9 clk per wave-instruction
// (~11 clk more per extra lane on the same address) while ds_add_f32 costs ~190 (measured,
// tools/ubench/lds_partial.hip, lds_atomics.hip), and the sums gain precision.
#ifndef TSP_WIN1
#define TSP_WIN1 60
#define TSP_WIN2 44
#define TSP_WIN4 40
#endif
#ifndef TSP_S_OCC
#define TSP_S_OCC 5
#endif
// edge of kernel S's LDS window by the number of channels it holds (1: density, 2: weighted / depth, 4: rgb)
template <int WC> struct WinSize { static constexpr int value = (WC == 1) ? TSP_WIN1 : (WC == 2 ? TSP_WIN2 : TSP_WIN4); };

constexpr int HBAND_H = 64;          // image rows per band of the huge-record bins (kernel H2's tallest tile)

enum { CLS_NONE = 0, CLS_SMALL = 1, CLS_MID = 2, CLS_HUGE = 3 };

// the render target is accumulated in float64 (global_atomic_add_f64) and rounded to float32 once per
// tsp_render call, so cross-workgroup summation adds no float32 noise however many flushes hit a pixel
__device__ __forceinline__ void gatomic_add(double *addr, double v) {
#ifdef TSP_DEBUG_NO_FLUSH      // measurement aid: what the float64 flush atomics cost (the image is then empty)
    if (v == 123.456) *addr = v;
#else
    __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// top-left quadrants of the four mip levels of a mirror-symmetric kernel image (32^2 + 16^2 + 8^2 + 4^2 floats)
constexpr int MIPQ_TOTAL = 1024 + 256 + 64 + 16;
__device__ __forceinline__ int mipq_offset(int lvl) { return lvl == 0 ? 0 : (lvl == 1 ? 1024 : (lvl == 2 ? 1280 : 1344)); }

// DPP modifier: every lane reads the operand from lane t of its own quad
#define TSP_DPP_QUAD(t) "quad_perm:[" #t "," #t "," #t "," #t "] row_mask:0xf bank_mask:0xf"

__device__ __forceinline__ void latomic_add(double *addr, float v) {
    __hip_atomic_fetch_add(addr, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifndef TSP_FOLD_EVERY
#define TSP_FOLD_EVERY 2048    // footprints a float32 accumulator of kernels G / H2 holds before it goes to the float64 target (512 until the end of round 5)
#endif
#ifndef TSP_HDEAL
#define TSP_HDEAL 16
#endif
constexpr int HDEAL = TSP_HDEAL;            // records per dealing run of the tile-gather kernels (1: 13.6 ms, 4: 10.67, 16: 10.49, 64: 10.71 for H2)

struct TileArgs {
    const float4 *geom; const float *w;
    long long n_records;
    Camera cam;
    const float *mips;
    double *img;
    Counters *cnt;
    int tiles_x, split;
    int count_frag;
    float disc_k2;     // (0.5235)^2 when the LUT is zero outside the inscribed disc (exact corner culling), else 0
    // kernel H2: the huge records binned by image band (huge_band_fill_kernel): band b (rows [b, b + 1) * HBAND_H) holds
    // hband_count[b] records at geom + b * hband_stride (w likewise); nullptr = one list for every tile (geom, n_records)
    // (kernel G: exact-size bins, band b starts at record hband_base[b])
    const int *hband_count; long long hband_stride; const long long *hband_base;
    // kernel G: the mid records binned by tile (hband_count[t] records from record hband_base[t] on); workgroup i draws work item i =
    // GCHUNK consecutive records of tile item_tile[i]'s bin (that tile's items start at item_base[tile]; item_base[n_tiles] = their number)
    int n_tiles; const int *item_tile; const int *item_base; int item_records;      // item_records: records per work item (a power of two)
};

// Kernel H2 (tsp_gather.hip) for the footprints >= 64 px of one render block (the records kernel S appended to the huge list).
// Records ctx->ev[10] after the launch (per-kernel time: ev[9] .. ev[10]).
int launch_gather_kernels(tsp_context *ctx, TileArgs ta, int mode, bool second_channel, const float4 *huge_geom, const float *huge_w,
                          long long n_huge);

// Kernel G (tsp_gather.hip): the MID records of one render block as a register gather; launched on `st`.
int launch_mid_gather(tsp_context *ctx, TileArgs ta, int mode, bool second_channel, const float4 *mid_geom, const float *mid_w,
                      long long n_mid, hipStream_t st);

}  // namespace tsp
