// tsp_pipeline.hip -- the three-class splat pipeline (default render path), gfx950.
//
// What it computes is exactly vertex_* + fragment_* + additive blend of the reference
// (src/topsy/shaders/sph.wgsl:54-91,139-165; src/topsy/sph.py:31-42) in the canonical arithmetic of
// tsp_math.h.  How it is scheduled is MI355X-specific and driven by the footprint distribution
// (DESIGN.md section 4): 80 % of particles cover < 8 px but > 90 % of all pixel updates come from the
// ~1 % of particles wider than 64 px.
//
//   kernel S  splat_stream_kernel  streams the SoA particle arrays once (coalesced, 512-particle
//             chunks, ~10 consecutive chunks per workgroup so load-time spatial order gives screen
//             locality).  Footprints < p_small px that fit the 64x64-pixel LDS window following the
//             chunks are rasterised at once (ds_add_f64), flushed with one global atomic per touched
//             pixel.  All other footprints are not rasterised here: their projected records
//             (pcx, pcy, P, weights) are appended to the MID list (per-chunk contiguous segments with
//             a pixel bounding box) or the HUGE list.
//   kernel M  splat_mid_kernel     one workgroup per (64x32 image tile, split): walks the segments
//             whose bbox meets the tile; every lane prepares one record (tile-clipped pixel ranges,
//             mip level), then each wave rasterises its records one at a time, parameters broadcast
//             into scalar registers, 8x8 lanes per step, nearest-mip sampling from an LDS copy of the
//             mip pyramid, ds_add_f64 into the LDS tile (row stride 72: distinct addresses per step).
//   kernel H  splat_huge_kernel    one workgroup per (128x64 tile, split): every lane owns a 4x4 pixel
//             block in registers (no atomics in the loop); records overlapping the tile are compacted
//             into an LDS queue and evaluated by the 8 waves (64x16-pixel strips) with bilinear sampling
//             from an LDS "quad table" (one ds_read_b128 fetches the 2x2 texel stencil of a pixel).
//
// All three add into the float64 render target with device-scope atomics only at flush time.
#include <string.h>

#include <algorithm>
#include <vector>

#include "tsp_internal.h"

namespace tsp {

constexpr int CHUNK = 512;           // particles per chunk
constexpr int KPT = CHUNK / 256;     // particles per thread per chunk
constexpr int TILE = 64;             // image tile edge of kernel H (and tile width of kernel M)
constexpr int MTILE_H = 32;          // tile height of kernel M (64 x 32 pixels per workgroup)
constexpr int MSTR = 72;             // LDS row stride (doubles) of the mid tile
// LDS accumulators are DOUBLE: on gfx950 a conflict-free ds_add_f64 costs ~9 clk per wave-instruction
// (~11 clk more per extra lane on the same address) while ds_add_f32 costs ~190 (measured,
// tools/ubench/lds_partial.hip, lds_atomics.hip), and the sums gain precision.
template <int MODE> struct WinSize { static constexpr int value = (MODE == TSP_MODE_RGB) ? 48 : 64; };

enum { CLS_NONE = 0, CLS_SMALL = 1, CLS_MID = 2, CLS_HUGE = 3, CLS_MEGA = 4 };

// the render target is accumulated in float64 (global_atomic_add_f64) and rounded to float32 once per
// tsp_render call, so cross-workgroup summation adds no float32 noise however many flushes hit a pixel
__device__ __forceinline__ void gatomic_add(double *addr, double v) {
    __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void latomic_add(double *addr, float v) {
    __hip_atomic_fetch_add(addr, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------------------------------------
// block-wide helpers (256 threads = 4 waves)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

struct StreamArgs {
    Particles p;
    const int64_t *ranges;     // starts[n] | lens[n] | chunk_prefix[n+1]
    int n_ranges;
    int n_chunks;
    int chunks_per_block;
    Camera cam;
    const float *mips;
    double *img;
    float4 *mid_geom;  float *mid_w;   int64_t mid_capacity;
    float4 *huge_geom; float *huge_w;  int64_t huge_capacity;
    int *seg_count; long long *seg_offset; float4 *seg_bbox;
    Counters *cnt;
    float p_small;
    float p_mega;              // footprints at least this wide are appended from the END of the huge list (kernel H3's share)
    int count_frag;
    int emit_small;            // 0: records only (replay after a record-list overflow)
};

// WC = channels kept in the LDS window: 1 for a density-only render (channel 1 is identically 0:
// half the LDS and half the atomics), else the image's channel count.
template <int MODE, int WC>
__global__ __launch_bounds__(256, 5) void splat_stream_kernel(StreamArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;      // extra weights per record
    constexpr int WIN = WinSize<MODE>::value;
    extern __shared__ __attribute__((aligned(16))) double smem_d[];
    double *win = smem_d;                                            // [WC][WIN*WIN]
    float *T3 = reinterpret_cast<float *>(win + WC * WIN * WIN);     // mip level 3: 8x8
    __shared__ float s_red[4][4], s_mbb[4][4];
    __shared__ int s_cnt[4][2];
    __shared__ long long s_base[2];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const Camera &cam = a.cam;
    const int R = cam.R;
    for (int i = tid; i < WC * WIN * WIN; i += 256) win[i] = 0.0;
    if (tid < 64) T3[tid] = a.mips[5376 + tid];
    __syncthreads();

    // window state (uniform): origin and the dirty rectangle (window coordinates, inclusive)
    int wox = 0, woy = 0;
    int dx0 = WIN, dy0 = WIN, dx1 = -1, dy1 = -1;
    unsigned long long n_small = 0, n_cull = 0, n_frag = 0;

    const int c_begin = blockIdx.x * a.chunks_per_block;
    const int c_end = min(c_begin + a.chunks_per_block, a.n_chunks);
    const int64_t *starts = a.ranges, *lens = a.ranges + a.n_ranges, *cprefix = a.ranges + 2 * a.n_ranges;

    auto flush = [&]() {
        // one global atomic per touched pixel and channel; only the dirty rectangle is visited
        if (dx1 >= dx0) {
            const int fw = dx1 - dx0 + 1, fn = fw * (dy1 - dy0 + 1);
            for (int idx = tid; idx < fn; idx += 256) {
                const int jj = idx / fw;
                const int wy = dy0 + jj, wx = dx0 + (idx - jj * fw);
                const int gx = wox + wx, gy = woy + wy;
                const int o = wy * WIN + wx;
#pragma unroll
                for (int c = 0; c < WC; ++c) {
                    const double v = win[c * WIN * WIN + o];
                    if (v != 0.0) {
                        if (gx < R && gy < R) gatomic_add(a.img + ((size_t)gy * R + gx) * C + c, v);
                        win[c * WIN * WIN + o] = 0.0;
                    }
                }
            }
        }
        dx0 = WIN; dy0 = WIN; dx1 = -1; dy1 = -1;
    };

    for (int c = c_begin; c < c_end; ++c) {
        // ---- locate the chunk: range r, particles [first, first + cnt) -------------------------
        int r = 0;
        if (a.n_ranges > 1) {
            int lo = 0, hi = a.n_ranges - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cprefix[mid] <= c) lo = mid; else hi = mid - 1;
            }
            r = lo;
        }
        const int64_t in_range = (int64_t)(c - cprefix[r]) * CHUNK;
        const int64_t first = starts[r] + in_range;
        const int cnt = (int)min((int64_t)CHUNK, lens[r] - in_range);

        // ---- phase 1: coalesced loads, projection, classification ------------------------------
        float pcx[KPT], pcy[KPT], PP[KPT], w0[KPT], w1[KPT], w2[KPT];
        int cls[KPT];
        int xr[KPT], yr[KPT];          // small footprints: first covered pixel | (number of covered pixels << 16)
        float bx0 = 3.0e38f, by0 = 3.0e38f, bx1 = -3.0e38f, by1 = -3.0e38f;      // small footprints
        float mx0 = 3.0e38f, my0 = 3.0e38f, mx1 = -3.0e38f, my1 = -3.0e38f;      // mid footprints
        int my_mid = 0, my_huge = 0;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int li = k * 256 + tid;
            cls[k] = CLS_NONE;
            pcx[k] = pcy[k] = PP[k] = w0[k] = w1[k] = w2[k] = 0.0f;
            xr[k] = yr[k] = 0;
            if (li < cnt) {
                const int64_t i = first + li;
                const float h = a.p.h[i];
                const Proj pr = project(cam, a.p.x[i], a.p.y[i], a.p.z[i], h);
                bool vis = false;
                if (pr.keep) {
                    // any pixel centre covered?  (exact test via the canonical interval)
                    int ilo, ihi, jlo, jhi;
                    cover_range(pr.pcx, pr.half, R, ilo, ihi);
                    cover_range(pr.pcy, pr.half, R, jlo, jhi);
                    vis = (ilo <= ihi) && (jlo <= jhi);
                    // a small footprint covers <= 12 pixels per axis and R <= 16384: both fit 16 bits
                    xr[k] = ilo | (min(ihi - ilo + 1, 0xffff) << 16);
                    yr[k] = jlo | (min(jhi - jlo + 1, 0xffff) << 16);
                }
                if (vis) {
                    const float hh = h * h;
                    if (MODE == TSP_MODE_RGB) {
                        w0[k] = a.p.r[i] / hh; w1[k] = a.p.g[i] / hh; w2[k] = a.p.b[i] / hh;
                    } else {
                        w0[k] = a.p.m[i] / hh;
                        w1[k] = (MODE == TSP_MODE_DEPTH) ? pr.cz : (a.p.q ? a.p.q[i] : 0.0f);
                    }
                    pcx[k] = pr.pcx; pcy[k] = pr.pcy; PP[k] = pr.P;
                    if (pr.P < a.p_small) {
                        cls[k] = CLS_SMALL;
                        bx0 = fminf(bx0, pr.pcx - pr.half); bx1 = fmaxf(bx1, pr.pcx + pr.half);
                        by0 = fminf(by0, pr.pcy - pr.half); by1 = fmaxf(by1, pr.pcy + pr.half);
                    } else if (pr.P < P_BILINEAR) {
                        cls[k] = CLS_MID; ++my_mid;
                        mx0 = fminf(mx0, pr.pcx - pr.half); mx1 = fmaxf(mx1, pr.pcx + pr.half);
                        my0 = fminf(my0, pr.pcy - pr.half); my1 = fmaxf(my1, pr.pcy + pr.half);
                    } else if (pr.P < a.p_mega) {
                        cls[k] = CLS_HUGE; ++my_huge;
                    } else {
                        cls[k] = CLS_MEGA;          // rare (~1e-3 of the particles): one atomic each, below
                    }
                } else {
                    ++n_cull;
                }
            }
        }

        // ---- phase 2: the chunk's small-footprint bounding box places the LDS window (uniform) ------
        // (each exchange has its own LDS scratch, so one barrier per exchange suffices: the barriers of
        // the following exchanges order its readers before the next chunk's writers)
        bx0 = wave_min(bx0); by0 = wave_min(by0); bx1 = wave_max(bx1); by1 = wave_max(by1);
        if (lane == 0) { s_red[wv][0] = bx0; s_red[wv][1] = by0; s_red[wv][2] = bx1; s_red[wv][3] = by1; }
        __syncthreads();
        bx0 = fminf(fminf(s_red[0][0], s_red[1][0]), fminf(s_red[2][0], s_red[3][0]));
        by0 = fminf(fminf(s_red[0][1], s_red[1][1]), fminf(s_red[2][1], s_red[3][1]));
        bx1 = fmaxf(fmaxf(s_red[0][2], s_red[1][2]), fmaxf(s_red[2][2], s_red[3][2]));
        by1 = fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3]));
        if (bx1 >= bx0) {
            const int ix0 = max((int)__builtin_floorf(bx0), 0), ix1 = min((int)__builtin_floorf(bx1), R - 1);
            const int iy0 = max((int)__builtin_floorf(by0), 0), iy1 = min((int)__builtin_floorf(by1), R - 1);
            const bool inside = ix0 >= wox && ix1 < wox + WIN && iy0 >= woy && iy1 < woy + WIN;
            if (!inside) {
                flush();
                // centre the window on the chunk's small-footprint bounding box
                wox = (ix0 + ix1 + 1 - WIN) / 2;
                woy = (iy0 + iy1 + 1 - WIN) / 2;
                wox = max(0, min(wox, R - WIN));
                woy = max(0, min(woy, R - WIN));
                wox = max(wox, 0); woy = max(woy, 0);
            }
            // dirty rectangle grows by this chunk's bbox (clipped to the window)
            dx0 = min(dx0, max(ix0 - wox, 0)); dx1 = max(dx1, min(ix1 - wox, WIN - 1));
            dy0 = min(dy0, max(iy0 - woy, 0)); dy1 = max(dy1, min(iy1 - woy, WIN - 1));
        }

        // ---- phase 3: a small footprint that does not fit the window joins the MID list (kernel M
        //      rasterises any width with the same nearest-mip rule), so phase 4 is LDS-only -----------
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if (cls[k] != CLS_SMALL) continue;
            const int ilo = xr[k] & 0xffff, ihi = ilo + (xr[k] >> 16) - 1;
            const int jlo = yr[k] & 0xffff, jhi = jlo + (yr[k] >> 16) - 1;
            if (ilo < wox || ihi >= wox + WIN || jlo < woy || jhi >= woy + WIN) {
                cls[k] = CLS_MID; ++my_mid;
                const float half = 0.5f * PP[k];
                mx0 = fminf(mx0, pcx[k] - half); mx1 = fmaxf(mx1, pcx[k] + half);
                my0 = fminf(my0, pcy[k] - half); my1 = fmaxf(my1, pcy[k] + half);
            }
        }
        // record counts + offsets and the bounding box of the chunk's MID footprints
        const int mid_incl = wave_incl_scan(my_mid, lane), huge_incl = wave_incl_scan(my_huge, lane);
        const unsigned long long any_mid = __ballot(my_mid > 0);
        if (any_mid) { mx0 = wave_min(mx0); my0 = wave_min(my0); mx1 = wave_max(mx1); my1 = wave_max(my1); }
        if (lane == 63) { s_cnt[wv][0] = mid_incl; s_cnt[wv][1] = huge_incl; }
        if (lane == 0) { s_mbb[wv][0] = mx0; s_mbb[wv][1] = my0; s_mbb[wv][2] = mx1; s_mbb[wv][3] = my1; }
        __syncthreads();
        int mid_before = 0, huge_before = 0, mid_total = 0, huge_total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wv) { mid_before += s_cnt[w][0]; huge_before += s_cnt[w][1]; }
            mid_total += s_cnt[w][0]; huge_total += s_cnt[w][1];
        }
        // reserve contiguous runs in the record lists (one atomic per chunk and list)
        if (tid == 0) {
            s_base[0] = mid_total ? (long long)atomicAdd(&a.cnt->n_mid, (unsigned long long)mid_total) : 0;
            s_base[1] = huge_total ? (long long)atomicAdd(&a.cnt->n_huge, (unsigned long long)huge_total) : 0;
            a.seg_count[c] = mid_total;
            a.seg_offset[c] = s_base[0];
            if (mid_total)
                a.seg_bbox[c] = make_float4(fminf(fminf(s_mbb[0][0], s_mbb[1][0]), fminf(s_mbb[2][0], s_mbb[3][0])),
                                            fminf(fminf(s_mbb[0][1], s_mbb[1][1]), fminf(s_mbb[2][1], s_mbb[3][1])),
                                            fmaxf(fmaxf(s_mbb[0][2], s_mbb[1][2]), fmaxf(s_mbb[2][2], s_mbb[3][2])),
                                            fmaxf(fmaxf(s_mbb[0][3], s_mbb[1][3]), fmaxf(s_mbb[2][3], s_mbb[3][3])));
        }
        __syncthreads();
        const long long mid_base = s_base[0], huge_base = s_base[1];

        // ---- phase 4: rasterise small footprints (one lane per particle, mip 3 nearest) ---------
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if (cls[k] != CLS_SMALL || !a.emit_small) continue;
            const float half = 0.5f * PP[k], invP = 1.0f / PP[k];
            const int ilo = xr[k] & 0xffff, ihi = ilo + (xr[k] >> 16) - 1;
            const int jlo = yr[k] & 0xffff, jhi = jlo + (yr[k] >> 16) - 1;
            for (int j = jlo; j <= jhi; ++j) {
                const float dy = ((float)j + 0.5f) - pcy[k];
                const int ty = nearest_index((dy + half) * invP, 8);
                double *drow = win + (j - woy) * WIN - wox;
                for (int i = ilo; i <= ihi; ++i) {
                    const float dx = ((float)i + 0.5f) - pcx[k];
                    const int tx = nearest_index((dx + half) * invP, 8);
                    const float kv = T3[ty * 8 + tx];
                    double *d = drow + i;
                    if (MODE == TSP_MODE_RGB) {
                        latomic_add(d, kv * w0[k]); latomic_add(d + WIN * WIN, kv * w1[k]);
                        latomic_add(d + 2 * WIN * WIN, kv * w2[k]); latomic_add(d + 3 * WIN * WIN, 1.0f);
                    } else {
                        if (kv == 0.0f) continue;      // corner texels are exactly 0: adding +-0 changes nothing
                        const float val = kv * w0[k];
                        latomic_add(d, val);
                        if (WC > 1) latomic_add(d + WIN * WIN, val * w1[k]);
                    }
                }
            }
            ++n_small;
            if (a.count_frag) n_frag += (unsigned long long)((ihi - ilo + 1) * (jhi - jlo + 1));
        }

        // ---- phase 5: append the deferred footprints -----------------------------------------------
        {
            long long mpos = mid_base + mid_before + (mid_incl - my_mid);
            long long hpos = huge_base + huge_before + (huge_incl - my_huge);
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                if (cls[k] == CLS_MID) {
                    if (mpos < a.mid_capacity) {
                        a.mid_geom[mpos] = make_float4(pcx[k], pcy[k], PP[k], w0[k]);
                        a.mid_w[mpos * NW] = w1[k];
                        if (NW == 2) a.mid_w[mpos * NW + 1] = w2[k];
                    }
                    ++mpos;
                } else if (cls[k] == CLS_HUGE) {
                    if (hpos < a.huge_capacity) {
                        a.huge_geom[hpos] = make_float4(pcx[k], pcy[k], PP[k], w0[k]);
                        a.huge_w[hpos * NW] = w1[k];
                        if (NW == 2) a.huge_w[hpos * NW + 1] = w2[k];
                    }
                    ++hpos;
                } else if (cls[k] == CLS_MEGA) {
                    // the mega records grow downwards from the end of the huge list: one allocation, two cursors
                    // (an overlap with the records growing from the front means overflow: the host sees
                    // n_huge + n_mega > capacity, enlarges the list and replays)
                    const long long mp = a.huge_capacity - 1 - (long long)atomicAdd(&a.cnt->n_mega, 1ull);
                    if (mp >= 0) {
                        a.huge_geom[mp] = make_float4(pcx[k], pcy[k], PP[k], w0[k]);
                        a.huge_w[mp * NW] = w1[k];
                        if (NW == 2) a.huge_w[mp * NW + 1] = w2[k];
                    }
                }
            }
        }
    }
    __syncthreads();
    flush();
    // statistics (one atomic per wave)
    for (int o = 32; o; o >>= 1) {
        n_small += __shfl_xor((long long)n_small, o);
        n_cull += __shfl_xor((long long)n_cull, o);
        n_frag += __shfl_xor((long long)n_frag, o);
    }
    if (lane == 0) {
        if (n_small) atomicAdd(&a.cnt->n_small, n_small);
        if (n_cull) atomicAdd(&a.cnt->n_culled, n_cull);
        if (n_frag) atomicAdd(&a.cnt->n_fragments, n_frag);
    }
}

// ---------------------------------------------------------------------------------------------
// kernel M: mid footprints (p_small <= P < 64), tile scatter with nearest-mip sampling
// ---------------------------------------------------------------------------------------------
struct TileArgs {
    const float4 *geom; const float *w;
    long long n_records;
    const int *seg_count; const long long *seg_offset; const float4 *seg_bbox; int n_chunks;
    Camera cam;
    const float *mips;
    double *img;
    Counters *cnt;
    int tiles_x, split;
    int count_frag;
    float disc_k2;     // (0.5235)^2 when the LUT is zero outside the inscribed disc (exact corner culling), else 0
    float p_lo, p_hi;  // kernels H2 / H3 take the records with p_lo <= P < p_hi
};

constexpr int MT = 512;              // threads per workgroup of kernel M (8 waves share tile + LUT)

// WC = channels accumulated in the LDS tile (1 for a density-only render, else the image's channel count)
template <int MODE, int WC>
__global__ __launch_bounds__(MT) void splat_mid_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) double smem_d[];
    double *tile = smem_d;                                                   // [WC][MTILE_H][MSTR]
    float *T = reinterpret_cast<float *>(tile + WC * MTILE_H * MSTR);        // mip pyramid, 5440 floats
    __shared__ long long s_seg_off[MT];
    __shared__ int s_seg_cnt[MT];
    __shared__ int s_wcnt[MT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R = a.cam.R;
    const int tile_id = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tx0 = (tile_id % a.tiles_x) * TILE, ty0 = (tile_id / a.tiles_x) * MTILE_H;
    const float fx0 = (float)tx0, fy0 = (float)ty0, fx1 = (float)(tx0 + TILE), fy1 = (float)(ty0 + MTILE_H);
    for (int i = tid; i < MIP_TOTAL; i += MT) T[i] = a.mips[i];
    for (int i = tid; i < WC * MTILE_H * MSTR; i += MT) tile[i] = 0.0;
    __syncthreads();
    const int lx = lane & 7, ly = lane >> 3;
    unsigned long long n_frag = 0;
    bool touched = false;

    // segment headers are examined 256 at a time (one per lane).  Consecutive segments are spatial
    // neighbours, so they are dealt to the `split` workgroups of this tile with stride `split`:
    // every workgroup sees a uniform sample of the tile's segments (an even share of its work)
    for (int sbase = 0; sbase * a.split < a.n_chunks; sbase += MT) {
        const int seg = (sbase + tid) * a.split + sp;
        bool shit = false;
        int scnt = 0;
        if (seg < a.n_chunks) {
            scnt = a.seg_count[seg];
            if (scnt > 0) {
                const float4 bb = a.seg_bbox[seg];
                shit = bb.x < fx1 && bb.z > fx0 && bb.y < fy1 && bb.w > fy0;
            }
        }
        const unsigned long long smask = __ballot(shit);
        const int sbefore = __popcll(smask & ((1ull << lane) - 1ull));
        __syncthreads();                       // previous batch's segment list is no longer read
        if (lane == 0) s_wcnt[wv] = __popcll(smask);
        __syncthreads();
        int wbase = 0, nseg = 0;
#pragma unroll
        for (int w = 0; w < MT / 64; ++w) {
            if (w < wv) wbase += s_wcnt[w];
            nseg += s_wcnt[w];
        }
        if (shit) {
            s_seg_off[wbase + sbefore] = a.seg_offset[seg];
            s_seg_cnt[wbase + sbefore] = scnt;
        }
        __syncthreads();
        for (int sidx = 0; sidx < nseg; ++sidx) {
            const long long off = s_seg_off[sidx];
            const int cnt = s_seg_cnt[sidx];
            for (int base = 0; base < cnt; base += MT) {
                // records are dealt to the waves round-robin (wave w takes records w, w + 8, ...): a segment
                // holds ~100 records on average, and handing them out in runs of 64 would leave six of the
                // eight waves idle
                const int li = base + lane * (MT / 64) + wv;
                // ---- per lane: one record, its tile-clipped pixel ranges and mip level ----------------
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                float gw1 = 0.f, gw2 = 0.f, g_half = 0.f, g_invP = 0.f;
                int packed = 0;
                bool hit = false;
                if (li < cnt && off + li < a.n_records) {
                    g = a.geom[off + li];
                    g_half = 0.5f * g.z;
                    if ((g.x + g_half > fx0) && (g.x - g_half < fx1) && (g.y + g_half > fy0) && (g.y - g_half < fy1)) {
                        int ilo, ihi, jlo, jhi;
                        cover_range(g.x, g_half, R, ilo, ihi);
                        cover_range(g.y, g_half, R, jlo, jhi);
                        ilo = max(ilo, tx0) - tx0; ihi = min(ihi, tx0 + TILE - 1) - tx0;
                        jlo = max(jlo, ty0) - ty0; jhi = min(jhi, ty0 + MTILE_H - 1) - ty0;
                        if (ilo <= ihi && jlo <= jhi) {
                            hit = true;
                            gw1 = a.w[(off + li) * NW];
                            if (NW == 2) gw2 = a.w[(off + li) * NW + 1];
                            g_invP = 1.0f / g.z;
                            const int lvl = max(level_for(g.z), 0);
                            packed = ilo | (ihi << 6) | (jlo << 12) | (jhi << 18) | (lvl << 24);
                        }
                    }
                }
                unsigned long long mask = __ballot(hit);
                touched = touched || (mask != 0ull);
                // ---- per wave: one footprint at a time, its parameters broadcast into scalar registers ----
                while (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const float q_pcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), src));
                    const float q_pcy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.y), src));
                    const float q_half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_half), src));
                    const float q_invP = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_invP), src));
                    const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.w), src));
                    const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gw1), src));
                    const float w2 = (NW == 2) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gw2), src)) : 0.0f;
                    const int pk = __builtin_amdgcn_readlane(packed, src);
                    const int ilo = pk & 63, ihi = (pk >> 6) & 63, jlo = (pk >> 12) & 63, jhi = (pk >> 18) & 63, lvl = pk >> 24;
                    const int n = 64 >> lvl, toff = mip_offset(lvl);
                    // texel row of this lane's pixel row in each 8-row block of the footprint (the tile is 32 rows: <= 4 blocks)
                    int trow[MTILE_H / 8];
#pragma unroll
                    for (int rb = 0; rb < MTILE_H / 8; ++rb) {
                        trow[rb] = -1;
                        if (jlo + 8 * rb <= jhi) {
                            const int j = jlo + 8 * rb + ly;
                            const float dy = ((float)(ty0 + j) + 0.5f) - q_pcy;
                            const int ty = nearest_index((dy + q_half) * q_invP, n);
                            trow[rb] = (j <= jhi) ? toff + ty * n : -1;
                        }
                    }
                    for (int ib = ilo; ib <= ihi; ib += 8) {
                        const int i = ib + lx;
                        const float dx = ((float)(tx0 + i) + 0.5f) - q_pcx;
                        const int tx = nearest_index((dx + q_half) * q_invP, n);
                        double *dcol = tile + (jlo + ly) * MSTR + i;
#pragma unroll
                        for (int rb = 0; rb < MTILE_H / 8; ++rb) {
                            if (jlo + 8 * rb > jhi) break;
                            if (i <= ihi && trow[rb] >= 0) {
                                const float kv = T[trow[rb] + tx];
                                double *d = dcol + rb * 8 * MSTR;
                                if (MODE == TSP_MODE_RGB) {     // the counter channel is summed by add_rect_counts()
                                    latomic_add(d, kv * w0); latomic_add(d + MTILE_H * MSTR, kv * w1);
                                    latomic_add(d + 2 * MTILE_H * MSTR, kv * w2);
                                } else {
                                    const float val = kv * w0;
                                    latomic_add(d, val);
                                    if (WC > 1) latomic_add(d + MTILE_H * MSTR, val * w1);
                                }
                            }
                        }
                    }
                    if (a.count_frag && lane == 0) n_frag += (unsigned long long)((ihi - ilo + 1) * (jhi - jlo + 1));
                }
            }
        }
    }
    const int any = __syncthreads_or(touched ? 1 : 0);
    if (any) {
        for (int idx = tid; idx < TILE * MTILE_H; idx += MT) {
            const int wy = idx / TILE, wx = idx % TILE;
            const int gx = tx0 + wx, gy = ty0 + wy;
            if (gx < R && gy < R) {
#pragma unroll
                for (int c = 0; c < WC; ++c) {
                    const double v = tile[c * MTILE_H * MSTR + wy * MSTR + wx];
                    if (v != 0.0) gatomic_add(a.img + ((size_t)gy * R + gx) * C + c, v);
                }
            }
        }
    }
    if (a.count_frag && n_frag) atomicAdd(&a.cnt->n_fragments, n_frag);
}

// ---------------------------------------------------------------------------------------------
// kernel H: huge footprints (P >= 64 px), tile gather with bilinear sampling
// ---------------------------------------------------------------------------------------------
// v_mov_b32_dpp: read a value from another lane of the same 16-lane row (no LDS traffic)
template <int N> __device__ __forceinline__ int dpp_row_ror(int v) {          // lane i reads lane (i - N) mod 16 of its row
    return __builtin_amdgcn_mov_dpp(v, 0x120 + N, 0xf, 0xf, true);
}
template <int N> __device__ __forceinline__ float dpp_row_ror(float v) { return __int_as_float(dpp_row_ror<N>(__float_as_int(v))); }
template <int T> __device__ __forceinline__ int dpp_quad_bcast(int v) {       // every lane of a quad reads the quad's lane T
    return __builtin_amdgcn_mov_dpp(v, T * 0x55, 0xf, 0xf, true);
}
template <int T> __device__ __forceinline__ float dpp_quad_bcast(float v) { return __int_as_float(dpp_quad_bcast<T>(__float_as_int(v))); }

constexpr int HT = 512;              // threads per workgroup of kernel H (8 waves share one quad table)
constexpr int HTILE_W = 128;         // its tile is 128 pixels wide: 32 lanes x 4 pixels
constexpr int HDEAL = 4;             // records per dealing run of kernel H

// NACC = value channels accumulated (1: density only, 2: density + weighted/depth, 3: rgb);
// PXH  = pixel rows per lane (4 or 8): the per-axis setup (12 instructions per row/column) is shared
//        by 4*PXH pixels, so the taller block costs ~30 % fewer instructions per pixel; it is used
//        when the accumulators still fit the 128-VGPR budget of a 512-thread workgroup (NACC == 1).
template <int MODE, int NACC, int PXH>
__global__ __launch_bounds__(HT, 4) void splat_huge_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    constexpr int NPX = 4 * PXH;
    constexpr int HTILE_H = 16 * PXH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // quad table: Q[j][i] = (T[j][i], T[j][i+1], T[j+1][i], T[j+1][i+1]) with +1 clamped to 63, so one
    // ds_read_b128 fetches the whole bilinear stencil; the 64-float4 row stride keeps the 16-lane
    // groups of ds_read_b128 on distinct 16-byte slots when neighbouring lanes step one texel
    float4 *Q = reinterpret_cast<float4 *>(smem);                 // [64][64]
    float4 *qg = Q + 64 * 64;                                     // queue: (pcx, pcy, half, 1/P)  [256]
    float4 *qw = qg + 256;                                        // queue: (w0, w1, w2, -)        [256]
    __shared__ int s_wcnt[4];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R = a.cam.R;
    const int tile_id = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tx0 = (tile_id % a.tiles_x) * HTILE_W, ty0 = (tile_id / a.tiles_x) * HTILE_H;
    const float fx0 = (float)tx0, fy0 = (float)ty0, fx1 = (float)(tx0 + HTILE_W), fy1 = (float)(ty0 + HTILE_H);
    for (int i = tid; i < 64 * 64; i += HT) {
        const int j = i >> 6, x = i & 63, j1 = min(j + 1, 63), x1 = min(x + 1, 63);
        Q[i] = make_float4(a.mips[j * 64 + x], a.mips[j * 64 + x1], a.mips[j1 * 64 + x], a.mips[j1 * 64 + x1]);
    }
    // A wave covers a 64 x 4*PXH pixel strip (16 x 4 lanes of 4 x PXH pixels) and the 8 waves tile the
    // 128 x 16*PXH tile 2 x 4: measured 5 % faster than full-width 128 x 2*PXH strips (more footprints
    // miss a strip entirely, and 64-256 pixel footprints fill the strips they do reach better)
    const int sx = tx0 + 64 * (wv & 1), sy = ty0 + 4 * PXH * (wv >> 1);
    // Lane layout inside the strip: the 16 lanes of a DPP row form a 4 x 4 grid of 4 x 4-pixel blocks (16 x 16
    // pixels; the wave's four rows sit side by side).  Lane (g, p) = (quad, position in quad) owns the block at
    // block-column p, block-row g, so the four lanes of a QUAD share their pixel ROWS and the four lanes at the
    // same quad position share their pixel COLUMNS.  Each lane evaluates ONE column coordinate and ONE row
    // coordinate per footprint and fetches the other three of each from its partners with v_mov_b32_dpp
    // (quad_perm broadcast for rows, row_ror:4k for columns) -- 2 + 21 moves instead of 8 evaluations of ~12
    // instructions.  Because a DPP rotation is relative, a lane's k-th column slot is pixel column (g - k) & 3.
    static_assert(PXH == 4, "the DPP sharing scheme is laid out for 4 x 4 pixels per lane");
    const int lg = (lane >> 2) & 3, lp = lane & 3;
    const int px0 = sx + 16 * (lane >> 4) + 4 * lp, py0 = sy + PXH * lg;
    const float sx0 = (float)sx, sx1 = (float)(sx + 64), sy0 = (float)sy, sy1 = (float)(sy + 4 * PXH);
    // pixel centre this lane evaluates itself: column slot 0 (= column lg) and row lp; +inf outside the image
    // so that it is never covered
    const float pxc_own = (px0 + lg < R) ? (float)(px0 + lg) + 0.5f : __builtin_inff();
    const float pyc_own = (py0 + lp < R) ? (float)(py0 + lp) + 0.5f : __builtin_inff();
    // Accumulation is two-level so the float32 error stays ~sqrt(run length) * 2^-24 instead of
    // sqrt(n): short runs in `acc`, folded into `tot` (PXH == 4) or, when the registers are needed
    // for the taller pixel block, straight into the render target (PXH == 8).
    constexpr bool REG_TOTALS = (PXH == 4) && (NACC < 3);     // rgb: 3 accumulators + counter leave no room for totals
    constexpr int NTOT = REG_TOTALS ? NPX : 1;
    constexpr int FOLD_EVERY = REG_TOTALS ? 64 : 1024;
    float acc[NPX][NACC], tot[NTOT][NACC];
#pragma unroll
    for (int p = 0; p < NPX; ++p) {
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[p][c] = 0.0f;
    }
#pragma unroll
    for (int p = 0; p < NTOT; ++p)
#pragma unroll
        for (int c = 0; c < NACC; ++c) tot[p][c] = 0.0f;
    unsigned long long n_frag = 0;
    int since_fold = 0;
    __syncthreads();

    // Records are dealt to the `split` workgroups of a tile in runs of HDEAL: consecutive records are spatial
    // neighbours (they come from consecutive chunks), so a workgroup's batch of 256 is made of 256 / HDEAL runs
    // taken `split` runs apart -- every workgroup sees an even sample of the tile's footprints
    const long long n_runs = (a.n_records + HDEAL - 1) / HDEAL;
    for (long long run0 = 0; run0 * a.split < n_runs; run0 += 256 / HDEAL) {
        // ---- waves 0-3 test 256 records against the tile and compact the hits into the LDS queue ----
        const long long ri = ((run0 + (tid & 255) / HDEAL) * a.split + sp) * HDEAL + (tid & (HDEAL - 1));
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        bool hit = false;
        if (tid < 256 && ri < a.n_records) {
            g = a.geom[ri];
            const float half = 0.5f * g.z;
            hit = (g.x + half > fx0) && (g.x - half < fx1) && (g.y + half > fy0) && (g.y - half < fy1);
            // the kernel vanishes outside the disc inscribed in the footprint square: a tile wholly
            // beyond radius 0.5221 P (all four stencil texels exactly 0) would only add +0.0
            const float ddx = fmaxf(fmaxf(fx0 - g.x, g.x - fx1), 0.0f), ddy = fmaxf(fmaxf(fy0 - g.y, g.y - fy1), 0.0f);
            hit = hit && !(a.disc_k2 > 0.0f && ddx * ddx + ddy * ddy >= a.disc_k2 * g.z * g.z);
        }
        const unsigned long long mask = __ballot(hit);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0 && wv < 4) s_wcnt[wv] = __popcll(mask);
        __syncthreads();
        int wbase = 0, nq = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wv) wbase += s_wcnt[w];
            nq += s_wcnt[w];
        }
        if (hit) {
            qg[wbase + before] = make_float4(g.x, g.y, 0.5f * g.z, 1.0f / g.z);
            const float w1 = a.w[ri * NW];
            const float w2 = (NW == 2) ? a.w[ri * NW + 1] : 0.0f;
            qw[wbase + before] = make_float4(g.w, (MODE == TSP_MODE_RGB) ? w1 : g.w * w1, w2, 0.0f);
        }
        __syncthreads();
        // ---- every lane evaluates its pixels for each queued footprint ------------------------------
        for (int e = 0; e < nq; ++e) {
            const float4 r4 = qg[e];
            const float pcx = r4.x, pcy = r4.y, half = r4.z, invP = r4.w;
            {   // this wave's strip: skip footprints whose square or disc misses it
                const float sdx = fmaxf(fmaxf(sx0 - pcx, pcx - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - pcy, pcy - sy1), 0.0f);
                if (sdx >= half || sdy >= half || (a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * (4.0f * half * half))) continue;
            }
            const float4 wq = qw[e];
            int col[4], row[PXH];
            float fxs[4], gxs[4], fys[PXH], gys[PXH];
            float cvx[4], cvy[PXH];               // coverage flags: fragment statistics only
            {
                // canonical texel coordinate: u = (d + half) * invP ; tu = u * 64 - 0.5 (tsp_math.h)
                const float d = pxc_own - pcx;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float u = (d + half) * invP;
                // clamping tu to [0, 63] reproduces clamp-to-edge: tu < 0 -> texel 0 weight 1,
                // tu in [63, 63.5) -> texel 63 (its quad holds T[63] twice)
                const float tu = __builtin_amdgcn_fmed3f(__builtin_fmaf(u, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tu);
                const float fr = (tu - f0) * cv;        // uncovered column: both weights 0
                const int c0 = (int)f0;
                const float g0 = cv - fr;
                col[0] = c0; fxs[0] = fr; gxs[0] = g0; cvx[0] = cv;
                col[1] = dpp_row_ror<4>(c0); fxs[1] = dpp_row_ror<4>(fr); gxs[1] = dpp_row_ror<4>(g0);
                col[2] = dpp_row_ror<8>(c0); fxs[2] = dpp_row_ror<8>(fr); gxs[2] = dpp_row_ror<8>(g0);
                col[3] = dpp_row_ror<12>(c0); fxs[3] = dpp_row_ror<12>(fr); gxs[3] = dpp_row_ror<12>(g0);
                if (a.count_frag) {
                    cvx[1] = dpp_row_ror<4>(cv); cvx[2] = dpp_row_ror<8>(cv); cvx[3] = dpp_row_ror<12>(cv);
                }
            }
            {
                const float d = pyc_own - pcy;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float v = (d + half) * invP;
                const float tv = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tv);
                const float fr = (tv - f0) * cv;
                const int r0 = ((int)f0) << 6;
                const float g0 = cv - fr;
                row[0] = dpp_quad_bcast<0>(r0); fys[0] = dpp_quad_bcast<0>(fr); gys[0] = dpp_quad_bcast<0>(g0);
                row[1] = dpp_quad_bcast<1>(r0); fys[1] = dpp_quad_bcast<1>(fr); gys[1] = dpp_quad_bcast<1>(g0);
                row[2] = dpp_quad_bcast<2>(r0); fys[2] = dpp_quad_bcast<2>(fr); gys[2] = dpp_quad_bcast<2>(g0);
                row[3] = dpp_quad_bcast<3>(r0); fys[3] = dpp_quad_bcast<3>(fr); gys[3] = dpp_quad_bcast<3>(g0);
                if (a.count_frag) {
                    cvy[0] = dpp_quad_bcast<0>(cv); cvy[1] = dpp_quad_bcast<1>(cv); cvy[2] = dpp_quad_bcast<2>(cv); cvy[3] = dpp_quad_bcast<3>(cv);
                }
            }
            int ncov_x = 0, ncov_y = 0;
            if (a.count_frag) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { ncov_x += (cvx[t] != 0.0f); ncov_y += (cvy[t] != 0.0f); }
            }
#pragma unroll
            for (int ty = 0; ty < PXH; ++ty) {
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) {
                    const float4 q = Q[row[ty] + col[tx]];
                    // T00*(1-fx) + T01*fx etc. in the cancellation-free form; each FMA differs from the
                    // two-rounding form by <= 1 ulp of a sum of non-negative terms
                    const float top = __builtin_fmaf(q.y, fxs[tx], q.x * gxs[tx]);
                    const float bot = __builtin_fmaf(q.w, fxs[tx], q.z * gxs[tx]);
                    const float kv = __builtin_fmaf(bot, fys[ty], top * gys[ty]);
                    const int p = ty * 4 + tx;
                    acc[p][0] = __builtin_fmaf(kv, wq.x, acc[p][0]);
                    if (NACC >= 2) acc[p][NACC >= 2 ? 1 : 0] = __builtin_fmaf(kv, wq.y, acc[p][NACC >= 2 ? 1 : 0]);
                    if (NACC >= 3) acc[p][NACC - 1] = __builtin_fmaf(kv, wq.z, acc[p][NACC - 1]);
                }
                // keep at most one pixel row of quad loads (4 x 4 VGPRs) in flight: without this the
                // scheduler hoists every ds_read_b128 of the block and spills
                __builtin_amdgcn_sched_barrier(0);
            }
            if (a.count_frag) n_frag += (unsigned long long)(ncov_x * ncov_y);
            // fold the short-run accumulators into the totals every 64 footprints: bounds the
            // float32 accumulation error at ~sqrt(64)*2^-24 per level instead of sqrt(n)
            if (++since_fold == FOLD_EVERY) {
                since_fold = 0;
                if (REG_TOTALS) {
#pragma unroll
                    for (int p = 0; p < NPX; ++p)
#pragma unroll
                        for (int c = 0; c < NACC; ++c) { tot[p < NTOT ? p : 0][c] += acc[p][c]; acc[p][c] = 0.0f; }
                } else {
#pragma unroll
                    for (int ty = 0; ty < PXH; ++ty)
#pragma unroll
                        for (int tx = 0; tx < 4; ++tx) {
                            const int p = ty * 4 + tx;
                            const int gxp = px0 + ((lg - tx) & 3);       // column slot tx
                            if (gxp < R && py0 + ty < R) {
                                double *d = a.img + ((size_t)(py0 + ty) * R + gxp) * C;
#pragma unroll
                                for (int c = 0; c < NACC; ++c) {
                                    if (acc[p][c] != 0.0f) gatomic_add(d + c, acc[p][c]);
                                    acc[p][c] = 0.0f;
                                }
                            }
                        }
                }
            }
        }
        __syncthreads();
    }
    // ---- add this workgroup's partial tile into the render target ---------------------------------
#pragma unroll
    for (int ty = 0; ty < PXH; ++ty) {
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) {
            const int p = ty * 4 + tx, gx = px0 + ((lg - tx) & 3), gy = py0 + ty;   // column slot tx
            if (gx < R && gy < R) {
                double *d = a.img + ((size_t)gy * R + gx) * C;
#pragma unroll
                for (int c = 0; c < NACC; ++c) {
                    const float v = (REG_TOTALS ? tot[p < NTOT ? p : 0][c] : 0.0f) + acc[p][c];
                    if (v != 0.0f) gatomic_add(d + c, v);
                }
            }
        }
    }
    if (a.count_frag) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) atomicAdd(&a.cnt->n_fragments, n_frag);
    }
}

template <int MODE, int NACC, int PXH>
static int launch_huge(tsp_context *ctx, TileArgs ta, size_t smem_h, long long n_huge) {
    const uint32_t attr_bit = 1u << (3 + MODE * 3 + (NACC - 1));
    if (!(ctx->kernel_attr_done & attr_bit)) {
        TSP_HIP(hipFuncSetAttribute((const void *)splat_huge_kernel<MODE, NACC, PXH>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_h));
        ctx->kernel_attr_done |= attr_bit;
    }
    const int htiles_x = (ctx->R + HTILE_W - 1) / HTILE_W, htiles_y = (ctx->R + 16 * PXH - 1) / (16 * PXH);
    const int htiles = htiles_x * htiles_y;
    // enough splits to give every CU many workgroups, but never more than there are record batches
    const long long batches = (n_huge + 255) / 256;
    int split = ctx->huge_split;
    if (split <= 0) split = std::max(1, (ctx->cu_count * 32 + htiles - 1) / htiles);
    split = (int)std::min<long long>(split, std::max<long long>(batches, 1));
    ta.split = split;
    ta.tiles_x = htiles_x;
    hipLaunchKernelGGL((splat_huge_kernel<MODE, NACC, PXH>), dim3(htiles * split), dim3(HT), smem_h, ctx->stream, ta);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// ---------------------------------------------------------------------------------------------
// kernel H2: huge footprints, row-uniform tile gather
// ---------------------------------------------------------------------------------------------
// For P >= 64 px a texel of the 64^2 kernel image is >= 1 pixel wide, so along a pixel ROW the y-interpolation
// factors (fy, gy) and the texel row are the same for every pixel, and along a pixel COLUMN the x-interpolated
// texel rows  L[r](col) = T[r][c]*gx + T[r][c+1]*fx  change only when the texel row r does -- every P/64 pixels.
// H2 maps that structure onto the wave: a lane owns W pixel COLUMNS (64 apart) x HR rows in registers, all 64
// lanes share the same HR pixel rows.  Per footprint a wave
//   * computes the row factors once, one row per lane (canonical texel coordinate, tsp_math.h), and redistributes
//     them through a per-wave LDS table so that lane l holds (fy, gy) of rows 4k + (l & 3), k = 0 .. HR/4 - 1:
//     every QUAD of lanes then carries the four rows of group k and a row's factor reaches all 64 lanes as the
//     DPP operand of the FMA itself (quad_perm:[t,t,t,t]) -- no LDS read, no scalar register per row;
//   * walks its rows with WAVE-UNIFORM control flow (bit tests on ballot masks):
//       on a texel-row change:  top = bot ; bot = L[r + 1](col) from the prefetched pair ; prefetch row r + 2
//       every covered row:      acc += gy*top ; acc += fy*bot                              -- 2 VALU per pixel
// against ~14.5 VALU + one 16-byte LDS read per pixel in kernel H.  The sum has the same non-negative terms as the
// canonical bilinear form in a different association (relative rounding differences of ~1e-7).
#define TSP_DPP_QUAD(t) "quad_perm:[" #t "," #t "," #t "," #t "] row_mask:0xf bank_mask:0xf"

template <int T> __device__ __forceinline__ void fmac_quad(float &acc, float rowval, float v) {
    static_assert(T >= 0 && T < 4, "quad lane");
    if (T == 0) asm volatile("v_fmac_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(0) : "+v"(acc) : "v"(rowval), "v"(v));
    if (T == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(1) : "+v"(acc) : "v"(rowval), "v"(v));
    if (T == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(2) : "+v"(acc) : "v"(rowval), "v"(v));
    if (T == 3) asm volatile("v_fmac_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(3) : "+v"(acc) : "v"(rowval), "v"(v));
}
template <int T> __device__ __forceinline__ float mul_quad(float rowval, float v) {
    float r;
    if (T == 0) asm volatile("v_mul_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(0) : "=v"(r) : "v"(rowval), "v"(v));
    if (T == 1) asm volatile("v_mul_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(1) : "=v"(r) : "v"(rowval), "v"(v));
    if (T == 2) asm volatile("v_mul_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(2) : "=v"(r) : "v"(rowval), "v"(v));
    if (T == 3) asm volatile("v_mul_f32_dpp %0, %1, %2 " TSP_DPP_QUAD(3) : "=v"(r) : "v"(rowval), "v"(v));
    return r;
}
__device__ __forceinline__ void fmac_plain(float &acc, float x, float y) {     // tied operand: the accumulator stays in place
    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y));
}

constexpr int PT_ROWS = 66;          // LDS kernel image rows: 64 + two clamp-to-edge copies of row 63 (for r + 1, r + 2)
constexpr int PT_STRIDE = 65;        // floats per row: 64 + one clamp-to-edge copy of column 63 (for c + 1); odd -> no bank conflicts

constexpr int H2T = 256;             // threads per workgroup of kernel H2: 4 waves = 2 x 2 strips sharing one pair table

template <int MODE, int NACC, int W, int HR, int OCC>
__global__ __launch_bounds__(H2T, OCC) void splat_huge2_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    constexpr int TW = 2 * 64 * W, TH = 2 * HR;            // tile: 2 x 2 wave strips of (64 W) x HR pixels
    constexpr int NG = HR / 4;                             // row groups (one quad of lanes carries a group's factors)
    static_assert(HR == 16 || HR == 32, "rows per wave strip");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // level-0 kernel image with clamp-to-edge padding: texels (r, c) and (r, c + 1) of an x-interpolation are adjacent
    // dwords, fetched by one ds_read2_b32
    float *PT = smem;                                                        // [PT_ROWS][PT_STRIDE]
    float2 *rt_all = reinterpret_cast<float2 *>(smem + ((PT_ROWS * PT_STRIDE + 3) & ~3));   // per wave: (fy, gy) of its HR rows

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R = a.cam.R;
    const int tile_id = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tx0 = (tile_id % a.tiles_x) * TW, ty0 = (tile_id / a.tiles_x) * TH;
    for (int i = tid; i < PT_ROWS * PT_STRIDE; i += H2T) {
        const int j = min(i / PT_STRIDE, 63), x = min(i % PT_STRIDE, 63);
        PT[i] = a.mips[j * 64 + x];
    }
    float2 *rt = rt_all + wv * HR;
    const float2 *rt_quad = rt + (lane & 3);               // this lane's slot in every row group
    const int sx = tx0 + 64 * W * (wv & 1), sy = ty0 + HR * (wv >> 1);
    const float sx0 = (float)sx, sx1 = (float)(sx + 64 * W), sy0 = (float)sy, sy1 = (float)(sy + HR);
    float pxc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) pxc[w] = (sx + 64 * w + lane < R) ? (float)(sx + 64 * w + lane) + 0.5f : __builtin_inff();
    const int myrow = lane & (HR - 1);
    const bool rowlane = lane < HR;
    const float pyc_own = (sy + myrow < R) ? (float)(sy + myrow) + 0.5f : __builtin_inff();
    const float pyc_prev = pyc_own - 1.0f;                 // centre of the row above (exact; +inf stays +inf)

    // float32 accumulators hold at most FOLD_EVERY footprints (rounding error ~ sqrt(n) * 2^-24 relative: < 2e-6 at
    // the worst pixel), then go to the float64 render target; second-level register totals (as kernel H keeps) would
    // cost HR * W more VGPRs and spill here
    constexpr int FOLD_EVERY = 512;
    float acc[HR * W][NACC];
#pragma unroll
    for (int p = 0; p < HR * W; ++p)
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[p][c] = 0.0f;
    unsigned long long n_frag = 0;
    int since_fold = 0;
    const char *PTb = reinterpret_cast<const char *>(PT);
    __syncthreads();                                       // the only workgroup barrier: from here on the waves run free

    // Every wave scans the workgroup's share of the record list on its own, 64 records at a time (one per lane),
    // and keeps those whose square and disc reach ITS strip -- no shared queue, so no wave ever waits for another.
    // The four waves read the same records at about the same time (L1 / L2 hits).  Records are dealt to the `split`
    // workgroups of a tile in runs of HDEAL: consecutive records are spatial neighbours (consecutive chunks), so
    // every workgroup sees an even sample of the tile's footprints.
    const long long n_runs = (a.n_records + HDEAL - 1) / HDEAL;
    auto fetch = [&](long long run0, float4 &g, float &gw1, float &gw2) {
        const long long ri = ((run0 + lane / HDEAL) * a.split + sp) * HDEAL + (lane & (HDEAL - 1));
        g = make_float4(0.f, 0.f, 0.f, 0.f); gw1 = gw2 = 0.0f;
        if (ri < a.n_records) {
            g = a.geom[ri];
            gw1 = a.w[ri * NW];
            if (NW == 2) gw2 = a.w[ri * NW + 1];
        }
    };
    float4 g_next; float gw1_next, gw2_next;
    fetch(0, g_next, gw1_next, gw2_next);
    for (long long run0 = 0; run0 * a.split < n_runs; run0 += 64 / HDEAL) {
        const float4 g = g_next;
        const float gw1 = gw1_next, gw2 = gw2_next;
        fetch(run0 + 64 / HDEAL, g_next, gw1_next, gw2_next);      // the next 64 records load while these are rasterised
        const float g_half = 0.5f * g.z;
        bool hit;
        {
            const float sdx = fmaxf(fmaxf(sx0 - g.x, g.x - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - g.y, g.y - sy1), 0.0f);
            // g.z = 0 marks an empty slot; the kernel vanishes outside the disc inscribed in the footprint square
            hit = g.z > 0.0f && g.z < a.p_hi && sdx < g_half && sdy < g_half && !(a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * g.z * g.z);
        }
        unsigned long long hits = __ballot(hit);
        if (hits == 0ull) continue;
        const float g_invP = 1.0f / g.z;
        const float g_w1 = (MODE == TSP_MODE_RGB) ? gw1 : g.w * gw1;
        while (hits) {
            const int src = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            // the footprint's parameters, wave-uniform (scalar registers)
            const float pcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), src));
            const float pcy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.y), src));
            const float half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_half), src));
            const float invP = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_invP), src));
            float4 wq;
            wq.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.w), src));
            wq.y = (NACC >= 2) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_w1), src)) : 0.0f;
            wq.z = (NACC >= 3) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gw2), src)) : 0.0f;
            wq.w = 0.0f;
            if (NACC >= 2) {
                // the channel weights feed tied-operand FMAs on every row: park them in VGPRs once per footprint (left to
                // itself the compiler re-copies the scalar before every use: three extra v_mov per row)
                asm volatile("v_mov_b32 %0, %1" : "=v"(wq.x) : "s"(wq.x));
                asm volatile("v_mov_b32 %0, %1" : "=v"(wq.y) : "s"(wq.y));
                if (NACC >= 3) asm volatile("v_mov_b32 %0, %1" : "=v"(wq.z) : "s"(wq.z));
            }
            // ---- rows: lane j < HR evaluates row j and the texel row of the row above it -----------------
            unsigned covmask, chgmask, jmpmask;
            int r512;                                   // byte offset of this lane's texel row in PT
            {
                const float d = pyc_own - pcy;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float v = (d + half) * invP;
                const float tv = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tv);
                const float fr = (tv - f0) * cv;
                const int r = (int)f0;
                const float vp = ((pyc_prev - pcy) + half) * invP;
                const int rprev = (int)__builtin_floorf(__builtin_amdgcn_fmed3f(__builtin_fmaf(vp, 64.0f, -0.5f), 0.0f, 63.0f));
                r512 = r * (PT_STRIDE * 4);
                asm volatile("" ::: "memory");          // (in-order LDS: the previous footprint's table reads are done)
                if (rowlane) rt[myrow] = make_float2(fr, cv - fr);
                asm volatile("" ::: "memory");
                const bool covered = rowlane && cv != 0.0f;
                covmask = (unsigned)__ballot(covered);
                chgmask = (unsigned)__ballot(covered && myrow > 0 && r != rprev);
                // texel rows advance by at most one per pixel row when P >= 64; rounding at P ~ 64 may still skip one
                jmpmask = (unsigned)__ballot(covered && myrow > 0 && r != rprev && r != rprev + 1);
            }
            if (covmask == 0u) continue;
            {   // the first covered row loads both texel rows
                const unsigned first = covmask & (0u - covmask);
                chgmask |= first; jmpmask |= first;
            }
            // row factors of group k for the DPP broadcast: lane l takes rows 4k + (l & 3)
            float2 rowf[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) rowf[k] = rt_quad[4 * k];
            // ---- columns: W per lane ----
            int caddr[W];
            float fxs[W], gxs[W];
            int ncov_x = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const float d = pxc[w] - pcx;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float u = (d + half) * invP;
                const float tu = __builtin_amdgcn_fmed3f(__builtin_fmaf(u, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tu);
                const float fr = (tu - f0) * cv;        // uncovered column: both weights 0
                caddr[w] = ((int)f0) * 4;
                // density: the particle weight rides on the column factors, so a pixel costs two FMAs
                fxs[w] = (NACC == 1) ? fr * wq.x : fr;
                gxs[w] = (NACC == 1) ? (cv - fr) * wq.x : (cv - fr);
                if (a.count_frag) ncov_x += (cv != 0.0f);
            }
            float top[W], bot[W];
            float2 nxt[W];                              // prefetched pair of texel row r + 2
            int r_off = 0;                              // byte offset of the current texel row (wave-uniform)
#pragma unroll
            for (int w = 0; w < W; ++w) { top[w] = bot[w] = 0.0f; nxt[w] = make_float2(0.f, 0.f); }
            auto pair_at = [&](int w, int byteoff) -> float2 {
                const float *t = reinterpret_cast<const float *>(PTb + byteoff + caddr[w]);
                return make_float2(t[0], t[1]);
            };
            auto lerp = [&](int w, float2 t) -> float { return __builtin_fmaf(t.y, fxs[w], t.x * gxs[w]); };
            auto row_change = [&](int ty, bool jump) {      // `jump` is wave-uniform
                if (jump) {
                    r_off = __builtin_amdgcn_readlane(r512, ty);
#pragma unroll
                    for (int w = 0; w < W; ++w) { top[w] = lerp(w, pair_at(w, r_off)); bot[w] = lerp(w, pair_at(w, r_off + PT_STRIDE * 4)); }
                } else {
                    r_off += PT_STRIDE * 4;
#pragma unroll
                    for (int w = 0; w < W; ++w) { top[w] = bot[w]; bot[w] = lerp(w, nxt[w]); }
                }
#pragma unroll
                for (int w = 0; w < W; ++w) nxt[w] = pair_at(w, r_off + 2 * PT_STRIDE * 4);
            };
#define TSP_H2_ROW(K, T)                                                                                       \
            {                                                                                                  \
                constexpr int ty_ = 4 * (K) + (T);                                                             \
                if ((chgmask >> ty_) & 1u) row_change(ty_, ((jmpmask >> ty_) & 1u) != 0u);                       \
                _Pragma("unroll") for (int w = 0; w < W; ++w) {                                                \
                    float *ac = acc[ty_ * W + w];                                                              \
                    if (NACC == 1) {                                                                           \
                        fmac_quad<T>(ac[0], rowf[K].y, top[w]);                                                \
                        fmac_quad<T>(ac[0], rowf[K].x, bot[w]);                                                \
                    } else {                                                                                   \
                        float kv = mul_quad<T>(rowf[K].y, top[w]);                                             \
                        fmac_quad<T>(kv, rowf[K].x, bot[w]);                                                   \
                        fmac_plain(ac[0], kv, wq.x);                                                           \
                        fmac_plain(ac[NACC >= 2 ? 1 : 0], kv, wq.y);                                           \
                        if (NACC >= 3) fmac_plain(ac[NACC - 1], kv, wq.z);                                     \
                    }                                                                                          \
                }                                                                                              \
            }
            // Only the rolling texel rows (top, bot, nxt) are touched under a (wave-uniform) branch; the accumulation itself
            // is straight-line (an uncovered row has fy = gy = 0); groups of four rows wholly outside the footprint are
            // skipped.  (Laying the change out of line as the unlikely path measured slower: this kernel serves the
            // footprints below p_mega, whose texel rows change every 1-8 pixel rows.)
#define TSP_H2_GROUP(K)                                                                                        \
            if ((K) < NG && ((covmask >> (4 * (K))) & 15u) != 0u) { TSP_H2_ROW(K, 0) TSP_H2_ROW(K, 1) TSP_H2_ROW(K, 2) TSP_H2_ROW(K, 3) }
            TSP_H2_GROUP(0) TSP_H2_GROUP(1) TSP_H2_GROUP(2) TSP_H2_GROUP(3)
            TSP_H2_GROUP(4) TSP_H2_GROUP(5) TSP_H2_GROUP(6) TSP_H2_GROUP(7)
#undef TSP_H2_GROUP
#undef TSP_H2_ROW
            if (a.count_frag) n_frag += (unsigned long long)(ncov_x * __popc(covmask));
            if (++since_fold == FOLD_EVERY) {
                since_fold = 0;
                double *img = a.img + ((size_t)sy * R + (sx + lane)) * C;
                asm volatile("" : "+v"(img));        // addresses are formed here, not hoisted to the kernel entry and spilled
#pragma unroll
                for (int ty = 0; ty < HR; ++ty)
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const int p = ty * W + w, gx = sx + 64 * w + lane, gy = sy + ty;
                        if (gx < R && gy < R) {
                            double *d = img + ((size_t)ty * R + 64 * w) * C;
#pragma unroll
                            for (int c = 0; c < NACC; ++c) {
                                if (acc[p][c] != 0.0f) gatomic_add(d + c, acc[p][c]);
                                acc[p][c] = 0.0f;
                            }
                        }
                    }
            }
        }
    }
    // ---- add this wave's partial strip into the render target ---------------------------------------
    double *img_end = a.img + ((size_t)sy * R + (sx + lane)) * C;
    asm volatile("" : "+v"(img_end));
#pragma unroll
    for (int ty = 0; ty < HR; ++ty) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int p = ty * W + w, gx = sx + 64 * w + lane, gy = sy + ty;
            if (gx < R && gy < R) {
                double *d = img_end + ((size_t)ty * R + 64 * w) * C;
#pragma unroll
                for (int c = 0; c < NACC; ++c) {
                    const float v = acc[p][c];
                    if (v != 0.0f) gatomic_add(d + c, v);
                }
            }
        }
    }
    if (a.count_frag) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) atomicAdd(&a.cnt->n_fragments, n_frag);
    }
}

template <int MODE, int NACC, int W, int HR, int OCC>
static int launch_huge2(tsp_context *ctx, TileArgs ta, long long n_huge) {
    const size_t smem = (size_t)((PT_ROWS * PT_STRIDE + 3) & ~3) * sizeof(float) + (H2T / 64) * HR * sizeof(float2);
    const int tw = 2 * 64 * W, th = 2 * HR;
    const int htiles_x = (ctx->R + tw - 1) / tw, htiles_y = (ctx->R + th - 1) / th;
    const int htiles = htiles_x * htiles_y;
    const long long batches = (n_huge + 63) / 64;
    int split = ctx->huge_split;
    // many short workgroups: a wave lives ~1 ms at split 64 and the tail of the launch (tiles differ 10x in work)
    // cost 2.5 ms of 21; measured 64 -> 128: 21.9 -> 19.5 ms, 256: 19.2 ms, 512: 22.5 ms
    if (split <= 0) split = std::max(1, (ctx->cu_count * 128 + htiles - 1) / htiles);
    split = (int)std::min<long long>(split, std::max<long long>(batches, 1));
    ta.split = split;
    ta.tiles_x = htiles_x;
    hipLaunchKernelGGL((splat_huge2_kernel<MODE, NACC, W, HR, OCC>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// ---------------------------------------------------------------------------------------------
// kernel H3: mega footprints on the matrix cores
// ---------------------------------------------------------------------------------------------
// Within a strip of pixels the contribution of one footprint is a sum of outer products,
//     img[row][col] += sum_k U[row][k] * V[k][col],   k = the texel rows the strip's pixel rows touch,
// with V[k][col] = w * (T[r0+k][c]*gx + T[r0+k][c+1]*fx) (x-interpolated texel row) and U[row][k] = gy(row) if row's
// texel row is r0 + k, fy(row) if it is r0 + k - 1, else 0 -- exactly the shape of v_mfma_f32_32x32x2_f32 (A: 32 rows x
// 2 k, B: 2 k x 32 columns, one VGPR each, exact f32 FMA chain).  When a texel is >= 8 pixels tall a 32-row strip
// touches <= 6 texel rows, i.e. <= 3 MFMA k-steps, and the row factors need no broadcast at all: the lane that
// evaluates row i IS the lane that supplies A[i][k].  The matrix pipe then does the per-pixel work (2 MFMAs per
// 64x32 strip and k-step) while the VALU only prepares ~80 instructions per footprint and strip.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NB = 32-column blocks per wave strip (strip = 32 NB columns x 32 rows): the row factors and the A operand of a k-step
// are shared by the NB column blocks, so wider strips spend fewer VALU instructions per pixel (the kernel is VALU-bound:
// ~60 preparation instructions per footprint and strip against 2 NB MFMAs per k-step)
template <int MODE, int NACC, int NB, int OCC>
__global__ __launch_bounds__(H2T, OCC) void splat_mega_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    constexpr int SW = 32 * NB, TW = 2 * SW, TH = 64;       // tile: 2 x 2 wave strips of SW x 32 pixels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *PT = smem;                                      // [PT_ROWS][PT_STRIDE] level-0 kernel image, clamp-to-edge padded
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;              // MFMA operand roles: row / column index, k half
    const int R = a.cam.R;
    const int tile_id = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tx0 = (tile_id % a.tiles_x) * TW, ty0 = (tile_id / a.tiles_x) * TH;
    for (int i = tid; i < PT_ROWS * PT_STRIDE; i += H2T) {
        const int j = min(i / PT_STRIDE, 63), x = min(i % PT_STRIDE, 63);
        PT[i] = a.mips[j * 64 + x];
    }
    const int sx = tx0 + SW * (wv & 1), sy = ty0 + 32 * (wv >> 1);
    const float sx0 = (float)sx, sx1 = (float)(sx + SW), sy0 = (float)sy, sy1 = (float)(sy + 32);
    const float pyc = (sy + li < R) ? (float)(sy + li) + 0.5f : __builtin_inff();
    const int last_row = min(31, R - 1 - sy);              // last pixel row of the strip inside the image (wave-uniform)
    float pxc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) pxc[b] = (sx + 32 * b + li < R) ? (float)(sx + 32 * b + li) + 0.5f : __builtin_inff();
    constexpr int FOLD_EVERY = 512;                        // as kernel H2: float32 accumulators hold <= 512 footprints
    f32x16 acc[NACC][NB];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[c][b][v] = 0.0f;
    unsigned long long n_frag = 0;
    int since_fold = 0;
    const char *PTb = reinterpret_cast<const char *>(PT);
    __syncthreads();                                       // the only workgroup barrier

    auto flush = [&]() {
        double *img = a.img + ((size_t)(sy + 4 * kh) * R + (sx + li)) * C;
        asm volatile("" : "+v"(img));
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = (v >> 2) * 8 + (v & 3);    // + 4 * kh (in img)
                if (sx + 32 * b + li < R && sy + 4 * kh + row < R) {
                    double *d = img + ((size_t)row * R + 32 * b) * C;
#pragma unroll
                    for (int c = 0; c < NACC; ++c) {
                        if (acc[c][b][v] != 0.0f) gatomic_add(d + c, acc[c][b][v]);
                        acc[c][b][v] = 0.0f;
                    }
                }
            }
    };

    const long long n_runs = (a.n_records + HDEAL - 1) / HDEAL;
    auto fetch = [&](long long run0, float4 &g, float &gw1, float &gw2) {
        const long long ri = ((run0 + lane / HDEAL) * a.split + sp) * HDEAL + (lane & (HDEAL - 1));
        g = make_float4(0.f, 0.f, 0.f, 0.f); gw1 = gw2 = 0.0f;
        if (ri < a.n_records) {
            g = a.geom[ri];
            gw1 = a.w[ri * NW];
            if (NW == 2) gw2 = a.w[ri * NW + 1];
        }
    };
    float4 g_next; float gw1_next, gw2_next;
    fetch(0, g_next, gw1_next, gw2_next);
    for (long long run0 = 0; run0 * a.split < n_runs; run0 += 64 / HDEAL) {
        const float4 g = g_next;
        const float gw1 = gw1_next, gw2 = gw2_next;
        fetch(run0 + 64 / HDEAL, g_next, gw1_next, gw2_next);
        const float g_half = 0.5f * g.z;
        bool hit;
        {
            const float sdx = fmaxf(fmaxf(sx0 - g.x, g.x - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - g.y, g.y - sy1), 0.0f);
            hit = g.z >= a.p_lo && sdx < g_half && sdy < g_half && !(a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * g.z * g.z);
        }
        unsigned long long hits = __ballot(hit);
        if (hits == 0ull) continue;
        const float g_invP = 1.0f / g.z;
        const float g_w1 = (MODE == TSP_MODE_RGB) ? gw1 : g.w * gw1;
        while (hits) {
            const int src = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const float pcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), src));
            const float pcy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.y), src));
            const float half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_half), src));
            const float invP = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_invP), src));
            const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.w), src));
            const float w1 = (NACC >= 2) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_w1), src)) : 0.0f;
            const float w2 = (NACC >= 3) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gw2), src)) : 0.0f;
            // ---- rows (A operand): lane (li, kh) evaluates pixel row li; canonical texel coordinate (tsp_math.h) ----
            float fy, gy;
            int rel, r0, nsteps;
            {
                const float d = pyc - pcy;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float v = (d + half) * invP;
                const float tv = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tv);
                fy = (tv - f0) * cv;
                gy = cv - fy;
                const int r = (int)f0;
                r0 = __builtin_amdgcn_readlane(r, 0);                 // texel row of the strip's first pixel row
                rel = r - r0;
                const int kmax = __builtin_amdgcn_readlane(rel, last_row);   // texel rows are monotone down the strip
                nsteps = (kmax + 3) >> 1;                             // texel rows r0 .. r0 + kmax + 1, two per MFMA
                if (a.count_frag) {
                    const unsigned long long rows = __ballot(cv != 0.0f && kh == 0);
                    int ncx = 0;
#pragma unroll
                    for (int b = 0; b < NB; ++b) ncx += (__builtin_fabsf(pxc[b] - pcx) < half) ? 1 : 0;
                    if (kh == 0) n_frag += (unsigned long long)(ncx * __popcll(rows));
                }
            }
            // ---- columns (B operand): lane (li, kh) evaluates pixel columns 32 b + li ----
            int caddr[NB];
            float fxs[NB], gxs[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float d = pxc[b] - pcx;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float u = (d + half) * invP;
                const float tu = __builtin_amdgcn_fmed3f(__builtin_fmaf(u, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tu);
                const float fr = (tu - f0) * cv;
                caddr[b] = ((int)f0) * 4;
                fxs[b] = (NACC == 1) ? fr * w0 : fr;    // density: the particle weight rides on the column factors
                gxs[b] = (NACC == 1) ? (cv - fr) * w0 : (cv - fr);
            }
            int rowoff = (r0 + kh) * (PT_STRIDE * 4);   // this lane's texel row of the current k-step (bytes)
            int kk = kh;
            for (int m = 0; m < nsteps; ++m) {
                const float A = (rel == kk) ? gy : ((rel + 1 == kk) ? fy : 0.0f);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float *t = reinterpret_cast<const float *>(PTb + rowoff + caddr[b]);
                    const float L = __builtin_fmaf(t[1], fxs[b], t[0] * gxs[b]);
                    acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, (NACC == 1) ? L : L * w0, acc[0][b], 0, 0, 0);
                    if (NACC >= 2) acc[NACC >= 2 ? 1 : 0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, L * w1, acc[NACC >= 2 ? 1 : 0][b], 0, 0, 0);
                    if (NACC >= 3) acc[NACC - 1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, L * w2, acc[NACC - 1][b], 0, 0, 0);
                }
                kk += 2;
                rowoff += 2 * PT_STRIDE * 4;
            }
            if (++since_fold == FOLD_EVERY) { since_fold = 0; flush(); }
        }
    }
    flush();
    if (a.count_frag) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) atomicAdd(&a.cnt->n_fragments, n_frag);
    }
}

template <int MODE, int NACC, int NB, int OCC>
static int launch_mega(tsp_context *ctx, TileArgs ta, long long n_huge) {
    const size_t smem = (size_t)((PT_ROWS * PT_STRIDE + 3) & ~3) * sizeof(float);
    const int htiles_x = (ctx->R + 64 * NB - 1) / (64 * NB), htiles_y = (ctx->R + 63) / 64;
    const int htiles = htiles_x * htiles_y;
    const long long batches = (n_huge + 63) / 64;
    int split = ctx->mega_split;
    if (split <= 0) split = std::max(1, (ctx->cu_count * 64 + htiles - 1) / htiles);
    split = (int)std::min<long long>(split, std::max<long long>(batches, 1));
    ta.split = split;
    ta.tiles_x = htiles_x;
    hipLaunchKernelGGL((splat_mega_kernel<MODE, NACC, NB, OCC>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// ---------------------------------------------------------------------------------------------
// rgb fragment-counter channel of the deferred (MID and HUGE) footprints
// ---------------------------------------------------------------------------------------------
// fragment_rgb writes (k r, k g, k b, 1): channel 3 counts the footprint SQUARES covering a pixel, also where
// the kernel value is exactly 0.  Per footprint that is the indicator of a pixel rectangle, so instead of one add
// per fragment kernels M and H leave the channel alone (H may then skip whatever lies outside the kernel's disc,
// M saves a quarter of its LDS atomics) and the rectangles are summed exactly in integers: +-1 at the four corners of each rectangle, then a 2-D prefix
// sum, added to the float64 render target.
__global__ __launch_bounds__(256) void rect_count_corners_kernel(const float4 *__restrict__ geom, long long n, int R, int *__restrict__ D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 g = geom[i];
    const float half = 0.5f * g.z;
    int ilo, ihi, jlo, jhi;
    cover_range(g.x, half, R, ilo, ihi);
    cover_range(g.y, half, R, jlo, jhi);
    if (ilo > ihi || jlo > jhi) return;
    const int S = R + 1;
    atomicAdd(&D[jlo * S + ilo], 1);
    atomicAdd(&D[jlo * S + ihi + 1], -1);
    atomicAdd(&D[(jhi + 1) * S + ilo], -1);
    atomicAdd(&D[(jhi + 1) * S + ihi + 1], 1);
}

// inclusive prefix sum along each of the first R rows (one 256-thread workgroup per row)
__global__ __launch_bounds__(256) void count_row_scan_kernel(int *__restrict__ D, int R) {
    __shared__ int part[256];
    const int S = R + 1, tid = threadIdx.x;
    int *row = D + (size_t)blockIdx.x * S;
    const int per = (S + 255) / 256, b = tid * per, e = min(b + per, S);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += row[i];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int i = b; i < e; ++i) { run += row[i]; row[i] = run; }
}

constexpr int CBAND = 64;
// column sums of each band of 64 rows
__global__ __launch_bounds__(256) void count_band_sum_kernel(const int *__restrict__ D, int R, int *__restrict__ band) {
    const int i = blockIdx.x * 256 + threadIdx.x, bnd = blockIdx.y;
    if (i >= R) return;
    const int S = R + 1, j0 = bnd * CBAND, j1 = min(j0 + CBAND, R);
    int sum = 0;
    for (int j = j0; j < j1; ++j) sum += D[(size_t)j * S + i];
    band[(size_t)bnd * R + i] = sum;
}
// finish the prefix sum down the columns and add the counts into channel 3
__global__ __launch_bounds__(256) void count_apply_kernel(const int *__restrict__ D, const int *__restrict__ band, int R, double *__restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x, bnd = blockIdx.y;
    if (i >= R) return;
    const int S = R + 1, j0 = bnd * CBAND, j1 = min(j0 + CBAND, R);
    long long run = 0;
    for (int b = 0; b < bnd; ++b) run += band[(size_t)b * R + i];
    for (int j = j0; j < j1; ++j) {
        run += D[(size_t)j * S + i];
        if (run != 0) img[((size_t)j * R + i) * 4 + 3] += (double)run;
    }
}

static int add_rect_counts(tsp_context *ctx, const float4 *mid_geom, long long n_mid, const float4 *huge_geom, long long n_huge,
                           const float4 *mega_geom, long long n_mega) {
    Workspace &ws = ctx->ws;
    const int R = ctx->R, S = R + 1, nb = (R + CBAND - 1) / CBAND;
    if (!ws.count_diff) {
        TSP_HIP(hipMalloc((void **)&ws.count_diff, (size_t)S * S * sizeof(int)));
        TSP_HIP(hipMalloc((void **)&ws.count_band, (size_t)nb * R * sizeof(int)));
    }
    hipStream_t st = ctx->stream;
    TSP_HIP(hipMemsetAsync(ws.count_diff, 0, (size_t)S * S * sizeof(int), st));
    if (n_mid > 0) hipLaunchKernelGGL(rect_count_corners_kernel, dim3((unsigned)((n_mid + 255) / 256)), dim3(256), 0, st, mid_geom, n_mid, R, ws.count_diff);
    if (n_huge > 0) hipLaunchKernelGGL(rect_count_corners_kernel, dim3((unsigned)((n_huge + 255) / 256)), dim3(256), 0, st, huge_geom, n_huge, R, ws.count_diff);
    if (n_mega > 0) hipLaunchKernelGGL(rect_count_corners_kernel, dim3((unsigned)((n_mega + 255) / 256)), dim3(256), 0, st, mega_geom, n_mega, R, ws.count_diff);
    hipLaunchKernelGGL(count_row_scan_kernel, dim3(R), dim3(256), 0, st, ws.count_diff, R);
    hipLaunchKernelGGL(count_band_sum_kernel, dim3((R + 255) / 256, nb), dim3(256), 0, st, ws.count_diff, R, ws.count_band);
    hipLaunchKernelGGL(count_apply_kernel, dim3((R + 255) / 256, nb), dim3(256), 0, st, ws.count_diff, ws.count_band, R, ctx->image64);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int grow(void **p, int64_t *cap, int64_t need, size_t elem) {
    if (*cap >= need) return TSP_OK;
    if (*p) TSP_HIP(hipFree(*p));
    *p = nullptr;
    TSP_HIP(hipMalloc(p, (size_t)need * elem));
    *cap = need;
    return TSP_OK;
}

template <int MODE>
static int run_pipeline(tsp_context *ctx, const Camera &cam, const int64_t *h_starts, const int64_t *h_lens, int n_ranges,
                        int64_t total) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    Workspace &ws = ctx->ws;
    hipStream_t st = ctx->stream;

    // ranges -> chunks
    std::vector<int64_t> pack(3 * (size_t)n_ranges + 1);
    int64_t n_chunks64 = 0;
    for (int i = 0; i < n_ranges; ++i) {
        pack[i] = h_starts[i];
        pack[n_ranges + i] = h_lens[i];
        pack[2 * n_ranges + i] = n_chunks64;
        n_chunks64 += (h_lens[i] + CHUNK - 1) / CHUNK;
    }
    pack[3 * n_ranges] = n_chunks64;
    TSP_REQUIRE(n_chunks64 < (1ll << 30), TSP_EINVAL, "too many chunks");
    const int n_chunks = (int)n_chunks64;
    if (ws.range_capacity < (int64_t)pack.size()) {
        if (ws.range_prefix) TSP_HIP(hipFree(ws.range_prefix));
        ws.range_capacity = (int64_t)pack.size() * 2 + 64;
        TSP_HIP(hipMalloc((void **)&ws.range_prefix, ws.range_capacity * sizeof(int64_t)));
    }
    TSP_HIP(hipMemcpyAsync(ws.range_prefix, pack.data(), pack.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    // per-chunk segment table
    if (ws.seg_capacity < n_chunks) {
        if (ws.seg_count) TSP_HIP(hipFree(ws.seg_count));
        if (ws.seg_offset) TSP_HIP(hipFree(ws.seg_offset));
        if (ws.seg_bbox) TSP_HIP(hipFree(ws.seg_bbox));
        ws.seg_capacity = (int64_t)n_chunks + n_chunks / 4 + 64;
        TSP_HIP(hipMalloc((void **)&ws.seg_count, ws.seg_capacity * sizeof(int)));
        TSP_HIP(hipMalloc((void **)&ws.seg_offset, ws.seg_capacity * sizeof(long long)));
        TSP_HIP(hipMalloc((void **)&ws.seg_bbox, ws.seg_capacity * sizeof(float4)));
    }
    // record lists: start modest, grow to the exact need when a frame overflows (rare)
    int rc;
    if (ws.mid_capacity == 0) {
        const int64_t guess = std::max<int64_t>(total / 4, 1 << 16);
        if ((rc = grow(&ws.mid_geom, &ws.mid_capacity, guess, sizeof(float4)))) return rc;
        if (ws.mid_w) TSP_HIP(hipFree(ws.mid_w));
        TSP_HIP(hipMalloc(&ws.mid_w, (size_t)ws.mid_capacity * 2 * sizeof(float)));
    }
    if (ws.huge_capacity == 0) {
        const int64_t guess = std::max<int64_t>(total / 16, 1 << 16);
        if ((rc = grow(&ws.huge_geom, &ws.huge_capacity, guess, sizeof(float4)))) return rc;
        if (ws.huge_w) TSP_HIP(hipFree(ws.huge_w));
        TSP_HIP(hipMalloc(&ws.huge_w, (size_t)ws.huge_capacity * 2 * sizeof(float)));
    }

    Particles parts = ctx->p;
    if (!ctx->use_quantity) parts.q = nullptr;
    const int tiles_x = (ctx->R + TILE - 1) / TILE;
    constexpr int WIN = WinSize<MODE>::value;
    const bool second_channel = (MODE == TSP_MODE_DEPTH) || (MODE == TSP_MODE_RGB) || (ctx->p.q != nullptr && ctx->use_quantity);
    const int WCr = (MODE == TSP_MODE_RGB) ? 4 : (second_channel ? 2 : 1);
    const size_t smem_s = (size_t)WCr * WIN * WIN * sizeof(double) + 64 * sizeof(float);
    constexpr int WCM = (MODE == TSP_MODE_RGB) ? 3 : C;      // LDS tile channels of kernel M (rgb: values only)
    const size_t smem_m = (size_t)(WCr == 1 ? 1 : WCM) * MTILE_H * MSTR * sizeof(double) + MIP_TOTAL * sizeof(float);
    const int mtiles_y = (ctx->R + MTILE_H - 1) / MTILE_H;
    const size_t smem_h = (size_t)(64 * 64 + 512) * sizeof(float4);
    if (!(ctx->kernel_attr_done & (1u << MODE))) {
        TSP_HIP(hipFuncSetAttribute((const void *)splat_stream_kernel<MODE, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)C * WIN * WIN * sizeof(double) + 256)));
        TSP_HIP(hipFuncSetAttribute((const void *)splat_stream_kernel<MODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)C * WIN * WIN * sizeof(double) + 256)));
        TSP_HIP(hipFuncSetAttribute((const void *)splat_mid_kernel<MODE, WCM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)C * MTILE_H * MSTR * sizeof(double) + MIP_TOTAL * sizeof(float))));
        TSP_HIP(hipFuncSetAttribute((const void *)splat_mid_kernel<MODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)C * MTILE_H * MSTR * sizeof(double) + MIP_TOTAL * sizeof(float))));
        ctx->kernel_attr_done |= 1u << MODE;
    }

    Counters hc, carry;
    memset(&carry, 0, sizeof(carry));
    for (int attempt = 0; attempt < 2; ++attempt) {
        StreamArgs sa;
        sa.p = parts;
        sa.ranges = ws.range_prefix; sa.n_ranges = n_ranges; sa.n_chunks = n_chunks;
        const int max_blocks = ctx->cu_count * ctx->stream_blocks_per_cu;
        // at least 8 consecutive chunks per workgroup: amortises the window set-up and keeps the window following the chunks
        sa.chunks_per_block = std::max(8, (n_chunks + max_blocks - 1) / max_blocks);
        const int grid_s = (n_chunks + sa.chunks_per_block - 1) / sa.chunks_per_block;
        sa.cam = cam; sa.mips = ctx->mips; sa.img = ctx->image64;
        sa.mid_geom = (float4 *)ws.mid_geom; sa.mid_w = (float *)ws.mid_w; sa.mid_capacity = ws.mid_capacity;
        sa.huge_geom = (float4 *)ws.huge_geom; sa.huge_w = (float *)ws.huge_w; sa.huge_capacity = ws.huge_capacity;
        sa.seg_count = ws.seg_count; sa.seg_offset = ws.seg_offset; sa.seg_bbox = ws.seg_bbox;
        sa.cnt = ctx->counters; sa.p_small = ctx->p_small; sa.p_mega = (ctx->huge_variant != 0 && MODE != TSP_MODE_RGB && ctx->p_mega > 0.0f) ? ctx->p_mega : __builtin_inff(); sa.count_frag = ctx->count_fragments ? 1 : 0;
        sa.emit_small = attempt == 0 ? 1 : 0;
        TSP_HIP(hipEventRecord(ctx->ev[2], st));
        if (WCr == 1) hipLaunchKernelGGL((splat_stream_kernel<MODE, 1>), dim3(grid_s), dim3(256), smem_s, st, sa);
        else hipLaunchKernelGGL((splat_stream_kernel<MODE, C>), dim3(grid_s), dim3(256), smem_s, st, sa);
        TSP_HIP(hipGetLastError());
        TSP_HIP(hipEventRecord(ctx->ev[3], st));
        // the record counts size the two tile launches (and reveal a list overflow)
        TSP_HIP(hipMemcpyAsync(&hc, ctx->counters, sizeof(hc), hipMemcpyDeviceToHost, st));
        TSP_HIP(hipStreamSynchronize(st));
        const bool mid_over = (int64_t)hc.n_mid > ws.mid_capacity, huge_over = (int64_t)(hc.n_huge + hc.n_mega) > ws.huge_capacity;
        if (!mid_over && !huge_over) break;
        TSP_REQUIRE(attempt == 0, TSP_ENOMEM, "record lists overflowed twice");
        // enlarge and replay kernel S in records-only mode (its small footprints are already in the image)
        if (mid_over) {
            if ((rc = grow(&ws.mid_geom, &ws.mid_capacity, (int64_t)hc.n_mid + (int64_t)hc.n_mid / 8 + 1024, sizeof(float4)))) return rc;
            if (ws.mid_w) TSP_HIP(hipFree(ws.mid_w));
            TSP_HIP(hipMalloc(&ws.mid_w, (size_t)ws.mid_capacity * 2 * sizeof(float)));
        }
        if (huge_over) {
            if ((rc = grow(&ws.huge_geom, &ws.huge_capacity, (int64_t)(hc.n_huge + hc.n_mega) + (int64_t)(hc.n_huge + hc.n_mega) / 8 + 1024, sizeof(float4)))) return rc;
            if (ws.huge_w) TSP_HIP(hipFree(ws.huge_w));
            TSP_HIP(hipMalloc(&ws.huge_w, (size_t)ws.huge_capacity * 2 * sizeof(float)));
        }
        TSP_HIP(hipMemsetAsync(ctx->counters, 0, sizeof(Counters), st));
        carry = hc;
    }

    if (carry.n_small || carry.n_fragments) {   // statistics of the first attempt (its small footprints stand)
        hc.n_small += carry.n_small;
        hc.n_fragments += carry.n_fragments;
        TSP_HIP(hipMemcpyAsync(ctx->counters, &hc, sizeof(hc), hipMemcpyHostToDevice, st));
        TSP_HIP(hipStreamSynchronize(st));
    }
    TileArgs ta;
    ta.seg_count = ws.seg_count; ta.seg_offset = ws.seg_offset; ta.seg_bbox = ws.seg_bbox; ta.n_chunks = n_chunks;
    ta.cam = cam; ta.mips = ctx->mips; ta.img = ctx->image64; ta.cnt = ctx->counters; ta.tiles_x = tiles_x;
    ta.count_frag = ctx->count_fragments ? 1 : 0;
    // corner culling is exact for the value channels; the rgb counter channel (which also counts zero-valued
    // fragments) is not touched by kernel H at all: add_rect_counts() sums the footprint rectangles instead
    ta.disc_k2 = (ctx->lut_zero_outside_disc && !ctx->count_fragments) ? 0.5235f * 0.5235f : 0.0f;
    // Kernel M (LDS-atomic-bound) and kernel H (VALU-bound) only depend on kernel S and add into the
    // float64 image with atomics, so they run concurrently on two streams and share the CUs.
    hipStream_t st_mid = ctx->overlap_mid_huge ? ctx->stream2 : st;
    if (ctx->overlap_mid_huge) {
        TSP_HIP(hipEventRecord(ctx->ev[8], st));
        TSP_HIP(hipStreamWaitEvent(st_mid, ctx->ev[8], 0));
    }
    TSP_HIP(hipEventRecord(ctx->ev[4], st_mid));
    if (hc.n_mid > 0) {
        ta.geom = (const float4 *)ws.mid_geom; ta.w = (const float *)ws.mid_w; ta.n_records = (long long)hc.n_mid;
        ta.split = ctx->mid_split;
        if (WCr == 1) hipLaunchKernelGGL((splat_mid_kernel<MODE, 1>), dim3(tiles_x * mtiles_y * ta.split), dim3(MT), smem_m, st_mid, ta);
        else hipLaunchKernelGGL((splat_mid_kernel<MODE, WCM>), dim3(tiles_x * mtiles_y * ta.split), dim3(MT), smem_m, st_mid, ta);
        TSP_HIP(hipGetLastError());
    }
    TSP_HIP(hipEventRecord(ctx->ev[5], st_mid));
    TSP_HIP(hipEventRecord(ctx->ev[9], st));
    const long long n_mega = (long long)hc.n_mega;
    const float4 *mega_geom = (const float4 *)ws.huge_geom + (ws.huge_capacity - n_mega);
    const float *mega_w = (const float *)ws.huge_w + (ws.huge_capacity - n_mega) * ((MODE == TSP_MODE_RGB) ? 2 : 1);
    ta.p_lo = 0.0f; ta.p_hi = __builtin_inff();
    if (hc.n_huge > 0) {
        ta.geom = (const float4 *)ws.huge_geom; ta.w = (const float *)ws.huge_w; ta.n_records = (long long)hc.n_huge;
        // rgb stays on kernel H: with three accumulators per pixel its per-pixel stencil set-up is shared by three FMAs
        // (0.49 clk per fragment at 2048^2), while H2 pays its per-strip set-up over 16-row strips (0.62) and H3 needs
        // three MFMAs per block and k-step (matrix-pipe-bound, 0.47)
        if (ctx->huge_variant == 0 || MODE == TSP_MODE_RGB) {
            if (MODE == TSP_MODE_RGB) rc = launch_huge<MODE, 3, 4>(ctx, ta, smem_h, (long long)hc.n_huge);
            else if (second_channel) rc = launch_huge<MODE, 2, 4>(ctx, ta, smem_h, (long long)hc.n_huge);
            else rc = launch_huge<MODE, 1, 4>(ctx, ta, smem_h, (long long)hc.n_huge);   // 4x8 px/lane measured slower (spills, larger tiles)
        } else {                                // kernel H2 (row-uniform gather): 64 px <= P < p_mega
            if (MODE == TSP_MODE_RGB) rc = launch_huge2<MODE, 3, 1, 16, 4>(ctx, ta, (long long)hc.n_huge);
            else if (second_channel) rc = launch_huge2<MODE, 2, 1, 16, 4>(ctx, ta, (long long)hc.n_huge);
            else if (ctx->huge_variant == 2) rc = launch_huge2<MODE, 1, 1, 32, 4>(ctx, ta, (long long)hc.n_huge);
            else rc = launch_huge2<MODE, 1, 1, 16, 6>(ctx, ta, (long long)hc.n_huge);   // 64x16 strips at 6 waves/SIMD: 17.3 ms against 18.6 for 64x32 at 4
        }
        if (rc) return rc;
    }
    TSP_HIP(hipEventRecord(ctx->ev[10], st));
    if (n_mega > 0) {                           // kernel H3 (matrix cores): P >= p_mega, the tail end of the huge list
        ta.geom = mega_geom; ta.w = mega_w; ta.n_records = n_mega;
        if (MODE == TSP_MODE_RGB) rc = launch_mega<MODE, 3, 2, 2>(ctx, ta, n_mega);
        else if (second_channel) rc = launch_mega<MODE, 2, 2, 3>(ctx, ta, n_mega);
        else if (ctx->huge_variant == 5) rc = launch_mega<MODE, 1, 2, 5>(ctx, ta, n_mega);
        else if (ctx->huge_variant == 6) rc = launch_mega<MODE, 1, 2, 6>(ctx, ta, n_mega);
        else if (ctx->huge_variant == 7) rc = launch_mega<MODE, 1, 1, 8>(ctx, ta, n_mega);
        else rc = launch_mega<MODE, 1, 2, 4>(ctx, ta, n_mega);   // 4 column blocks per strip measured no faster (18.8 vs 18.6 ms)
        if (rc) return rc;
    }
    TSP_HIP(hipEventRecord(ctx->ev[11], st));
    if (MODE == TSP_MODE_RGB && (hc.n_mid > 0 || hc.n_huge > 0 || n_mega > 0)) {
        if (ctx->overlap_mid_huge) TSP_HIP(hipStreamWaitEvent(st, ctx->ev[5], 0));
        if ((rc = add_rect_counts(ctx, (const float4 *)ws.mid_geom, (long long)hc.n_mid, (const float4 *)ws.huge_geom, (long long)hc.n_huge, mega_geom, n_mega))) return rc;
    }
    TSP_HIP(hipEventRecord(ctx->ev[6], st));
    if (ctx->overlap_mid_huge) TSP_HIP(hipStreamWaitEvent(st, ctx->ev[5], 0));     // join: later work on `st` sees both
    TSP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->stats.ms_stream = ms;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); ctx->stats.ms_mid = ms;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[9], ctx->ev[10])); ctx->stats.ms_huge = ms;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[10], ctx->ev[11])); ctx->stats.ms_mega = ms;
    return TSP_OK;
}

int launch_pipeline(tsp_context *ctx, const Camera &cam, const int64_t *h_starts, const int64_t *h_lens, int n_ranges,
                    int64_t total, int mode) {
    switch (mode) {
        case TSP_MODE_WEIGHTED: return run_pipeline<TSP_MODE_WEIGHTED>(ctx, cam, h_starts, h_lens, n_ranges, total);
        case TSP_MODE_DEPTH: return run_pipeline<TSP_MODE_DEPTH>(ctx, cam, h_starts, h_lens, n_ranges, total);
        case TSP_MODE_RGB: return run_pipeline<TSP_MODE_RGB>(ctx, cam, h_starts, h_lens, n_ranges, total);
    }
    set_error("bad mode %d", mode);
    return TSP_EINVAL;
}

}  // namespace tsp
