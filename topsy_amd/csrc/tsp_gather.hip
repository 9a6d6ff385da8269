// tsp_gather.hip -- the tile-gather kernels of the splat pipeline (footprints >= 64 px, bilinear sampling on mip 0), gfx950.
//
// What they compute is fragment_* + additive blend of the reference for its magnified footprints
// (src/topsy/shaders/sph.wgsl:139-165, sampler src/topsy/sph.py:425-426) in the arithmetic of tsp_math.h; the records they
// consume (pixel-space centre, width, weights) are written by kernel S (tsp_pipeline.hip).
//   kernel H   splat_huge_kernel   round-1 kernel, one 4x4 pixel block per lane, per-pixel bilinear stencil: rgb mode
//   kernel H2  splat_huge2_kernel  row-uniform gather, 64 px <= P < p_mega
//   kernel H3  splat_mega_kernel   outer products on the matrix cores, P >= p_mega
#include <algorithm>
#include <type_traits>

#include "tsp_pipeline.h"

namespace tsp {

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

#ifndef TSP_FOLD_H
#define TSP_FOLD_H 1024      // footprints a float32 accumulator of kernel H holds when it has no register totals (rgb)
#endif
#ifndef TSP_H_OCC3
#define TSP_H_OCC3 4         // waves per SIMD of kernel H's rgb build
#endif
constexpr int HT = 512;              // threads per workgroup of kernel H (8 waves share one quad table)
constexpr int HTILE_W = 128;         // its tile is 128 pixels wide: 32 lanes x 4 pixels

// NACC = value channels accumulated (1: density only, 2: density + weighted/depth, 3: rgb);
// PXH  = pixel rows per lane (4 or 8): the per-axis setup (12 instructions per row/column) is shared
//        by 4*PXH pixels, so the taller block costs ~30 % fewer instructions per pixel; it is used
//        when the accumulators still fit the 128-VGPR budget of a 512-thread workgroup (NACC == 1).
template <int MODE, int NACC, int PXH>
__global__ __launch_bounds__(HT, (NACC == 3) ? TSP_H_OCC3 : 4) void splat_huge_kernel(TileArgs a) {
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
    constexpr int FOLD_EVERY = REG_TOTALS ? 64 : TSP_FOLD_H;
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
    // (32-bit record indices: the launcher refuses lists of 2^31 records or more)
    const unsigned n_rec = (unsigned)a.n_records, n_runs = (n_rec + HDEAL - 1) / HDEAL, usplit = (unsigned)a.split;
    for (unsigned run0 = 0; run0 * usplit < n_runs; run0 += 256 / HDEAL) {
        // ---- waves 0-3 test 256 records against the tile and compact the hits into the LDS queue ----
        const unsigned ri = ((run0 + (tid & 255) / HDEAL) * usplit + sp) * HDEAL + (tid & (HDEAL - 1));
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        bool hit = false;
        if (tid < 256 && ri < n_rec) {
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
            if (REG_TOTALS) {
                // fold the short-run accumulators into the totals every 64 footprints: bounds the
                // float32 accumulation error at ~sqrt(64)*2^-24 per level instead of sqrt(n)
                if (++since_fold == FOLD_EVERY) {
                    since_fold = 0;
#pragma unroll
                    for (int p = 0; p < NPX; ++p)
#pragma unroll
                        for (int c = 0; c < NACC; ++c) { tot[p < NTOT ? p : 0][c] += acc[p][c]; acc[p][c] = 0.0f; }
                }
            }
        }
        if (!REG_TOTALS) {
            // no register totals (rgb): the accumulators go to the float64 target once they may hold FOLD_EVERY footprints
            // (<= FOLD_EVERY + 255: counted per 256-record batch) -- HERE, between the batches, not under the footprint loop,
            // where the 48 conditional atomics and their addresses cost the hot loop 300 bytes of scratch per lane
            since_fold += nq;
            if (since_fold >= FOLD_EVERY) {
                since_fold = 0;
                int Rl = R;
                asm volatile("" : "+s"(Rl));
#pragma unroll
                for (int ty = 0; ty < PXH; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 4; ++tx) {
                        const int p = ty * 4 + tx;
                        const int gxp = px0 + ((lg - tx) & 3);       // column slot tx
                        if (gxp < Rl && py0 + ty < Rl) {
                            double *d = a.img + ((size_t)(py0 + ty) * Rl + gxp) * C;
#pragma unroll
                            for (int c = 0; c < NACC; ++c) {
                                if (acc[p][c] != 0.0f) gatomic_add(d + c, acc[p][c]);
                                acc[p][c] = 0.0f;
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
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[2], n_frag); }
    }
}

template <int MODE, int NACC, int PXH>
static int launch_huge(tsp_context *ctx, TileArgs ta, size_t smem_h, long long n_huge) {
    TSP_REQUIRE(ta.n_records < (1ll << 31), TSP_EINVAL, "%lld deferred footprints in one render block (the tile-gather kernels index them with 32 bits)", ta.n_records);
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

// CNT: fragment counting compiled in (tsp_set_option "count_fragments"); the product instantiation carries none of it
template <int MODE, int NACC, int W, int HR, int OCC, bool CNT>
__global__ __launch_bounds__(H2T, OCC) void splat_huge2_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    constexpr int TW = 2 * 64 * W, TH = 2 * HR;            // tile: 2 x 2 wave strips of (64 W) x HR pixels
    constexpr int NG = HR / 4;                             // row groups (one quad of lanes carries a group's factors)
    static_assert(HR == 16 || HR == 32 || HR == 64, "rows per wave strip");
    typedef typename std::conditional<HR == 64, unsigned long long, unsigned>::type mask_t;     // one bit per pixel row of the strip
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
    float2 *rt = rt_all + wv * 64;
    const float2 *rt_quad = rt + (lane & 3);               // this lane's slot in every row group
    const int sx = tx0 + 64 * W * (wv & 1), sy = ty0 + HR * (wv >> 1);
    const float sx0 = (float)sx, sx1 = (float)(sx + 64 * W), sy0 = (float)sy, sy1 = (float)(sy + HR);
    float pxc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) pxc[w] = (sx + 64 * w + lane < R) ? (float)(sx + 64 * w + lane) + 0.5f : __builtin_inff();
    const int myrow = lane & (HR - 1);
    const float pyc_own = (sy + myrow < R) ? (float)(sy + myrow) + 0.5f : __builtin_inff();

    // float32 accumulators hold at most FOLD_EVERY footprints (rounding error ~ sqrt(n) * 2^-24 relative: < 2e-6 at
    // the worst pixel), then go to the float64 render target; second-level register totals (as kernel H keeps) would
    // cost HR * W more VGPRs and spill here
    constexpr int FOLD_EVERY = TSP_FOLD_EVERY;
    float acc[HR * W][NACC];
#pragma unroll
    for (int p = 0; p < HR * W; ++p)
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[p][c] = 0.0f;
    unsigned long long n_frag = 0;
    const char *PTb = reinterpret_cast<const char *>(PT);
    __syncthreads();                                       // the only workgroup barrier: from here on the waves run free
    if (sx >= R || sy >= R) return;                        // a strip wholly outside the image (R not a multiple of the tile)

    // Every wave scans the workgroup's share of the record list on its own, 64 records at a time (one per lane),
    // and keeps those whose square and disc reach ITS strip -- no shared queue, so no wave ever waits for another.
    // The four waves read the same records at about the same time (L1 / L2 hits).  Records are dealt to the `split`
    // workgroups of a tile in runs of HDEAL: consecutive records are spatial neighbours (consecutive chunks), so
    // every workgroup sees an even sample of the tile's footprints.
    // 32-bit record indices (the launcher refuses lists of 2^31 records or more)
    const unsigned n_rec = (unsigned)a.n_records, n_runs = (n_rec + HDEAL - 1) / HDEAL, usplit = (unsigned)a.split;
    auto fetch = [&](unsigned run0, float4 &g, float &gw1, float &gw2) {
        const unsigned ri = ((run0 + lane / HDEAL) * usplit + sp) * HDEAL + (lane & (HDEAL - 1));
        g = make_float4(0.f, 0.f, 0.f, 0.f); gw1 = gw2 = 0.0f;
        if (ri < n_rec) {
            g = a.geom[ri];
            gw1 = a.w[ri * NW];
            if (NW == 2) gw2 = a.w[ri * NW + 1];
        }
    };
    float4 g_next; float gw1_next, gw2_next;
    fetch(0, g_next, gw1_next, gw2_next);
    // The record loop is cut into segments of >= FOLD_EVERY footprints (<= FOLD_EVERY + 63); the accumulators go to the
    // float64 target between segments: ONE flush site, outside the hot loops (see splat_mega64_kernel)
    unsigned run0 = 0;
    do {
    int since_fold = 0;
    for (; run0 * usplit < n_runs && since_fold < FOLD_EVERY; run0 += 64 / HDEAL) {
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
        since_fold += __popcll(hits);
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
            mask_t covmask, chgmask, jmpmask;
            int r512;                                   // byte offset of this lane's texel row in PT
            {
                const float d = pyc_own - pcy;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float v = (d + half) * invP;
                const float tv = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tv);
                const float fr = (tv - f0) * cv;
                const int r = (int)f0;
                // texel row of the pixel row above = the value of the lane before (v_mov_b32_dpp wave_shr:1); lane 0 and
                // row 0 of the second half are excluded by `myrow > 0` below
                const int rprev = __builtin_amdgcn_mov_dpp(r, 0x138, 0xf, 0xf, false);
                r512 = r * (PT_STRIDE * 4);
                asm volatile("" ::: "memory");          // (in-order LDS: the previous footprint's table reads are done)
                rt[lane] = make_float2(fr, cv - fr);      // (every lane writes: the table has 64 slots per wave, the rows sit in the first HR)
                asm volatile("" ::: "memory");
                // the row masks straight from vector compares (as __ballot(bool expression) each costs a v_cndmask + v_cmp round trip);
                // the lanes that hold rows, and row 0 of the strip, are constants
                constexpr unsigned long long ROWS = (HR == 64) ? ~0ull : ((1ull << (HR & 63)) - 1ull);
                const unsigned long long cov64 = __builtin_amdgcn_fcmpf(__builtin_fabsf(d), half, 4 /* FCMP_OLT */) & ROWS;
                const unsigned long long chg64 = __builtin_amdgcn_uicmp((unsigned)r, (unsigned)rprev, 33 /* ICMP_NE */) & cov64 & ~1ull;
                // texel rows advance by at most one per pixel row when P >= 64; rounding at P ~ 64 may still skip one
                const unsigned long long jmp64 = __builtin_amdgcn_uicmp((unsigned)r, (unsigned)(rprev + 1), 33 /* ICMP_NE */) & chg64;
                covmask = (mask_t)cov64; chgmask = (mask_t)chg64; jmpmask = (mask_t)jmp64;
            }
            if (covmask == 0) continue;
            // the first covered row's texel rows are loaded before the row walk (below): it is never a "change"
            const mask_t first = covmask & ((mask_t)0 - covmask);
            chgmask &= ~first; jmpmask &= ~first;
            // row factors of group k for the DPP broadcast: lane l takes rows 4k + (l & 3)
            // HR = 16: the four groups' factors sit in registers; HR = 32: two registers pairs in turn (group K + 1 loads while
            // group K is walked), 12 VGPRs fewer -- what lets the 64 x 32 strips run at 6 waves per SIMD
            constexpr bool JIT = (HR >= 32);
            constexpr int NRF = JIT ? 2 : NG;
            float2 rowf[NRF];
            if constexpr (JIT) rowf[0] = rt_quad[0];
            else {
#pragma unroll
                for (int k = 0; k < NG; ++k) rowf[k] = rt_quad[4 * k];
            }
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
                if (CNT) ncov_x += (cv != 0.0f);
            }
            float top[W], bot[W];
            float2 nxt[W];                              // prefetched pair of texel row r + 2
            auto pair_at = [&](int w, int byteoff) -> float2 {
                const float *t = reinterpret_cast<const float *>(PTb + byteoff + caddr[w]);
                return make_float2(t[0], t[1]);
            };
            auto lerp = [&](int w, float2 t) -> float { return __builtin_fmaf(t.y, fxs[w], t.x * gxs[w]); };
            // texel rows of the first covered pixel row (wave-uniform byte offset of its texel row in PT), the pair after them in flight
            int r_off = __builtin_amdgcn_readlane(r512, (HR == 64 ? __ffsll((long long)covmask) : __ffs((int)covmask)) - 1);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                top[w] = lerp(w, pair_at(w, r_off)); bot[w] = lerp(w, pair_at(w, r_off + PT_STRIDE * 4));
                nxt[w] = pair_at(w, r_off + 2 * PT_STRIDE * 4);
            }
            // ONE way to advance a texel row -- top = bot, bot = the x-interpolated prefetched pair, the next pair loads -- so that the
            // rolling registers never meet a second definition at a control-flow merge (with a separate reload-from-scratch path for
            // the first row and for skips the compiler copied the pair aside on every change: three v_mov).  Where float32 rounding
            // at P ~ 64 makes the texel row skip one, the step runs twice.
            auto row_step = [&]() {
                r_off += PT_STRIDE * 4;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    // top = bot; bot = nxt.x * gx + nxt.y * fx -- in place (left to the compiler the new row lands in a third register
                    // and is moved: one more v_mov per texel-row change)
                    asm volatile("v_mov_b32 %0, %1\n\tv_mul_f32 %1, %2, %4\n\tv_fmac_f32 %1, %3, %5"
                                 : "=&v"(top[w]), "+v"(bot[w]) : "v"(nxt[w].x), "v"(nxt[w].y), "v"(gxs[w]), "v"(fxs[w]));
                    nxt[w] = pair_at(w, r_off + 2 * PT_STRIDE * 4);
                }
            };
            auto row_change = [&](int /*ty*/, bool skip) {      // `skip` is wave-uniform
                if (skip) row_step();                           // (rare; first, so that the common step below ends at the join)
                row_step();
            };
            // DPP hazard (gfx9: a VGPR written by a VALU instruction may not be read as a DPP operand in the next two issue
            // slots).  The DPP operands below are the row factors: they come from LDS (no VALU write) long before their use,
            // and the compiler does not see inside the asm statements, so pin them in registers here and leave two wait
            // states; the other operands of the DPP FMAs (top, bot) are ordinary sources and carry no such restriction.
            if constexpr (!JIT) {
#pragma unroll
                for (int k = 0; k < NG; ++k) asm volatile("" : "+v"(rowf[k].x), "+v"(rowf[k].y));
                asm volatile("s_nop 1");
            }
#define TSP_H2_ROW(K, T)                                                                                       \
            {                                                                                                  \
                constexpr int ty_ = 4 * (K) + (T);                                                             \
                if ((chgmask >> ty_) & 1) row_change(ty_, ((jmpmask >> ty_) & 1) != 0);                       \
                _Pragma("unroll") for (int w = 0; w < W; ++w) {                                                \
                    float *ac = acc[ty_ * W + w];                                                              \
                    if (NACC == 1) {                                                                           \
                        fmac_quad<T>(ac[0], rowf[JIT ? ((K) & 1) : (K)].y, top[w]);                                                \
                        fmac_quad<T>(ac[0], rowf[JIT ? ((K) & 1) : (K)].x, bot[w]);                                                \
                    } else {                                                                                   \
                        float kv = mul_quad<T>(rowf[JIT ? ((K) & 1) : (K)].y, top[w]);                                             \
                        fmac_quad<T>(kv, rowf[JIT ? ((K) & 1) : (K)].x, bot[w]);                                                   \
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
            if constexpr ((K) < NG) {                                                                          \
                if constexpr (JIT && (K) + 1 < NG) rowf[((K) + 1) & 1] = rt_quad[4 * ((K) + 1)];               \
                if (((covmask >> (4 * (K))) & 15) != 0) {                                                    \
                    if constexpr (JIT) asm volatile("" : "+v"(rowf[(K) & 1].x), "+v"(rowf[(K) & 1].y));        \
                    TSP_H2_ROW(K, 0) TSP_H2_ROW(K, 1) TSP_H2_ROW(K, 2) TSP_H2_ROW(K, 3)                          \
                }                                                                                              \
            }
            TSP_H2_GROUP(0) TSP_H2_GROUP(1) TSP_H2_GROUP(2) TSP_H2_GROUP(3)
            TSP_H2_GROUP(4) TSP_H2_GROUP(5) TSP_H2_GROUP(6) TSP_H2_GROUP(7)
            TSP_H2_GROUP(8) TSP_H2_GROUP(9) TSP_H2_GROUP(10) TSP_H2_GROUP(11)
            TSP_H2_GROUP(12) TSP_H2_GROUP(13) TSP_H2_GROUP(14) TSP_H2_GROUP(15)
#undef TSP_H2_GROUP
#undef TSP_H2_ROW
            if (CNT) n_frag += (unsigned long long)(ncov_x * __popcll((unsigned long long)covmask));
#ifdef TSP_H2_DEBUG      // analysis build: (footprint, strip) pairs, covered rows and texel-row changes instead of the S / M / H3 fragment counts
            if (CNT && lane == 0) {
                atomicAdd(&a.cnt->n_frag_class[0], 1ull);
                atomicAdd(&a.cnt->n_frag_class[1], (unsigned long long)__popcll((unsigned long long)covmask));
                atomicAdd(&a.cnt->n_frag_class[3], (unsigned long long)__popcll((unsigned long long)chgmask));
            }
#endif
        }
    }
    // ---- add this wave's partial strip into the render target ---------------------------------------
    {
        int Rl = R;                                   // laundered: the row offsets are formed here, not hoisted out of the record loop
        asm volatile("" : "+s"(Rl));
        double *img = a.img + ((size_t)sy * Rl + (sx + lane)) * C;
        asm volatile("" : "+v"(img));
#pragma unroll
        for (int ty = 0; ty < HR; ++ty)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const int p = ty * W + w, gx = sx + 64 * w + lane, gy = sy + ty;
                if (gx < Rl && gy < Rl) {
                    double *d = img + ((size_t)ty * Rl + 64 * w) * C;
#pragma unroll
                    for (int c = 0; c < NACC; ++c) {
                        if (acc[p][c] != 0.0f) gatomic_add(d + c, acc[p][c]);
                        acc[p][c] = 0.0f;
                    }
                }
            }
    }
    } while (run0 * usplit < n_runs);
    if (CNT) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[2], n_frag); }
    }
}

template <int MODE, int NACC, int W, int HR, int OCC>
static int launch_huge2(tsp_context *ctx, TileArgs ta, long long n_huge) {
    TSP_REQUIRE(ta.n_records < (1ll << 31), TSP_EINVAL, "%lld deferred footprints in one render block (the tile-gather kernels index them with 32 bits)", ta.n_records);
    const size_t smem = (size_t)((PT_ROWS * PT_STRIDE + 3) & ~3) * sizeof(float) + (H2T / 64) * 64 * sizeof(float2);
    const int tw = 2 * 64 * W, th = 2 * HR;
    const int htiles_x = (ctx->R + tw - 1) / tw, htiles_y = (ctx->R + th - 1) / th;
    const int htiles = htiles_x * htiles_y;
    const long long batches = (n_huge + 63) / 64;
    int split = ctx->huge_split;
    // many short workgroups: a wave lives ~1 ms at split 64 and the tail of the launch (tiles differ 10x in work)
    // cost 2.5 ms of 21; measured 64 -> 128: 21.9 -> 19.5 ms, 256: 19.2 ms, 512: 22.5 ms
    if (split <= 0) {
        split = std::max(1, (ctx->cu_count * 128 + htiles - 1) / htiles);
        // a small render block: fewer, longer workgroups (each loads the kernel image) in proportion below 2^16 records
        if (n_huge < (1ll << 16)) split = std::max(32, (int)((long long)split * n_huge >> 16));
    }
    split = (int)std::min<long long>(split, std::max<long long>(batches, 1));
    ta.split = split;
    ta.tiles_x = htiles_x;
    if (ta.count_frag) hipLaunchKernelGGL((splat_huge2_kernel<MODE, NACC, W, HR, OCC, true>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    else hipLaunchKernelGGL((splat_huge2_kernel<MODE, NACC, W, HR, OCC, false>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
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
template <int MODE, int NACC, int NB, int OCC, bool CNT>
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
    constexpr int FOLD_EVERY = TSP_FOLD_EVERY;              // as kernel H2: float32 accumulators hold <= 512 footprints
    f32x16 acc[NACC][NB];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[c][b][v] = 0.0f;
    unsigned long long n_frag = 0;
    const char *PTb = reinterpret_cast<const char *>(PT);
    __syncthreads();                                       // the only workgroup barrier
    if (sx >= R || sy >= R) return;                        // a strip wholly outside the image (last_row would be negative)

    auto flush = [&]() {
        int Rl = R;                                        // laundered: see splat_mega64_kernel
        asm volatile("" : "+s"(Rl));
        double *img = a.img + ((size_t)(sy + 4 * kh) * Rl + (sx + li)) * C;
        asm volatile("" : "+v"(img));
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = (v >> 2) * 8 + (v & 3);    // + 4 * kh (in img)
                if (sx + 32 * b + li < Rl && sy + 4 * kh + row < Rl) {
                    double *d = img + ((size_t)row * Rl + 32 * b) * C;
#pragma unroll
                    for (int c = 0; c < NACC; ++c) {
                        if (acc[c][b][v] != 0.0f) gatomic_add(d + c, acc[c][b][v]);
                        acc[c][b][v] = 0.0f;
                    }
                }
            }
    };

    const unsigned n_rec = (unsigned)a.n_records, n_runs = (n_rec + HDEAL - 1) / HDEAL, usplit = (unsigned)a.split;
    auto fetch = [&](unsigned run0, float4 &g, float &gw1, float &gw2) {
        const unsigned ri = ((run0 + lane / HDEAL) * usplit + sp) * HDEAL + (lane & (HDEAL - 1));
        g = make_float4(0.f, 0.f, 0.f, 0.f); gw1 = gw2 = 0.0f;
        if (ri < n_rec) {
            g = a.geom[ri];
            gw1 = a.w[ri * NW];
            if (NW == 2) gw2 = a.w[ri * NW + 1];
        }
    };
    float4 g_next; float gw1_next, gw2_next;
    fetch(0, g_next, gw1_next, gw2_next);
    unsigned run0 = 0;
    do {                                                   // segments of >= FOLD_EVERY footprints, one flush site (see splat_mega64_kernel)
    int since_fold = 0;
    for (; run0 * usplit < n_runs && since_fold < FOLD_EVERY; run0 += 64 / HDEAL) {
        const float4 g = g_next;
        const float gw1 = gw1_next, gw2 = gw2_next;
        fetch(run0 + 64 / HDEAL, g_next, gw1_next, gw2_next);
        const float g_half = 0.5f * g.z;
        bool hit;
        {
            const float sdx = fmaxf(fmaxf(sx0 - g.x, g.x - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - g.y, g.y - sy1), 0.0f);
            hit = g.z >= a.p_lo && g.z < a.p_hi && sdx < g_half && sdy < g_half && !(a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * g.z * g.z);
        }
        unsigned long long hits = __ballot(hit);
        if (hits == 0ull) continue;
        since_fold += __popcll(hits);
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
                if (CNT) {
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
            if constexpr (NACC == 1) {
                for (int m = 0; m < nsteps; ++m) {
                    const float A = (rel == kk) ? gy : ((rel + 1 == kk) ? fy : 0.0f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float *t = reinterpret_cast<const float *>(PTb + rowoff + caddr[b]);
                        const float L = __builtin_fmaf(t[1], fxs[b], t[0] * gxs[b]);
                        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, L, acc[0][b], 0, 0, 0);
                    }
                    kk += 2;
                    rowoff += 2 * PT_STRIDE * 4;
                }
            } else {
                // Several channels share the kernel image of the footprint: the matrix cores form it ONCE in a scratch
                // accumulator (one MFMA per k-step and block instead of one per channel), then every channel takes its multiple of
                // it on the VALU (16 FMAs per channel and block) -- rgb: a third of the MFMAs.
                f32x16 kimg[NB];
                for (int m = 0; m < nsteps; ++m) {
                    const float A = (rel == kk) ? gy : ((rel + 1 == kk) ? fy : 0.0f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float *t = reinterpret_cast<const float *>(PTb + rowoff + caddr[b]);
                        const float L = __builtin_fmaf(t[1], fxs[b], t[0] * gxs[b]);
                        if (m == 0) {
                            f32x16 zero;
#pragma unroll
                            for (int v = 0; v < 16; ++v) zero[v] = 0.0f;
                            kimg[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, L, zero, 0, 0, 0);
                        } else {
                            kimg[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, L, kimg[b], 0, 0, 0);
                        }
                    }
                    kk += 2;
                    rowoff += 2 * PT_STRIDE * 4;
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        acc[0][b][v] = __builtin_fmaf(kimg[b][v], w0, acc[0][b][v]);
                        acc[NACC >= 2 ? 1 : 0][b][v] = __builtin_fmaf(kimg[b][v], w1, acc[NACC >= 2 ? 1 : 0][b][v]);
                        if (NACC >= 3) acc[NACC - 1][b][v] = __builtin_fmaf(kimg[b][v], w2, acc[NACC - 1][b][v]);
                    }
            }
        }
    }
    flush();
    } while (run0 * usplit < n_runs);
    if (CNT) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[3], n_frag); }
    }
}

// Density-only variant of kernel H3 with 64 x 64 wave strips (two row blocks of two column blocks): the column factors and
// the footprint's parameters are prepared once for 4096 pixels instead of 2048 (experiment, `mega_variant` = 1).
template <int MODE, int OCC, bool CNT>
__global__ __launch_bounds__(H2T, OCC) void splat_mega64_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NB = 2, NR = 2;
    constexpr int TW = 128, TH = 128;                      // tile: 2 x 2 wave strips of 64 x 64 pixels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *PT = smem;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int R = a.cam.R;
    const int tile_id = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tx0 = (tile_id % a.tiles_x) * TW, ty0 = (tile_id / a.tiles_x) * TH;
    for (int i = tid; i < PT_ROWS * PT_STRIDE; i += H2T) {
        const int j = min(i / PT_STRIDE, 63), x = min(i % PT_STRIDE, 63);
        PT[i] = a.mips[j * 64 + x];
    }
    const int sx = tx0 + 64 * (wv & 1), sy = ty0 + 64 * (wv >> 1);
    const float sx0 = (float)sx, sx1 = (float)(sx + 64), sy0 = (float)sy, sy1 = (float)(sy + 64);
    float pyc[NR], pxc[NB];
    int last_row[NR];
#pragma unroll
    for (int rb = 0; rb < NR; ++rb) {
        pyc[rb] = (sy + 32 * rb + li < R) ? (float)(sy + 32 * rb + li) + 0.5f : __builtin_inff();
        last_row[rb] = min(31, R - 1 - (sy + 32 * rb));    // < 0: the row block lies outside the image
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) pxc[b] = (sx + 32 * b + li < R) ? (float)(sx + 32 * b + li) + 0.5f : __builtin_inff();
    // a wave of this kernel sees ~500 footprints per launch (the mega list is short and split over >= 96 workgroups per tile), so
    // with 1024 per segment nearly every wave flushes ONCE, at the end: the float64 flush atomics are this kernel's only HBM write
    // traffic (2.5 GB per launch at 512, profiles/round4a); float32 sums of <= 1087 terms: ~2e-6 relative at worst
    constexpr int FOLD_EVERY = 2 * TSP_FOLD_EVERY;
    f32x16 acc[NR][NB];
#pragma unroll
    for (int rb = 0; rb < NR; ++rb)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[rb][b][v] = 0.0f;
    unsigned long long n_frag = 0;
    const char *PTb = reinterpret_cast<const char *>(PT);
    __syncthreads();
    if (sx >= R || sy >= R) return;

    auto flush = [&]() {
        // R is laundered here so that the 64 row / block offsets below are formed at the flush, not hoisted out of the record
        // loop as loop invariants (they filled the scalar file and spilled into VGPR lanes)
        int Rl = R;
        asm volatile("" : "+s"(Rl));
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) {
            double *img = a.img + ((size_t)(sy + 32 * rb + 4 * kh) * Rl + (sx + li)) * C;
            asm volatile("" : "+v"(img));
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = (v >> 2) * 8 + (v & 3);
                    if (sx + 32 * b + li < Rl && sy + 32 * rb + 4 * kh + row < Rl) {
                        double *d = img + ((size_t)row * Rl + 32 * b) * C;
                        if (acc[rb][b][v] != 0.0f) gatomic_add(d, acc[rb][b][v]);
                        acc[rb][b][v] = 0.0f;
                    }
                }
        }
    };

    // 32-bit record indices (the launcher refuses lists of 2^31 records or more)
    const unsigned n_rec = (unsigned)a.n_records, n_runs = (n_rec + HDEAL - 1) / HDEAL, usplit = (unsigned)a.split;
    auto fetch = [&](unsigned run0, float4 &g) {
        const unsigned ri = ((run0 + lane / HDEAL) * usplit + sp) * HDEAL + (lane & (HDEAL - 1));
        g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ri < n_rec) g = a.geom[ri];
    };
    float4 g_next;
    fetch(0, g_next);
    // The record loop is cut into segments of >= FOLD_EVERY footprints (<= FOLD_EVERY + 63), the accumulators go to the
    // float64 target BETWEEN segments: one flush site outside the hot loops.  (With the flush inlined under the innermost
    // loop the register allocator parked three of the four accumulator blocks in scratch around every 64-record batch:
    // 6.6 GB of spill writes per launch, profiles/round3g.)
    unsigned run0 = 0;
    do {
    int since_fold = 0;
    for (; run0 * usplit < n_runs && since_fold < FOLD_EVERY; run0 += 64 / HDEAL) {
        const float4 g = g_next;
        fetch(run0 + 64 / HDEAL, g_next);
        const float g_half = 0.5f * g.z;
        bool hit;
        {
            const float sdx = fmaxf(fmaxf(sx0 - g.x, g.x - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - g.y, g.y - sy1), 0.0f);
            hit = g.z >= a.p_lo && g.z < a.p_hi && sdx < g_half && sdy < g_half && !(a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * g.z * g.z);
        }
        unsigned long long hits = __ballot(hit);
        if (hits == 0ull) continue;
        since_fold += __popcll(hits);
        const float g_invP = 1.0f / g.z;
        while (hits) {
            const int src = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const float pcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), src));
            const float pcy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.y), src));
            const float half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_half), src));
            const float invP = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_invP), src));
            const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.w), src));
            // ---- columns (B operand), shared by the row blocks ----
            int caddr[NB];
            float fxs[NB], gxs[NB];
            int ncx = 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float d = pxc[b] - pcx;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float u = (d + half) * invP;
                const float tu = __builtin_amdgcn_fmed3f(__builtin_fmaf(u, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tu);
                const float fr = (tu - f0) * cv;
                caddr[b] = ((int)f0) * 4;
                fxs[b] = fr * w0;
                gxs[b] = (cv - fr) * w0;
                if (CNT) ncx += (cv != 0.0f) ? 1 : 0;
            }
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                if (last_row[rb] < 0) continue;
                const float d = pyc[rb] - pcy;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const unsigned long long rows = __ballot(cv != 0.0f && kh == 0);
                if (rows == 0ull) continue;                           // the footprint misses this row block
                const float v = (d + half) * invP;
                const float tv = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tv);
                const float fy = (tv - f0) * cv, gy = cv - fy;
                const int r = (int)f0;
                const int r0 = __builtin_amdgcn_readlane(r, 0);
                const int rel = r - r0;
                const int kmax = __builtin_amdgcn_readlane(rel, last_row[rb]);
                const int nsteps = (kmax + 3) >> 1;
                if (CNT && kh == 0) n_frag += (unsigned long long)(ncx * __popcll(rows));
                int rowoff = (r0 + kh) * (PT_STRIDE * 4);
                int kk = kh;
                for (int m = 0; m < nsteps; ++m) {
                    const float A = (rel == kk) ? gy : ((rel + 1 == kk) ? fy : 0.0f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float *t = reinterpret_cast<const float *>(PTb + rowoff + caddr[b]);
                        const float L = __builtin_fmaf(t[1], fxs[b], t[0] * gxs[b]);
                        acc[rb][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, L, acc[rb][b], 0, 0, 0);
                    }
                    kk += 2;
                    rowoff += 2 * PT_STRIDE * 4;
                }
            }
        }
    }
    flush();
    } while (run0 * usplit < n_runs);
    if (CNT) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[3], n_frag); }
    }
}

template <int MODE, int OCC>
static int launch_mega64(tsp_context *ctx, TileArgs ta, long long n_huge) {
    TSP_REQUIRE(ta.n_records < (1ll << 31), TSP_EINVAL, "%lld deferred footprints in one render block (the tile-gather kernels index them with 32 bits)", ta.n_records);
    const size_t smem = (size_t)((PT_ROWS * PT_STRIDE + 3) & ~3) * sizeof(float);
    const int htiles_x = (ctx->R + 127) / 128, htiles_y = (ctx->R + 127) / 128;
    const int htiles = htiles_x * htiles_y;
    const long long batches = (n_huge + 63) / 64;
    int split = ctx->mega_split;
    if (split <= 0) split = std::max(1, (ctx->cu_count * 32 + htiles - 1) / htiles);      // 128 at 1024^2 (96 ... 160 alike; 256: +10 %)
    split = (int)std::min<long long>(split, std::max<long long>(batches, 1));
    ta.split = split;
    ta.tiles_x = htiles_x;
    if (ta.count_frag) hipLaunchKernelGGL((splat_mega64_kernel<MODE, OCC, true>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    else hipLaunchKernelGGL((splat_mega64_kernel<MODE, OCC, false>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

template <int MODE, int NACC, int NB, int OCC>
static int launch_mega(tsp_context *ctx, TileArgs ta, long long n_huge) {
    TSP_REQUIRE(ta.n_records < (1ll << 31), TSP_EINVAL, "%lld deferred footprints in one render block (the tile-gather kernels index them with 32 bits)", ta.n_records);
    const size_t smem = (size_t)((PT_ROWS * PT_STRIDE + 3) & ~3) * sizeof(float);
    const int htiles_x = (ctx->R + 64 * NB - 1) / (64 * NB), htiles_y = (ctx->R + 63) / 64;
    const int htiles = htiles_x * htiles_y;
    const long long batches = (n_huge + 63) / 64;
    int split = ctx->mega_split;
    if (split <= 0) split = std::max(1, (ctx->cu_count * 64 + htiles - 1) / htiles);
    split = (int)std::min<long long>(split, std::max<long long>(batches, 1));
    ta.split = split;
    ta.tiles_x = htiles_x;
    if (ta.count_frag) hipLaunchKernelGGL((splat_mega_kernel<MODE, NACC, NB, OCC, true>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    else hipLaunchKernelGGL((splat_mega_kernel<MODE, NACC, NB, OCC, false>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// ---------------------------------------------------------------------------------------------
// kernel H4: footprints >= 64 px on the matrix cores, 16-row strips (v_mfma_f32_16x16x4_f32)
// ---------------------------------------------------------------------------------------------
// The same outer-product form as kernel H3 -- img[row][col] += sum_k U[row][k] V[k][col] over the texel rows k the strip
// touches -- but on 16 x 16 pixel blocks with FOUR texel rows per instruction.  What an MFMA costs is K slots times the block
// area, whether or not a slot's texel row meets a pixel row, so the useful fraction is 2 / (rows / t + 2) for a texel t
// pixels tall: a 16-row block wastes half as many slots on the "+ 2" as a 32-row block, and for t >= 8 px one instruction
// (32 cycles) covers a block.  A wave owns a 64 x 16 strip = four 16 x 16 blocks side by side: 16 accumulator registers
// for a density render, so eight waves fit a SIMD.  Lane l = (j = l & 15, k = l >> 4):
//   rows    lane evaluates pixel row j (its copies k = 0..3 alike) and supplies A[j][k] = gy / fy / 0 for texel row r0 + 4 m + k;
//   columns lane l evaluates pixel column l ONCE per footprint and leaves (texel column, fx w, gx w) in a per-wave LDS
//           table; block b then reads entry 16 b + j and supplies B[k][j] = the x-interpolated texel row r0 + 4 m + k there.
// Barrier-free like H2 / H3 (per-wave record scans); several channels share the kernel image as in H3.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PT4_ROWS = 72;         // 64 texel rows + clamp-to-edge copies of row 63: the k-steps read up to row 63 + 2 + 3 (+ slack)

template <int MODE, int NACC, int OCC>
__global__ __launch_bounds__(H2T, OCC) void splat_tile4_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    constexpr int NB = 4;                                    // 16-column blocks per wave strip
    constexpr int TW = 128, TH = 32;                         // tile: 2 x 2 wave strips of 64 x 16 pixels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *PT = smem;                                        // [PT4_ROWS][PT_STRIDE] level-0 kernel image, clamp-to-edge padded
    float4 *ct_all = reinterpret_cast<float4 *>(smem + ((PT4_ROWS * PT_STRIDE + 3) & ~3));   // per wave: 64 column entries
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lj = lane & 15, lk = lane >> 4;                // MFMA operand roles: row / column index, k index
    const int R = a.cam.R;
    const int tile_id = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tx0 = (tile_id % a.tiles_x) * TW, ty0 = (tile_id / a.tiles_x) * TH;
    for (int i = tid; i < PT4_ROWS * PT_STRIDE; i += H2T) {
        const int j = min(i / PT_STRIDE, 63), x = min(i % PT_STRIDE, 63);
        PT[i] = a.mips[j * 64 + x];
    }
    float4 *ct = ct_all + wv * 64;
    const int sx = tx0 + 64 * (wv & 1), sy = ty0 + 16 * (wv >> 1);
    const float sx0 = (float)sx, sx1 = (float)(sx + 64), sy0 = (float)sy, sy1 = (float)(sy + 16);
    const float pyc = (sy + lj < R) ? (float)(sy + lj) + 0.5f : __builtin_inff();
    const float pxc = (sx + lane < R) ? (float)(sx + lane) + 0.5f : __builtin_inff();
    const int last_row = min(15, R - 1 - sy);                // last pixel row of the strip inside the image (wave-uniform)
    constexpr int FOLD_EVERY = TSP_FOLD_EVERY;
    f32x4 acc[NACC][NB];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[c][b][v] = 0.0f;
    unsigned long long n_frag = 0;
    int since_fold = 0;
    const char *PTb = reinterpret_cast<const char *>(PT);
    __syncthreads();                                         // the only workgroup barrier
    if (sx >= R || sy >= R) return;

    auto flush = [&]() {
        double *img = a.img + ((size_t)(sy + 4 * lk) * R + (sx + lj)) * C;
        asm volatile("" : "+v"(img));
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (sx + 16 * b + lj < R && sy + 4 * lk + v < R) {
                    double *d = img + ((size_t)v * R + 16 * b) * C;
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
            hit = g.z >= a.p_lo && g.z < a.p_hi && sdx < g_half && sdy < g_half && !(a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * g.z * g.z);
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
            // ---- columns: lane l evaluates pixel column l and publishes it in the wave's table ----
            int ncx = 0;
            {
                const float d = pxc - pcx;
                const float cv = (__builtin_fabsf(d) < half) ? 1.0f : 0.0f;
                const float u = (d + half) * invP;
                const float tu = __builtin_amdgcn_fmed3f(__builtin_fmaf(u, 64.0f, -0.5f), 0.0f, 63.0f);
                const float f0 = __builtin_floorf(tu);
                const float fr = (tu - f0) * cv;
                const float fxw = (NACC == 1) ? fr * w0 : fr;          // density: the particle weight rides on the column factors
                const float gxw = (NACC == 1) ? (cv - fr) * w0 : (cv - fr);
                asm volatile("" ::: "memory");                          // (the previous footprint's table reads are done: in-order LDS)
                ct[lane] = make_float4(__int_as_float(((int)f0) * 4), fxw, gxw, 0.0f);
                asm volatile("" ::: "memory");
                if (a.count_frag) ncx = __popcll(__ballot(cv != 0.0f));
            }
            // ---- rows (A operand): lane (j, k) evaluates pixel row j; canonical texel coordinate (tsp_math.h) ----
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
                nsteps = (kmax + 5) >> 2;                             // texel rows r0 .. r0 + kmax + 1, four per MFMA
                if (a.count_frag) {
                    const unsigned long long rows = __ballot(cv != 0.0f && lk == 0);
                    if (lane == 0) n_frag += (unsigned long long)(ncx * __popcll(rows));
                }
            }
            float4 col[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) col[b] = ct[16 * b + lj];
            int rowoff = (r0 + lk) * (PT_STRIDE * 4);   // this lane's texel row of the current k-step (bytes)
            int kk = lk;
            if constexpr (NACC == 1) {
                for (int m = 0; m < nsteps; ++m) {
                    const float A = (rel == kk) ? gy : ((rel + 1 == kk) ? fy : 0.0f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float *t = reinterpret_cast<const float *>(PTb + rowoff + __float_as_int(col[b].x));
                        const float L = __builtin_fmaf(t[1], col[b].y, t[0] * col[b].z);
                        acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, L, acc[0][b], 0, 0, 0);
                    }
                    kk += 4;
                    rowoff += 4 * PT_STRIDE * 4;
                }
            } else {
                f32x4 kimg[NB];
                for (int m = 0; m < nsteps; ++m) {
                    const float A = (rel == kk) ? gy : ((rel + 1 == kk) ? fy : 0.0f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float *t = reinterpret_cast<const float *>(PTb + rowoff + __float_as_int(col[b].x));
                        const float L = __builtin_fmaf(t[1], col[b].y, t[0] * col[b].z);
                        if (m == 0) {
                            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                            kimg[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, L, zero, 0, 0, 0);
                        } else {
                            kimg[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, L, kimg[b], 0, 0, 0);
                        }
                    }
                    kk += 4;
                    rowoff += 4 * PT_STRIDE * 4;
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        acc[0][b][v] = __builtin_fmaf(kimg[b][v], w0, acc[0][b][v]);
                        acc[NACC >= 2 ? 1 : 0][b][v] = __builtin_fmaf(kimg[b][v], w1, acc[NACC >= 2 ? 1 : 0][b][v]);
                        if (NACC >= 3) acc[NACC - 1][b][v] = __builtin_fmaf(kimg[b][v], w2, acc[NACC - 1][b][v]);
                    }
            }
            if (++since_fold == FOLD_EVERY) { since_fold = 0; flush(); }
        }
    }
    flush();
    if (a.count_frag) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[2], n_frag); }
    }
}

template <int MODE, int NACC, int OCC>
static int launch_tile4(tsp_context *ctx, TileArgs ta, long long n_records) {
    const size_t smem = (size_t)((PT4_ROWS * PT_STRIDE + 3) & ~3) * sizeof(float) + (H2T / 64) * 64 * sizeof(float4);
    const int htiles_x = (ctx->R + 127) / 128, htiles_y = (ctx->R + 31) / 32;
    const int htiles = htiles_x * htiles_y;
    const long long batches = (n_records + 63) / 64;
    int split = ctx->huge_split;
    if (split <= 0) split = std::max(1, (ctx->cu_count * 128 + htiles - 1) / htiles);
    split = (int)std::min<long long>(split, std::max<long long>(batches, 1));
    ta.split = split;
    ta.tiles_x = htiles_x;
    hipLaunchKernelGGL((splat_tile4_kernel<MODE, NACC, OCC>), dim3(htiles * split), dim3(H2T), smem, ctx->stream, ta);
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// ---------------------------------------------------------------------------------------------
// host side: which kernel takes which class
// ---------------------------------------------------------------------------------------------
template <int MODE>
static int launch_gather_mode(tsp_context *ctx, TileArgs ta, bool second_channel, const float4 *huge_geom, const float *huge_w,
                              long long n_huge, const float4 *mega_geom, const float *mega_w, long long n_mega) {
    hipStream_t st = ctx->stream;
    int rc = TSP_OK;
    ta.p_lo = 0.0f; ta.p_hi = __builtin_inff();
    if (n_huge > 0) {
        ta.geom = huge_geom; ta.w = huge_w; ta.n_records = n_huge;
        // rgb stays on kernel H: with three accumulators per pixel its per-pixel stencil set-up is shared by three FMAs
        // (0.49 clk per fragment at 2048^2), while H2 pays its per-strip set-up over 16-row strips (0.62) and H3 needs
        // three MFMAs per block and k-step (0.47)
        // rgb: kernel H2 with three accumulator sets since round 4 (116 VGPRs once its flush left the hot loop: 13.0 ms against
        // kernel H's 14.8 ms for the 64-128 px band of config 4); huge_variant 0 keeps kernel H for A/B
        if (MODE == TSP_MODE_RGB && ctx->huge_variant != 0) {
            if (ctx->huge_variant == 4) rc = launch_huge2<MODE, 3, 1, 16, 4>(ctx, ta, n_huge);
            else rc = launch_huge2<MODE, 3, 1, 16, 5>(ctx, ta, n_huge);      // 96 VGPRs at 5 waves/SIMD: 11.5 against 12.5 ms at 4 (config 4)
        } else if (ctx->huge_variant == 0 || MODE == TSP_MODE_RGB) {
            const size_t smem_h = (size_t)(64 * 64 + 512) * sizeof(float4);
            if (MODE == TSP_MODE_RGB) rc = launch_huge<MODE, 3, 4>(ctx, ta, smem_h, n_huge);
            else if (second_channel) rc = launch_huge<MODE, 2, 4>(ctx, ta, smem_h, n_huge);
            else rc = launch_huge<MODE, 1, 4>(ctx, ta, smem_h, n_huge);   // 4x8 px/lane measured slower (spills, larger tiles)
        } else if (ctx->huge_variant == 3) {    // kernel H4 (matrix cores, 16-row strips): 64 px <= P < p_mega
            if (second_channel) rc = launch_tile4<MODE, 2, 5>(ctx, ta, n_huge);
            else rc = launch_tile4<MODE, 1, 6>(ctx, ta, n_huge);
        } else {                                // kernel H2 (row-uniform gather): 64 px <= P < p_mega
            if (second_channel) {
                // 72 VGPRs at 7 waves/SIMD (24 B of scratch outside the row loop): 9.48 against 9.78 ms at 6 (80 VGPRs), 10.8 at 8 (spills)
                if (ctx->huge_variant == 4) rc = launch_huge2<MODE, 2, 1, 16, 4>(ctx, ta, n_huge);
                else rc = launch_huge2<MODE, 2, 1, 16, 7>(ctx, ta, n_huge);
            }
            else if (ctx->huge_variant == 2) rc = launch_huge2<MODE, 1, 1, 32, 6>(ctx, ta, n_huge);
            else if (ctx->huge_variant == 4) rc = launch_huge2<MODE, 1, 1, 16, 7>(ctx, ta, n_huge);
            else if (ctx->huge_variant == 5) rc = launch_huge2<MODE, 1, 1, 16, 8>(ctx, ta, n_huge);
            else if (ctx->huge_variant == 6) rc = launch_huge2<MODE, 1, 1, 32, 7>(ctx, ta, n_huge);
            // Round 4, once the flush had left the hot loop: 64x32 strips with the row factors fetched group by group -- half as
            // many (footprint, strip) pairs to set up -- at 8 waves/SIMD (64 VGPRs; three reloads per 64-record batch): this kernel
            // is latency-bound per wave, occupancy is what pays.  1.25e8 particles, records 64-768 px: 10.20 / 9.86 / 9.59 ms at
            // 6 / 7 / 8 waves (1e9: 28.7 / 26.9 / 26.9); 64x16 strips at 8: 10.7.  With fewer records the shorter strips' finer work
            // units win (3.4e5 records: 3.05 against 3.49 ms)
            else if (ctx->huge_variant == 7 || (ctx->huge_variant == 1 && n_huge >= 700000)) rc = launch_huge2<MODE, 1, 1, 32, 8>(ctx, ta, n_huge);
            else rc = launch_huge2<MODE, 1, 1, 16, 8>(ctx, ta, n_huge);
        }
        if (rc) return rc;
    }
    TSP_HIP(hipEventRecord(ctx->ev[10], st));
    if (n_mega > 0) {                           // kernel H3 (matrix cores): P >= p_mega, the tail end of the huge list
        ta.geom = mega_geom; ta.w = mega_w; ta.n_records = n_mega;
        // option integrated_px: kernel I takes the records at least that wide, the matrix cores what is left of the mega list
        // below it (rgb 128 px ... integrated_px; nothing when the mode's own boundary is not below it)
        const bool integ = integrated_active(ctx);
        const float pm_mode = (MODE == TSP_MODE_RGB) ? (ctx->rgb_mega_variant > 0 ? ctx->p_mega_rgb : 0.0f) : (second_channel ? ctx->p_mega2 : ctx->p_mega);
        const bool mfma_part = !integ || (pm_mode > 0.0f && pm_mode < ctx->integrated_px);
        if (integ) {
            rc = launch_integrated(ctx, ta, mega_geom, mega_w, MODE == TSP_MODE_RGB ? 2 : (second_channel ? 1 : 0), n_mega, ctx->integrated_px);
            if (rc) return rc;
            ta.p_hi = ctx->integrated_px;
        }
        if (!mfma_part) {
        } else if constexpr (MODE == TSP_MODE_RGB) {
            // three accumulator sets (96 registers for a 64 x 32 strip): 2-3 waves per SIMD of the 512-entry register file
            if (ctx->rgb_mega_variant == 1) rc = launch_mega<MODE, 3, 2, 2>(ctx, ta, n_mega);
            else if (ctx->rgb_mega_variant == 2) rc = launch_mega<MODE, 3, 2, 3>(ctx, ta, n_mega);
            else if (ctx->rgb_mega_variant == 4) rc = launch_mega<MODE, 3, 1, 4>(ctx, ta, n_mega);
            else rc = launch_mega<MODE, 3, 1, 3>(ctx, ta, n_mega);
        } else if (second_channel) {
            // 64x32 strips at 4 waves/SIMD (spills 112 B outside the k-loop) 13.16 ms against 13.37 at 3 (1.25e8 weighted); with
            // few records the narrower 32x32 strips (more, shorter workgroups) win: 1e7 weighted 4.08 against 4.49 / 4.66 ms
            if (ctx->mega_variant == 3) rc = launch_mega<MODE, 2, 2, 3>(ctx, ta, n_mega);
            else if (ctx->mega_variant == 5 || (ctx->mega_variant == 0 && n_mega < 100000)) rc = launch_mega<MODE, 2, 1, 4>(ctx, ta, n_mega);
            else rc = launch_mega<MODE, 2, 2, 4>(ctx, ta, n_mega);
        }
        // density: 64 x 64 strips (column factors and parameters prepared once per 4096 pixels) once there are enough records to
        // keep their fewer, longer workgroups busy: 1.25e8 particles 6.3 -> 6.0 ms, 1e9: 18.2 -> 17.3 ms, but 1e7: 2.1 -> 2.25 ms
        else if (ctx->mega_variant == 2 || (ctx->mega_variant == 0 && n_mega >= 40000)) rc = launch_mega64<MODE, 4>(ctx, ta, n_mega);
        else if (ctx->mega_variant == 3) rc = launch_mega64<MODE, 3>(ctx, ta, n_mega);
        else if (ctx->mega_variant == 4) rc = launch_mega<MODE, 1, 2, 5>(ctx, ta, n_mega);
        // (round 4: the records >= 256 px through kernel H2 on LARGER strips -- fewer (footprint, strip) pairs to set up -- lose to
        // their lower occupancy: 64x32 at 6 waves/SIMD 8.6 ms, 128x32 at 4: 10.4, 64x64 at 4: 12.9 for the same records)
        else rc = launch_mega<MODE, 1, 2, 4>(ctx, ta, n_mega);   // 4 column blocks per strip and 5-6 waves/SIMD measured no faster
        if (rc) return rc;
    }
    TSP_HIP(hipEventRecord(ctx->ev[11], st));
    return TSP_OK;
}

int launch_gather_kernels(tsp_context *ctx, TileArgs ta, int mode, bool second_channel, const float4 *huge_geom, const float *huge_w,
                          long long n_huge, const float4 *mega_geom, const float *mega_w, long long n_mega) {
    switch (mode) {
        case TSP_MODE_WEIGHTED: return launch_gather_mode<TSP_MODE_WEIGHTED>(ctx, ta, second_channel, huge_geom, huge_w, n_huge, mega_geom, mega_w, n_mega);
        case TSP_MODE_DEPTH: return launch_gather_mode<TSP_MODE_DEPTH>(ctx, ta, true, huge_geom, huge_w, n_huge, mega_geom, mega_w, n_mega);
        case TSP_MODE_RGB: return launch_gather_mode<TSP_MODE_RGB>(ctx, ta, true, huge_geom, huge_w, n_huge, mega_geom, mega_w, n_mega);
    }
    set_error("bad mode %d", mode);
    return TSP_EINVAL;
}

}  // namespace tsp
