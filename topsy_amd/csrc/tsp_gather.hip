// tsp_gather.hip -- the strip-gather kernels of the splat pipeline, gfx950: the footprints kernel S defers (tsp_pipeline.hip).
//
// What they compute is fragment_* + additive blend of the reference (src/topsy/shaders/sph.wgsl:139-165, sampler
// src/topsy/sph.py:425-426) in the arithmetic of tsp_math.h; the records they consume (pixel-space centre, width, weights) are
// written by kernel S.  A wave owns a strip of the image, its accumulators sit in registers:
//   kernel H2  splat_huge2_kernel          row-uniform gather: every footprint >= 64 px (bilinear on mip 0), every mode
//   kernel N   splat_narrow_gather_kernel  the mid footprints (16-64 px, nearest on mips 0-3), four records per wave step (round 6)
//   kernel G   splat_mid_gather_kernel     round 5's mid kernel, one record per wave step (option mid_narrow_px_milli < 64000)
//   + the binning passes: huge_band_fill_kernel (H2), tile_count / tile_prefix / tile_fill_kernel (N, G)
// (Rounds 1-4 also carried the per-pixel gather kernel H, the matrix-core kernels H3 / H4 and the option kernel I: none was
// selected by a default rule -- f32 MFMA has no peak advantage over the VALU on gfx950 and H2 issues half the flop; kernel I is
// exact only to ~1e-6 of a footprint's peak -- so round 5 removed them; HISTORY.md keeps their designs and measurements.)
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "tsp_pipeline.h"

namespace tsp {

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
// against ~14.5 VALU + one 16-byte LDS read per pixel in a per-pixel bilinear gather (the round-1 kernel H).  The sum has the same non-negative terms as the
// canonical bilinear form in a different association (relative rounding differences of ~1e-7).

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

    // (the wave index as a scalar: the strip's origin and edges then live in scalar registers -- as vector values two of them were
    // spilled and re-read from scratch for every 64 records, behind the record prefetch they then waited for)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = a.cam.R;
    // (blockIdx = tile * split + sp: an XCD-aware order -- the workgroups resident on one XCD walking several tiles per slice of the
    // record list, so that the list crosses the fabric once per tile group -- measured 7-11 % SLOWER at 1e9 particles: round 5)
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

    // float32 accumulators hold at most FOLD_EVERY (2048) footprints, then go to the float64 render target: a sum of n non-negative
    // terms accumulates ~sqrt(n) 2^-24 of relative rounding error at the worst pixel; measured on a 4e7-particle sample of the 1e9
    // snapshot the image's largest relative error is 2.9e-7 / 3.1e-7 / 3.3e-7 at 512 / 1024 / 2048 (it is the final float32 rounding
    // that shows), while the float64 flush atomics were 2/3 of the rgb render's HBM writes at 512.  Second-level register totals
    // would cost HR * W more VGPRs and spill here
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
    // With band bins (huge_band_fill_kernel) the tile scans only the records whose squares reach its 64-row image band:
    // ~1/3 of the list for the reference h-law at 1024^2, the same records in the same dealing
    const float4 *geom = a.geom;
    const float *wts = a.w;
    unsigned n_rec = (unsigned)a.n_records;
    if (a.hband_count) {
        const int band = ty0 / HBAND_H;
        geom += (size_t)band * a.hband_stride;
        wts += (size_t)band * a.hband_stride * NW;
        n_rec = (unsigned)a.hband_count[band];
    }
    const unsigned n_runs = (n_rec + HDEAL - 1) / HDEAL, usplit = (unsigned)a.split, n_last = max(n_rec, 1u) - 1u;
    // record index of this lane in batch run0: ((run0 + lane / HDEAL) * split + sp) * HDEAL + lane % HDEAL = a wave-uniform base + lane_off
    const unsigned lane_off = (unsigned)(lane / HDEAL) * usplit * HDEAL + (unsigned)(lane & (HDEAL - 1));      // (loop-invariant)
    auto batch_base = [&](unsigned run0) -> unsigned { return (run0 * usplit + sp) * HDEAL; };
    auto fetch = [&](unsigned run0, float4 &g, float &gw1, float &gw2) {
        // unconditional loads (a slot past the end re-reads the last record; the batch empties such slots when it starts): under a
        // branch the compiler cannot count the loads in flight and waits for this prefetch right after issuing it
        const unsigned rc = min(batch_base(run0) + lane_off, n_last);
        g = geom[rc];
        gw1 = (NACC >= 2) ? wts[rc * NW] : 0.0f;
        gw2 = (NW == 2) ? wts[rc * NW + 1] : 0.0f;
    };
    float4 g_next; float gw1_next, gw2_next;
    fetch(0, g_next, gw1_next, gw2_next);
    // (the first batch lands before the loop: with loads of the preheader still in flight at the loop header the compiler's
    // counter model gives up and waits for every prefetch right after issuing it -- vmcnt(0) at the top of each batch: round 6)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    // The record loop is cut into segments of >= FOLD_EVERY footprints (<= FOLD_EVERY + 63); the accumulators go to the
    // float64 target between segments: ONE flush site, outside the hot loops (under the innermost loop
    // its offsets filled the scalar file and the accumulators were parked in scratch around every 64-record batch)
    unsigned run0 = 0;
    do {
    int since_fold = 0;
    for (; run0 * usplit < n_runs && since_fold < FOLD_EVERY; run0 += 64 / HDEAL) {
        float4 g = g_next;
        const float gw1 = gw1_next, gw2 = gw2_next;
        if (batch_base(run0) + lane_off >= n_rec) g.z = 0.0f;      // (a slot past the end of the list)
        fetch(run0 + 64 / HDEAL, g_next, gw1_next, gw2_next);      // the next 64 records load while these are rasterised
        const float g_half = 0.5f * g.z;
        bool hit;
        {
            const float sdx = fmaxf(fmaxf(sx0 - g.x, g.x - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - g.y, g.y - sy1), 0.0f);
            // g.z = 0 marks an empty slot; the kernel vanishes outside the disc inscribed in the footprint square
            hit = g.z > 0.0f && sdx < g_half && sdy < g_half && !(a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * g.z * g.z);
        }
        unsigned long long hits = __ballot(hit);
        if (hits == 0ull) continue;
        since_fold += __popcll(hits);
        const float g_invP = 1.0f / g.z;
        const float g_w1 = (MODE == TSP_MODE_RGB) ? gw1 : g.w * gw1;
        // The rows of NH = 64 / HR hits are evaluated in ONE pass: lane j works out pixel row j % HR for hit j / HR (with HR = 32 the
        // upper half of the wave used to repeat the lower half's work); the hits' parameters reach their lanes through ds_bpermute
        // (the LDS crossbar, not the vector ALU) instead of a v_readlane each per hit.  Columns and the row walk follow hit by hit.
        constexpr int NH = 64 / HR;
        while (hits) {
            int srcs[NH], n_h = 0;
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                srcs[i] = i ? srcs[0] : 0;
                if (hits) { srcs[i] = __ffsll((long long)hits) - 1; hits &= hits - 1; n_h = i + 1; }
            }
            // ---- rows: lane j evaluates row j % HR of hit j / HR and the texel row of the row above it -----------------
            unsigned long long cov64, chg64, jmp64;
            int r512;                                   // byte offset of this lane's texel row in PT
            {
                int src_l = srcs[0];
#pragma unroll
                for (int i = 1; i < NH; ++i) src_l = (lane >= HR * i) ? srcs[i] : src_l;
                const int bp = src_l << 2;
                const float pcy_l = __int_as_float(__builtin_amdgcn_ds_bpermute(bp, __float_as_int(g.y)));
                const float half_l = __int_as_float(__builtin_amdgcn_ds_bpermute(bp, __float_as_int(g_half)));
                const float invP_l = __int_as_float(__builtin_amdgcn_ds_bpermute(bp, __float_as_int(g_invP)));
                const float d = pyc_own - pcy_l;
                const float cv = (__builtin_fabsf(d) < half_l) ? 1.0f : 0.0f;
                // the canonical float32 texel coordinate (tsp_math.h: the oracle forms it with the same operations -- a value next to
                // a zero texel is proportional to its fraction, so even one ulp of difference here shows at 1e-5 relative)
                const float v = (d + half_l) * invP_l;
                const float tv = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, 64.0f, -0.5f), 0.0f, 63.0f);
                const float fr = __builtin_amdgcn_fractf(tv) * cv;      // (tv - floor(tv), exact for 0 <= tv <= 63, in one instruction)
                const int r = (int)tv;                  // (tv >= 0: the conversion truncates = floor)
                // texel row of the pixel row above = the value of the lane before (v_mov_b32_dpp wave_shr:1); row 0 of every hit
                // is excluded below
                const int rprev = __builtin_amdgcn_mov_dpp(r, 0x138, 0xf, 0xf, false);
                r512 = __mul24(r, PT_STRIDE * 4);      // (v_mul_u32_u24: full rate; v_mul_lo_u32 takes four issue slots)
                asm volatile("" ::: "memory");          // (in-order LDS: the previous footprints' table reads are done)
                rt[lane] = make_float2(fr, cv - fr);      // (the table has 64 slots per wave: HR rows for each of the NH hits)
                asm volatile("" ::: "memory");
                // the row masks straight from vector compares (as __ballot(bool expression) each costs a v_cndmask + v_cmp round trip)
                constexpr unsigned long long ROW0S = (HR == 64) ? 1ull : ((HR == 32) ? 0x0000000100000001ull : 0x0001000100010001ull);
                cov64 = __builtin_amdgcn_fcmpf(__builtin_fabsf(d), half_l, 4 /* FCMP_OLT */);
                chg64 = __builtin_amdgcn_uicmp((unsigned)r, (unsigned)rprev, 33 /* ICMP_NE */) & cov64 & ~ROW0S;
                // texel rows advance by at most one per pixel row when P >= 64; rounding at P ~ 64 may still skip one
                jmp64 = __builtin_amdgcn_uicmp((unsigned)r, (unsigned)(rprev + 1), 33 /* ICMP_NE */) & chg64;
            }
            for (int hh = 0; hh < n_h; ++hh) {
            int src = srcs[0];
#pragma unroll
            for (int i = 1; i < NH; ++i) src = (hh == i) ? srcs[i] : src;
            constexpr unsigned long long ROWS = (HR == 64) ? ~0ull : ((1ull << (HR & 63)) - 1ull);      // (the bits above belong to the next hit)
            mask_t covmask = (mask_t)((cov64 >> ((HR * hh) & 63)) & ROWS), chgmask = (mask_t)((chg64 >> ((HR * hh) & 63)) & ROWS),
                   jmpmask = (mask_t)((jmp64 >> ((HR * hh) & 63)) & ROWS);
            // the footprint's parameters, wave-uniform (scalar registers)
            const float pcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), src));
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
            const float2 *rt_quad_h = rt_quad + HR * hh;      // this hit's rows of the table
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
            if constexpr (JIT) rowf[0] = rt_quad_h[0];
            else {
#pragma unroll
                for (int k = 0; k < NG; ++k) rowf[k] = rt_quad_h[4 * k];
            }
            // ---- columns: W per lane ----
            int caddr[W];
            float fxs[W], gxs[W];
            int ncov_x = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const float d = pxc[w] - pcx;
                const bool covered = __builtin_fabsf(d) < half;
                const float u = (d + half) * invP;
                const float tu = __builtin_amdgcn_fmed3f(__builtin_fmaf(u, 64.0f, -0.5f), 0.0f, 63.0f);
                caddr[w] = ((int)tu) * 4;
                // uncovered column: both weights 0.  Density: the particle weight rides on the column factors, so a pixel costs two FMAs
                // (fr * (c w) and (1 - fr) * (c w) with c = 0 or 1: the same bits as (fr c) * w and (c - fr c) * w)
                const float cw = covered ? ((NACC == 1) ? wq.x : 1.0f) : 0.0f;
                const float fr = __builtin_amdgcn_fractf(tu);
                fxs[w] = fr * cw;
                gxs[w] = (1.0f - fr) * cw;
                if (CNT) ncov_x += covered ? 1 : 0;
            }
            float top[W], bot[W];
            float2 nxt[W];                              // prefetched pair of texel row r + 2
            auto pair_at = [&](int w, int byteoff) -> float2 {
                const float *t = reinterpret_cast<const float *>(PTb + byteoff + caddr[w]);
                return make_float2(t[0], t[1]);
            };
            auto lerp = [&](int w, float2 t) -> float { return __builtin_fmaf(t.y, fxs[w], t.x * gxs[w]); };
            // texel rows of the first covered pixel row (wave-uniform byte offset of its texel row in PT), the pair after them in flight
            int r_off = __builtin_amdgcn_readlane(r512, (HR == 64 ? __ffsll((long long)covmask) : __ffs((int)covmask)) - 1 + HR * hh);
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
            // footprints of 64 px and up, whose texel rows change every 1-8 pixel rows below 512 px.)
            // (Round 5: a second, test-free copy of a group's eight FMAs for the groups without a texel-row change -- the per-row bit
            // tests are three quarters of this kernel's scalar instructions -- made the register allocator spill around every
            // footprint's set-up at the control-flow merges: 32.7 -> 49 ms at 1e9 particles.  One code path per row it stays.)
#define TSP_H2_GROUP(K)                                                                                        \
            if constexpr ((K) < NG) {                                                                          \
                if constexpr (JIT && (K) + 1 < NG) rowf[((K) + 1) & 1] = rt_quad_h[4 * ((K) + 1)];               \
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
#ifdef TSP_H2_DEBUG      // analysis build: (footprint, strip) pairs, covered rows and texel-row changes instead of the S / G fragment counts
            if (CNT && lane == 0) {
                atomicAdd(&a.cnt->n_frag_class[0], 1ull);
                atomicAdd(&a.cnt->n_frag_class[1], (unsigned long long)__popcll((unsigned long long)covmask));
                unsigned long long groups = 0;        // row groups of four with a covered row (each costs 8 FMAs per column register)
                for (int k = 0; k < NG; ++k) groups += (((unsigned long long)covmask >> (4 * k)) & 15ull) ? 1ull : 0ull;
                atomicAdd(&a.cnt->n_frag_class[3], (unsigned long long)__popcll((unsigned long long)chgmask) + (groups << 32));
            }
#endif
            }      // (hits of this row pass)
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

// ---------------------------------------------------------------------------------------------
// kernel G: the MID footprints (nearest sampling on mips 0-3, < 64 px) as a register gather
// ---------------------------------------------------------------------------------------------
// The same wave-owns-a-strip structure as kernel H2, for the nearest-texel rule: a lane owns one pixel COLUMN of a 64 x HR strip, the
// accumulators of its HR pixels sit in registers.  Per (footprint, strip) pair a wave evaluates the LUT row of every pixel row once (one
// row per lane, redistributed through a per-wave LDS table so that a quad of lanes carries the four rows of a group) and the LUT column
// + weight of every pixel column once (one per lane); a covered pixel row then costs
//       address = row address (DPP operand, quad_perm) + column offset ;  k = LUT[address] (LDS read) ;  acc += k * weight
// -- two vector instructions and a 4-byte LDS read per 64 pixels, against a multiply, a float64 conversion and a 9-clock ds_add_f64 per
// 64 pixels (plus their share of the row / column set-up) in the scatter kernel this one replaced (kernel M, round 5: HISTORY.md), whose LDS atomics bound it.
// Records: the mid list binned per strip (bin_mid_records below); every wave draws one work item of equal record count.
// strip height and waves per SIMD of kernel G by accumulator sets (1e9 density: 10.6 ms at 7 waves, 10.0 at 8; 64 x 16 strips 11.2)
#ifndef TSP_G_OCC1
#define TSP_G_OCC1 8
#define TSP_G_HR1 32
#define TSP_G_OCC2 8
#define TSP_G_OCC3 5
#endif
constexpr int GCHUNK_MAX = 1024;          // records per work item of kernel G (fewer for short lists: enough items to fill the device)
#ifndef TSP_BIN_PER
#define TSP_BIN_PER 4          // records per thread of the mid binning passes
#endif
constexpr int G_WIN_TILES = 4096;         // strips per LDS window of the binning passes (a window = whole rows of strips)
constexpr int G_LDS_TILES = 16384;        // the binning passes keep their tile counters in LDS up to this many tiles (global atomics beyond): 128 KB in the fill pass
                                          // (kernel N's 16 x 16 strips at 2048^2; with global atomics its rgb binning took 20 ms instead of 1)

template <int MODE, int NACC, int HR, int OCC, bool QUAD, bool CNT>
__global__ __launch_bounds__(H2T, OCC) void splat_mid_gather_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    constexpr int NG = HR / 4;
    static_assert(HR == 16 || HR == 32, "rows per wave strip");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the mip pyramid, or (QUAD: the kernel image is mirror-symmetric) the top-left quadrant of each level -- 5.4 KB instead of 21.8:
    // with the whole pyramid seven workgroups fill a CU's LDS, and a workgroup keeps its share until its LAST wave ends (the four
    // strips of a tile differ in work): on average four waves per SIMD were resident, not seven
    constexpr int TSIZE = QUAD ? MIPQ_TOTAL : MIP_TOTAL;
    float *T = smem;
    int *rt_all = reinterpret_cast<int *>(smem + TSIZE);               // per wave: LDS address of the LUT row of each of its HR pixel rows
    typedef const __attribute__((address_space(3))) float LdsF;
    const int T_lds = (int)(unsigned)(unsigned long long)(LdsF *)T;

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = a.cam.R;
    // every WAVE draws its own work item: item_records consecutive records of one strip's bin (four items per workgroup, which
    // shares the LUT; the items of a launch are equal in size, so its waves end together and their slots refill as whole workgroups)
    const int n_items = a.item_base[a.n_tiles];
    if ((int)blockIdx.x * (H2T / 64) >= n_items) return;
    const int item = min((int)blockIdx.x * (H2T / 64) + wv, n_items - 1);
    const bool idle_wave = (int)blockIdx.x * (H2T / 64) + wv >= n_items;       // (the last workgroup may have fewer than four items)
    const int strip = a.item_tile[item];
    const int chunk = item - a.item_base[strip];
    if (QUAD) {
        for (int i = tid; i < MIPQ_TOTAL; i += H2T) {
            const int lvl = i < 1024 ? 0 : (i < 1280 ? 1 : (i < 1344 ? 2 : 3));
            const int hn = 32 >> lvl, k = i - mipq_offset(lvl);
            T[i] = a.mips[mip_offset(lvl) + (k / hn) * (2 * hn) + (k % hn)];
        }
    } else {
        for (int i = tid; i < MIP_TOTAL; i += H2T) T[i] = a.mips[i];
    }
    int *rt = rt_all + wv * 64;
    const int *rt_quad = rt + (lane & 3);
    const int sx = (strip % a.tiles_x) * 64, sy = (strip / a.tiles_x) * HR;
    const float sx0 = (float)sx, sx1 = (float)(sx + 64), sy0 = (float)sy, sy1 = (float)(sy + HR);
    const float pxc = (sx + lane < R) ? (float)(sx + lane) + 0.5f : __builtin_inff();
    const int myrow = lane & (HR - 1);
    const float pyc_own = (sy + myrow < R) ? (float)(sy + myrow) + 0.5f : __builtin_inff();

    constexpr int FOLD_EVERY = TSP_FOLD_EVERY;
    float acc[HR][NACC];
#pragma unroll
    for (int p = 0; p < HR; ++p)
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[p][c] = 0.0f;
    unsigned long long n_frag = 0;
    __syncthreads();                                       // the only workgroup barrier
    if (idle_wave) return;

    const size_t first = (size_t)a.hband_base[strip] + (size_t)chunk * a.item_records;
    const float4 *geom = a.geom + first;
    const float *wts = a.w + first * NW;
    const unsigned n_rec = (unsigned)min(a.item_records, a.hband_count[strip] - chunk * a.item_records);
    auto fetch = [&](unsigned b0, float4 &g, float &gw1, float &gw2) {      // records b0 + lane of the item (one per lane)
        const unsigned rc = min(b0 + lane, n_rec - 1u);       // (unconditional loads, as in kernel H2; an item holds >= 1 record)
        g = geom[rc];
        gw1 = (NACC >= 2) ? wts[rc * NW] : 0.0f;
        gw2 = (NW == 2) ? wts[rc * NW + 1] : 0.0f;
    };
    float4 g_next; float gw1_next, gw2_next;
    fetch(0, g_next, gw1_next, gw2_next);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see kernel H2
    unsigned run0 = 0;                    // first record of the next batch of 64
    do {
    int since_fold = 0;
    for (; run0 < n_rec && since_fold < FOLD_EVERY; run0 += 64) {
        float4 g = g_next;
        const float gw1 = gw1_next, gw2 = gw2_next;
        if (run0 + lane >= n_rec) g.z = 0.0f;                 // (a slot past the end of the item)
        fetch(run0 + 64, g_next, gw1_next, gw2_next);
        const float g_half = 0.5f * g.z;
        bool hit;
        {
            const float sdx = fmaxf(fmaxf(sx0 - g.x, g.x - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - g.y, g.y - sy1), 0.0f);
            hit = g.z > 0.0f && sdx < g_half && sdy < g_half && !(a.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= a.disc_k2 * g.z * g.z);
        }
        unsigned long long hits = __ballot(hit);
        if (hits == 0ull) continue;
        since_fold += __popcll(hits);
        const float g_invP = 1.0f / g.z;
        const int g_lvl = max(level_for(g.z), 0);
        const float g_w1 = (MODE == TSP_MODE_RGB) ? gw1 : g.w * gw1;
        while (hits) {
            const int src = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const float pcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), src));
            const float pcy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.y), src));
            const float half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_half), src));
            const float invP = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_invP), src));
            const int lvl = __builtin_amdgcn_readlane(g_lvl, src);
            float wq[3];
            wq[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.w), src));
            wq[1] = (NACC >= 2) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_w1), src)) : 0.0f;
            wq[2] = (NACC >= 3) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gw2), src)) : 0.0f;
            const int n = 64 >> lvl, tshift = QUAD ? 5 - lvl : 6 - lvl;
            const int tbase = T_lds + (QUAD ? mipq_offset(lvl) : mip_offset(lvl)) * 4;
            // ---- rows: lane j < HR evaluates the LUT row of pixel row j (the canonical nearest-texel rule, tsp_math.h) ----
            unsigned covmask;
            {
                const float d = pyc_own - pcy;
                int ty = floor_clamp_s(((d + half) * invP) * (float)n, n - 1);      // = nearest_index((d + half) * invP, n), tsp_math.h
                if (QUAD) ty = min(ty, n - 1 - ty);
                asm volatile("" ::: "memory");          // (in-order LDS: the previous footprint's table reads are done)
                rt[lane] = tbase + (ty << (tshift + 2));
                asm volatile("" ::: "memory");
                constexpr unsigned long long ROWS = (1ull << HR) - 1ull;
                covmask = (unsigned)(__builtin_amdgcn_fcmpf(__builtin_fabsf(d), half, 4 /* FCMP_OLT */) & ROWS);
            }
            if (covmask == 0) continue;
            // ---- this lane's column: texel column (byte offset in a LUT row) and weights (+0 where the column is not covered) ----
            int tx4;
            float wl[NACC];
            {
                const float d = pxc - pcx;
                const bool covered = __builtin_fabsf(d) < half;
                int tx = floor_clamp_s(((d + half) * invP) * (float)n, n - 1);
                if (QUAD) tx = min(tx, n - 1 - tx);
                tx4 = tx * 4;
#pragma unroll
                for (int c = 0; c < NACC; ++c) wl[c] = covered ? wq[c] : 0.0f;
                if (CNT) n_frag += covered ? (unsigned long long)__popc(covmask) : 0ull;
            }
            // ---- row walk: groups of four rows; a group's row addresses sit in the quads (lane l: row 4 k + (l & 3)) ----
            int roq[2];
            roq[0] = rt_quad[0];
#define TSP_G_ROW(K, T_)                                                                                        \
            if ((covmask >> (4 * (K) + (T_))) & 1) {                                                            \
                _Pragma("unroll") for (int c = 0; c < NACC; ++c) fmac_plain(acc[4 * (K) + (T_)][c], kv[T_], wl[c]); \
            }
#define TSP_G_GROUP(K)                                                                                          \
            if constexpr ((K) < NG) {                                                                           \
                if constexpr ((K) + 1 < NG) roq[((K) + 1) & 1] = rt_quad[4 * ((K) + 1)];                         \
                if (((covmask >> (4 * (K))) & 15u) != 0u) {                                                     \
                    asm volatile("" : "+v"(roq[(K) & 1]));                                                      \
                    int ad[4]; float kv[4];                                                                     \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(0) : "=v"(ad[0]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(1) : "=v"(ad[1]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(2) : "=v"(ad[2]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(3) : "=v"(ad[3]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t) kv[t] = *reinterpret_cast<LdsF *>(ad[t]);    \
                    TSP_G_ROW(K, 0) TSP_G_ROW(K, 1) TSP_G_ROW(K, 2) TSP_G_ROW(K, 3)                               \
                }                                                                                               \
            }
            TSP_G_GROUP(0) TSP_G_GROUP(1) TSP_G_GROUP(2) TSP_G_GROUP(3)
            TSP_G_GROUP(4) TSP_G_GROUP(5) TSP_G_GROUP(6) TSP_G_GROUP(7)
#undef TSP_G_GROUP
#undef TSP_G_ROW
        }
    }
    // ---- add this wave's partial strip into the render target ---------------------------------------
    {
        int Rl = R;
        asm volatile("" : "+s"(Rl));
        double *img = a.img + ((size_t)sy * Rl + (sx + lane)) * C;
        asm volatile("" : "+v"(img));
#pragma unroll
        for (int ty = 0; ty < HR; ++ty) {
            const int gx = sx + lane, gy = sy + ty;
            if (gx < Rl && gy < Rl) {
                double *d = img + ((size_t)ty * Rl) * C;
#pragma unroll
                for (int c = 0; c < NACC; ++c) {
                    if (acc[ty][c] != 0.0f) gatomic_add(d + c, acc[ty][c]);
                    acc[ty][c] = 0.0f;
                }
            }
        }
    }
    } while (run0 < n_rec);
    if (CNT) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[1], n_frag); }
    }
}


// ---------------------------------------------------------------------------------------------
// kernel N: the NARROW mid footprints (below ~32 px) -- four records per wave step
// ---------------------------------------------------------------------------------------------
// Kernel G gives a whole wave to one (footprint, strip) pair: for a footprint of 18 px a quarter of its 64 pixel columns are covered
// and the pair's set-up (seven v_readlane, row table, column weights) costs as much as its rows -- 3.4 SIMD-clocks per fragment at
// 16 px against 0.86 at 48 px (tools/gpu_huge_classes.py, round 6), and zoomed cameras turn most of a snapshot into such records.
// Here a wave owns a strip of 16 columns x HR rows and its four DPP rows of 16 lanes are four record SLOTS: lane (s, c) holds the
// partial sums of pixel column c over slot s's records.  A step draws four records of the strip's bin at once:
//   * every lane loads ITS slot's record (16 lanes read one address): no v_readlane, every parameter is a vector value;
//   * lane (s, c) evaluates the LUT row of pixel rows c and c + 16 for its slot's record -> per-slot row table in LDS (the address
//     of the row, or of a block of zeros when the record does not cover that pixel row), and its column + weight;
//   * the row walk is wave-uniform over the UNION of the four records' rows, a quad of lanes carrying the four rows of a group as
//     in kernel G (v_add_u32_dpp quad_perm + a 4-byte LDS read + v_fmac per row); a slot that does not cover a row reads zeros.
// One instruction stream per four pairs instead of one per pair.  The bins hold only the records that reach the strip (the binning
// passes make kernel G's per-pair square-and-disc test themselves: `exact`), so no lane waits for another slot's miss.  A record whose
// weight is not finite would turn "0 x weight" into NaN where only another slot covers a row: such a step is drawn slot by slot.
// At the end of the item the four slots' partial strips are summed across the DPP rows and go to the float64 target.
constexpr int NSW = 16;                   // pixel columns of kernel N's strips (= lanes per record slot)
// Kernel N's LUT in LDS (mirror-symmetric kernel image: the top-left quadrants).  A ds_read_b32 is served in two groups of 32 lanes
// over 32 banks, i.e. two record slots per group, and two slots reading two different LUT rows collide: 35 % of the LDS cycles of
// the first version were bank conflicts (SQ_LDS_BANK_CONFLICT), on a kernel whose LDS pipe is busy 3/4 of the time.  So the
// quadrants are stored TWICE, interleaved in LINES of 32 floats: floats 0-15 of a line serve the even slots, 16-31 (the same
// values) the odd slots -- a slot only ever touches its own 16 banks.  A LUT row takes whole half-lines (16 floats): level 0 (32
// floats per quadrant row) two, levels 1-3 (16, 8, 4 floats) one each, so that a row's address is base(level) + (ty << 7 or 8);
// then two lines of zeros (what a slot reads in a pixel row its record does not cover).
constexpr int NQ_L1 = 64, NQ_L2 = 80, NQ_L3 = 88, NQ_ZERO = 92, NQ_LINES = 94;       // first line of levels 1, 2, 3, of the zeros; lines in all

template <int MODE, int NACC, int HR, int OCC, bool QUAD, bool CNT>
__global__ __launch_bounds__(H2T, OCC) void splat_narrow_gather_kernel(TileArgs a) {
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;
    constexpr int NG = HR / 4, NE = HR / 16;          // row groups; pixel rows a lane evaluates per record
    static_assert(HR == 16 || HR == 32, "rows per wave strip");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TSIZE = QUAD ? NQ_ZERO * 32 : MIP_TOTAL;             // (floats before the 64 zeros)
    static_assert(NQ_LINES == NQ_ZERO + 2, "two lines of zeros end the table");
    float *T = smem;
    float *Z = smem + TSIZE;                                            // 64 zeros: what a slot reads in a row its record does not cover
    int *rt_all = reinterpret_cast<int *>(smem + TSIZE + 64);           // per wave: 4 slots x HR row addresses
    typedef const __attribute__((address_space(3))) float LdsF;

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = lane >> 4, col = lane & 15;
    // (QUAD: this slot's half of every line)
    const int T_lds = (int)(unsigned)(unsigned long long)(LdsF *)T + (QUAD ? (slot & 1) * 64 : 0);
    const int Z_lds = (int)(unsigned)(unsigned long long)(LdsF *)Z + (QUAD ? (slot & 1) * 64 : 0);
    const int R = a.cam.R;
    const int n_items = a.item_base[a.n_tiles];
    if ((int)blockIdx.x * (H2T / 64) >= n_items) return;
    const int item = min((int)blockIdx.x * (H2T / 64) + wv, n_items - 1);
    const bool idle_wave = (int)blockIdx.x * (H2T / 64) + wv >= n_items;
    const int strip = a.item_tile[item];
    const int chunk = item - a.item_base[strip];
    if (QUAD) {
        for (int i = tid; i < MIPQ_TOTAL; i += H2T) {
            const int lvl = i < 1024 ? 0 : (i < 1280 ? 1 : (i < 1344 ? 2 : 3));
            const int hn = 32 >> lvl, k = i - mipq_offset(lvl), r = k / hn, c = k % hn;
            const float t = a.mips[mip_offset(lvl) + r * (2 * hn) + c];
            const int line = lvl == 0 ? 2 * r + (c >> 4) : (lvl == 1 ? NQ_L1 : (lvl == 2 ? NQ_L2 : NQ_L3)) + r;
            T[line * 32 + (c & 15)] = t; T[line * 32 + 16 + (c & 15)] = t;
        }
    } else {
        for (int i = tid; i < MIP_TOTAL; i += H2T) T[i] = a.mips[i];
    }
    if (tid < 64) Z[tid] = 0.0f;
    int *rt = rt_all + wv * (4 * HR) + slot * HR;
    const int *rt_quad = rt + (col & 3);
    const int sx = (strip % a.tiles_x) * NSW, sy = (strip / a.tiles_x) * HR;
    const float pxc = (sx + col < R) ? (float)(sx + col) + 0.5f : __builtin_inff();
    float pyc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) pyc[e] = (sy + col + 16 * e < R) ? (float)(sy + col + 16 * e) + 0.5f : __builtin_inff();

    constexpr int FOLD_EVERY = TSP_FOLD_EVERY;
    float acc[HR][NACC];
#pragma unroll
    for (int p = 0; p < HR; ++p)
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[p][c] = 0.0f;
    unsigned long long n_frag = 0;
    __syncthreads();                                       // the only workgroup barrier
    if (idle_wave) return;

    const size_t first = (size_t)a.hband_base[strip] + (size_t)chunk * a.item_records;
    constexpr int NWN = 2 * NW;           // weight floats per record of kernel N's bins: (w0, w1) / (w0, w1, w2, -): one 8- / 16-byte element; the geometry carries 1 / P
    const float4 *geom = a.geom + first;
    const float *wts = a.w + first * NWN;
    const unsigned n_rec = (unsigned)min(a.item_records, a.hband_count[strip] - chunk * a.item_records);
    // record r0 + slot of the item (one per slot).  Unconditional loads (a slot past the end re-reads the last record and is
    // emptied where it is used): under a branch the compiler waits for this prefetch right after issuing it
    auto fetch = [&](unsigned r0, float4 &g, float &gw0, float &gw1, float &gw2) {
        const unsigned ri = min(r0 + slot, n_rec - 1u);
        // (32-bit byte offsets from the item's first record -- an item holds <= 8192 records: one scalar base + one vector offset per load)
        g = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(geom) + (ri << 4));
        const char *wp = reinterpret_cast<const char *>(wts) + ri * (NWN * 4u);
        if (NW == 2) { const float4 q = *reinterpret_cast<const float4 *>(wp); gw0 = q.x; gw1 = q.y; gw2 = q.z; }
        else if (NACC >= 2) { const float2 q = *reinterpret_cast<const float2 *>(wp); gw0 = q.x; gw1 = q.y; gw2 = 0.0f; }
        else { gw0 = *reinterpret_cast<const float *>(wp); gw1 = gw2 = 0.0f; }
    };
    const int n_pass = a.cnt->mid_odd_weights ? 4 : 1;
    float4 g_next; float gw0_next, gw1_next, gw2_next;
    fetch(0, g_next, gw0_next, gw1_next, gw2_next);
    // (the first records land before the loop: with loads of the preheader still in flight at the loop header the compiler's
    // counter model gives up and waits for every prefetch right after issuing it -- vmcnt(0) at the top of each step)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    unsigned r0 = 0;                      // first record of the next step
    do {
    int since_fold = 0;
    for (; r0 < n_rec && since_fold < FOLD_EVERY; r0 += 4, ++since_fold) {
        float4 g_all = g_next;
        const float gw0_all = gw0_next, gw1_all = gw1_next, gw2_all = gw2_next;
        if (r0 + slot >= n_rec) g_all.z = 0.0f;                     // (a slot past the end of the item: covers nothing)
        fetch(r0 + 4, g_next, gw0_next, gw1_next, gw2_next);
        // weights that are not finite (the fill pass raises the flag when the list holds any): every step is drawn slot by slot --
        // then "0 x weight" only occurs inside the slot's own rows, which the wave-uniform row test skips when uncovered
        for (int pass = 0; pass < n_pass; ++pass) {
            float4 g = g_all;
            if (n_pass == 4 && slot != pass) g.z = 0.0f;            // (an empty slot: covers nothing, weight 0)
            const float half = 0.5f * g.z;
            const float invP = g.w;                                 // (1.0f / P, formed by the fill pass)
            // mip level (tsp_math.h level_for, branch-free: every lane has its own record) and that level's LUT geometry
            const int lvl = (g.z > P_L0 ? 0 : 1) + (g.z > P_L1 ? 0 : 1) + (g.z > P_L2 ? 0 : 1);
            const int n = 64 >> lvl;
            const float nf = (float)n;
            // LDS address of LUT row ty of this level: whole pyramid -- first float of the level 0, 4096, 5120, 5376, rows of n floats;
            // quadrants (the interleaved lines above) -- first line of the level, one line per row (level 0: two)
            const int tshift = QUAD ? (lvl == 0 ? 8 : 7) : 8 - lvl;
            const int tbase = QUAD ? T_lds + (lvl == 0 ? 0 : (lvl == 1 ? NQ_L1 : (lvl == 2 ? NQ_L2 : NQ_L3))) * 128
                                   : T_lds + ((lvl > 0 ? 4096 : 0) + (lvl > 1 ? 1024 : 0) + (lvl > 2 ? 256 : 0)) * 4;
            auto row_addr = [&](int ty) -> int { return tbase + (ty << tshift); };
            float wq[3];
            wq[0] = gw0_all;
            wq[1] = (NACC >= 2) ? ((MODE == TSP_MODE_RGB) ? gw1_all : gw0_all * gw1_all) : 0.0f;
            wq[2] = (NACC >= 3) ? gw2_all : 0.0f;
            // ---- rows: lane (s, c) evaluates pixel rows c (and c + 16) for slot s's record (canonical nearest-texel rule, tsp_math.h) ----
            unsigned covmask = 0, ownmask = 0;
            asm volatile("" ::: "memory");          // (in-order LDS: the previous step's table reads are done)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float d = pyc[e] - g.y;
                int ty = floor_clamp_v(((d + half) * invP) * nf, n - 1);
                if (QUAD) ty = min(ty, n - 1 - ty);
                const bool cov = __builtin_fabsf(d) < half;
                rt[col + 16 * e] = cov ? row_addr(ty) : Z_lds;
                const unsigned long long b = __ballot(cov);
                covmask |= (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull) << (16 * e);       // the union of the four slots' rows
                if (CNT) ownmask |= (unsigned)((b >> (16 * slot)) & 0xffffull) << (16 * e);
            }
            asm volatile("" ::: "memory");
            if (covmask == 0) continue;
            // ---- this lane's column: texel column (byte offset in a LUT row) and weights (+0 where the column is not covered) ----
            int tx4;
            float wl[NACC];
            {
                const float d = pxc - g.x;
                const bool covered = __builtin_fabsf(d) < half;
                int tx = floor_clamp_v(((d + half) * invP) * nf, n - 1);
                if (QUAD) tx = min(tx, n - 1 - tx);
                tx4 = tx * 4;
                if (QUAD) tx4 += (tx & 16) << 2;            // (level 0: floats 16-31 of a row sit in the next line)
#pragma unroll
                for (int c = 0; c < NACC; ++c) wl[c] = covered ? wq[c] : 0.0f;
                if (CNT) n_frag += covered ? (unsigned long long)__popc(ownmask) : 0ull;
            }
            // ---- row walk over the union of the slots' rows: a group's row addresses sit in the quads (lane (s, c): row 4 k + (c & 3) of slot s) ----
            int roq[2];
            roq[0] = rt_quad[0];
#define TSP_N_ROW(K, T_)                                                                                        \
            if (__builtin_expect((covmask >> (4 * (K) + (T_))) & 1, 1)) {      /* (likely: the FMA stays in line) */ \
                _Pragma("unroll") for (int c = 0; c < NACC; ++c) fmac_plain(acc[4 * (K) + (T_)][c], kv[T_], wl[c]); \
            }
#define TSP_N_GROUP(K)                                                                                          \
            if constexpr ((K) < NG) {                                                                           \
                if constexpr ((K) + 1 < NG) roq[((K) + 1) & 1] = rt_quad[4 * ((K) + 1)];                         \
                if (((covmask >> (4 * (K))) & 15u) != 0u) {                                                     \
                    asm volatile("" : "+v"(roq[(K) & 1]));                                                      \
                    int ad[4]; float kv[4];                                                                     \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(0) : "=v"(ad[0]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(1) : "=v"(ad[1]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(2) : "=v"(ad[2]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    asm volatile("v_add_u32_dpp %0, %1, %2 " TSP_DPP_QUAD(3) : "=v"(ad[3]) : "v"(roq[(K) & 1]), "v"(tx4)); \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t) kv[t] = *reinterpret_cast<LdsF *>(ad[t]);    \
                    TSP_N_ROW(K, 0) TSP_N_ROW(K, 1) TSP_N_ROW(K, 2) TSP_N_ROW(K, 3)                               \
                }                                                                                               \
            }
            TSP_N_GROUP(0) TSP_N_GROUP(1) TSP_N_GROUP(2) TSP_N_GROUP(3)
            TSP_N_GROUP(4) TSP_N_GROUP(5) TSP_N_GROUP(6) TSP_N_GROUP(7)
#undef TSP_N_GROUP
#undef TSP_N_ROW
        }
    }
    // ---- the four slots' partial strips summed across the DPP rows, then into the render target: lane (s, c) adds rows s, s + 4, ... ----
    {
        int Rl = R;
        asm volatile("" : "+s"(Rl));
        double *img = a.img + ((size_t)sy * Rl + (sx + col)) * C;
        asm volatile("" : "+v"(img));
#pragma unroll
        for (int ty = 0; ty < HR; ++ty) {
            const int gx = sx + col, gy = sy + ty;
#pragma unroll
            for (int c = 0; c < NACC; ++c) {
                float v = acc[ty][c];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                acc[ty][c] = 0.0f;
                if ((ty & 3) == slot && gx < Rl && gy < Rl && v != 0.0f) gatomic_add(img + ((size_t)ty * Rl) * C + c, v);
            }
        }
    }
    } while (r0 < n_rec);
    if (CNT) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[1], n_frag); }
#ifdef TSP_N_DEBUG       // analysis build: records of the bins (= (record, strip) pairs) instead of the unused fourth fragment class
        if (lane == 0) atomicAdd(&a.cnt->n_frag_class[3], (unsigned long long)n_rec);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// band bins of the huge records
// ---------------------------------------------------------------------------------------------
// Every tile of kernel H2 used to scan the WHOLE huge list (at 1e9 particles: 128 tiles x 85 MB through eight non-coherent
// L2s = 12.7 GB of fabric reads per launch, and 1/12 of the kernel's instructions spent on records that cannot reach the tile).
// One pass copies every record into the bin of each 64-row image band its square reaches (a square of P pixels reaches
// P / 64 + 1 or + 2 of them; one pixel of margin per side so that float rounding can only add a band, never drop one -- kernel
// H2 repeats the exact test per strip).  Bins are fixed regions of n_huge records each, so no sizes need to be known first;
// a workgroup reserves its slots per band with ONE global atomic (counts formed in LDS): the 12-ns same-address atomics that
// made per-record binning cost 3.3 ms in round 2 are ~n_bands per 1024 records here.
template <int NW>
__global__ __launch_bounds__(256) void huge_band_fill_kernel(const float4 *__restrict__ geom, const float *__restrict__ w, long long n,
                                                             int R, int n_bands, float4 *__restrict__ out_geom, float *__restrict__ out_w,
                                                             long long stride, int *__restrict__ band_count, const long long *__restrict__ band_base) {
    constexpr int PER = 4;                 // records per thread
    extern __shared__ int s_band[];        // [n_bands] counts, then [n_bands] bases
    int *s_cnt = s_band, *s_base = s_band + n_bands;
    for (int b = threadIdx.x; b < n_bands; b += 256) s_cnt[b] = 0;
    __syncthreads();
    const long long first = ((long long)blockIdx.x * 256 + threadIdx.x) * PER;
    float4 g[PER];
    int b0[PER], b1[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        b0[k] = 1; b1[k] = 0;
        if (first + k < n) {
            g[k] = geom[first + k];
            // (margin: one pixel plus two ulps of the coordinate, so that it still covers the rounding of g.y -+ half beyond 2^23 px)
            const float half = 0.5f * g[k].z, mg = 1.0f + 2.4e-7f * (__builtin_fabsf(g[k].y) + half), lo = g[k].y - half - mg, hi = g[k].y + half + mg;
            // (non-finite or off-image squares: no band; kernel S emits only records that cover a pixel)
            if (hi >= 0.0f && lo < (float)R && lo == lo && hi == hi) {
                b0[k] = max(0, (int)__builtin_floorf(fmaxf(lo, 0.0f) * (1.0f / HBAND_H)));
                b1[k] = min(n_bands - 1, (int)__builtin_floorf(fminf(hi, (float)R) * (1.0f / HBAND_H)));
            }
            for (int b = b0[k]; b <= b1[k]; ++b) atomicAdd(&s_cnt[b], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_bands; b += 256) {
        const int c = s_cnt[b];
        s_base[b] = c ? atomicAdd(&band_count[b], c) : 0;
        s_cnt[b] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (first + k >= n || b0[k] > b1[k]) continue;
        float w0 = w[(first + k) * NW], w1 = (NW == 2) ? w[(first + k) * NW + 1] : 0.0f;
        for (int b = b0[k]; b <= b1[k]; ++b) {
            const long long slot = (band_base ? band_base[b] : (long long)b * stride) + s_base[b] + atomicAdd(&s_cnt[b], 1);
            out_geom[slot] = g[k];
            out_w[slot * NW] = w0;
            if (NW == 2) out_w[slot * NW + 1] = w1;
        }
    }
}

// bins the huge list when that pays and fits the memory budget; sets ta.hband_* (or leaves them null)
template <int NW>
static int bin_huge_records(tsp_context *ctx, TileArgs &ta, const float4 *huge_geom, const float *huge_w, long long n_huge) {
    ta.hband_count = nullptr; ta.hband_stride = 0; ta.hband_base = nullptr;
    Workspace &ws = ctx->ws;
    const int n_bands = (ctx->R + HBAND_H - 1) / HBAND_H;
    const size_t rec_bytes = sizeof(float4) + NW * sizeof(float);
    // (one band: nothing to gain; a short list is scanned in microseconds; a huge image with a long list would not fit)
    if (n_bands < 2 || n_huge < 4096 || (long long)n_bands * n_huge * (long long)rec_bytes > ctx->huge_band_budget) return TSP_OK;
    if (ws.hband_stride < n_huge || ws.hband_bands < n_bands) {
        if (ws.hband_geom) TSP_HIP(hipFree(ws.hband_geom));
        if (ws.hband_w) TSP_HIP(hipFree(ws.hband_w));
        if (ws.hband_count) TSP_HIP(hipFree(ws.hband_count));
        ws.hband_geom = ws.hband_w = nullptr; ws.hband_count = nullptr;
        ws.hband_stride = n_huge + n_huge / 8 + 1024;
        ws.hband_bands = n_bands;
        const size_t slots = (size_t)n_bands * (size_t)ws.hband_stride;
        if ((long long)slots * (long long)(sizeof(float4) + 2 * sizeof(float)) > ctx->huge_band_budget + (ctx->huge_band_budget >> 2)) ws.hband_stride = n_huge;
        const size_t slots2 = (size_t)n_bands * (size_t)ws.hband_stride;
        TSP_HIP(hipMalloc(&ws.hband_geom, slots2 * sizeof(float4)));
        TSP_HIP(hipMalloc(&ws.hband_w, slots2 * 2 * sizeof(float)));
        TSP_HIP(hipMalloc((void **)&ws.hband_count, 256 * sizeof(int)));
    }
    hipStream_t st = ctx->stream;
    TSP_HIP(hipMemsetAsync(ws.hband_count, 0, 256 * sizeof(int), st));
    const unsigned grid = (unsigned)((n_huge + 1023) / 1024);
    hipLaunchKernelGGL((huge_band_fill_kernel<NW>), dim3(grid), dim3(256), 2 * n_bands * sizeof(int), st, huge_geom, huge_w, n_huge, ctx->R, n_bands,
                       (float4 *)ws.hband_geom, (float *)ws.hband_w, (long long)ws.hband_stride, ws.hband_count, (const long long *)nullptr);
    TSP_HIP(hipGetLastError());
    ta.geom = (const float4 *)ws.hband_geom; ta.w = (const float *)ws.hband_w;
    ta.hband_count = ws.hband_count; ta.hband_stride = ws.hband_stride;
    return TSP_OK;
}

// ---- strip bins of the mid records (kernel G; "tile" in the names below = one 64 x HR strip) --------------------------
// Every mid record is copied into the bin of each 64 x HR-pixel strip its square reaches (a footprint below 64 px, one pixel of margin
// per side: <= 3 strips across, <= 4 or 6 down; ~2.9 on average), in three passes -- count, prefix, fill -- so that the bins are exact in
// size; a WAVE of kernel G then draws one WORK ITEM: item_records consecutive records of one strip's bin.  Items are equal in size and
// nearly equal in work (every record of a bin reaches the strip), a strip gets as many as its bin needs, and the launch is greedy over
// ~7e4 of them: binned by image band only, with the same number of workgroups for every tile, the workgroups of the densest
// tiles ran ten times longer than the rest and set the kernel's time (1e9 particles: 19.9 / 15.4 / 13.3 ms at 128 / 256 / 512
// workgroups per tile), and every workgroup scanned the whole band's records for the few that reach its tile; bins per 128 x 64 tile
// with one item per workgroup: 12.35 ms (the four strips of a tile differ in work); per strip with one item per wave: 10.0.

struct TileSpan { int x0, x1, y0, y1; };
// what the binning passes of one kernel-G launch share: strip shape, the footprint widths it takes, and (exact) whether a record
// goes only into the bins of the strips its square AND the kernel's disc reach (kernel N draws every record of a bin unasked)
struct BinArgs {
    int R, tw, th, tiles_x, tiles_y;
    float pmin, pmax;          // footprints with pmin <= P < pmax
    float disc_k2;             // as TileArgs::disc_k2 (0: the square alone decides)
    int exact;
    int narrow;                // kernel N's record layout: geometry (pcx, pcy, P, 1 / P), weights (w0, w1[, w2]) -- see tile_fill_kernel
    // The passes keep their per-strip counters in LDS, a WINDOW of win_rows rows of strips at a time (blockIdx.y = window): every
    // workgroup reads its records once per window and handles the pairs whose strip lies in it.  One window while the image has
    // <= 4096 strips; 16384 strips (16 x 16-px strips at 2048^2) in one window left one workgroup per CU (128 KB of LDS)
    int win_rows;
};
__device__ __forceinline__ TileSpan tile_span(const float4 g, const BinArgs &b) {
    TileSpan s; s.x0 = s.y0 = 1; s.x1 = s.y1 = 0;
    if (!(g.z >= b.pmin && g.z < b.pmax)) return s;
    // (margin: one pixel plus two ulps of the coordinate -- it covers the rounding of g -+ half at any magnitude)
    const float half = 0.5f * g.z, mx = 1.0f + 2.4e-7f * (__builtin_fabsf(g.x) + half), my = 1.0f + 2.4e-7f * (__builtin_fabsf(g.y) + half);
    const float xl = g.x - half - mx, xh = g.x + half + mx, yl = g.y - half - my, yh = g.y + half + my;
    // (non-finite or off-image squares: no tile; kernel S emits only records that cover a pixel)
    if (xh >= 0.0f && xl < (float)b.R && yh >= 0.0f && yl < (float)b.R && xl == xl && xh == xh && yl == yl && yh == yh) {
        s.x0 = max(0, (int)__builtin_floorf(fmaxf(xl, 0.0f) / (float)b.tw));
        s.x1 = min(b.tiles_x - 1, (int)__builtin_floorf(fminf(xh, (float)b.R) / (float)b.tw));
        s.y0 = max(0, (int)__builtin_floorf(fmaxf(yl, 0.0f) / (float)b.th));
        s.y1 = min(b.tiles_y - 1, (int)__builtin_floorf(fminf(yh, (float)b.R) / (float)b.th));
    }
    return s;
}
// the test kernels G and H2 make per (record, strip) pair: the footprint square and the disc inscribed in it reach the strip
__device__ __forceinline__ bool strip_hit(const float4 g, int tx, int ty, const BinArgs &b) {
    if (!b.exact) return true;
    const float sx0 = (float)(tx * b.tw), sx1 = (float)(tx * b.tw + b.tw), sy0 = (float)(ty * b.th), sy1 = (float)(ty * b.th + b.th);
    const float half = 0.5f * g.z;
    const float sdx = fmaxf(fmaxf(sx0 - g.x, g.x - sx1), 0.0f), sdy = fmaxf(fmaxf(sy0 - g.y, g.y - sy1), 0.0f);
    return g.z > 0.0f && sdx < half && sdy < half && !(b.disc_k2 > 0.0f && sdx * sdx + sdy * sdy >= b.disc_k2 * g.z * g.z);
}
// pass 1: records per tile (counted in LDS first when the image has few enough tiles: one global atomic per workgroup and tile)
__global__ __launch_bounds__(256) void tile_count_kernel(const float4 *__restrict__ geom, long long n, BinArgs b, int *__restrict__ tile_count) {
    constexpr int PER = TSP_BIN_PER;
    extern __shared__ int s_tile[];
    const int tiles_x = b.tiles_x;
    const int row0 = (int)blockIdx.y * b.win_rows, row1 = min(b.tiles_y, row0 + b.win_rows) - 1, t0 = row0 * tiles_x, n_win = (row1 - row0 + 1) * tiles_x;
    const bool lds = n_win <= G_LDS_TILES;
    if (lds) {
        for (int t = threadIdx.x; t < n_win; t += 256) s_tile[t] = 0;
        __syncthreads();
    }
    int *cnt = lds ? s_tile : tile_count + t0;
    const long long first = ((long long)blockIdx.x * 256 + threadIdx.x) * PER;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (first + k >= n) continue;
        const float4 g = geom[first + k];
        const TileSpan sp = tile_span(g, b);
        for (int ty = max(sp.y0, row0); ty <= min(sp.y1, row1); ++ty)
            for (int tx = sp.x0; tx <= sp.x1; ++tx)
                if (strip_hit(g, tx, ty, b)) atomicAdd(&cnt[ty * tiles_x + tx - t0], 1);
    }
    if (lds) {
        __syncthreads();
        for (int t = threadIdx.x; t < n_win; t += 256)
            if (s_tile[t]) atomicAdd(&tile_count[t0 + t], s_tile[t]);
    }
}
// between the passes (one workgroup): first record of every bin, first work item of every tile, the item -> tile table
__global__ __launch_bounds__(1024) void tile_prefix_kernel(const int *__restrict__ tile_count, int n_tiles, long long *__restrict__ tile_base,
                                                           int *__restrict__ item_base, int *__restrict__ item_tile, int item_capacity, int item_records) {
    __shared__ long long s_rec[1024];
    __shared__ int s_item[1024];
    __shared__ long long s_carry_rec;
    __shared__ int s_carry_item;
    const int tid = threadIdx.x;
    if (tid == 0) { s_carry_rec = 0; s_carry_item = 0; }
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int t = base + tid;
        const int c = t < n_tiles ? tile_count[t] : 0;
        const int it = (c + item_records - 1) / item_records;
        s_rec[tid] = c; s_item[tid] = it;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const long long vr = tid >= o ? s_rec[tid - o] : 0;
            const int vi = tid >= o ? s_item[tid - o] : 0;
            __syncthreads();
            s_rec[tid] += vr; s_item[tid] += vi;
            __syncthreads();
        }
        const long long rec0 = s_carry_rec + s_rec[tid] - c;
        const int item0 = s_carry_item + s_item[tid] - it;
        if (t < n_tiles) {
            tile_base[t] = rec0; item_base[t] = item0;
            for (int i = 0; item_tile && i < it; ++i)
                if ( item0 + i < item_capacity) item_tile[item0 + i] = t;
        }
        __syncthreads();
        if (tid == 1023) { s_carry_rec += s_rec[1023]; s_carry_item += s_item[1023]; }
        __syncthreads();
    }
    if (tid == 0) { tile_base[n_tiles] = s_carry_rec; item_base[n_tiles] = s_carry_item; }
}
// pass 3: the records into their bins (with LDS counters a workgroup reserves its slots per tile with one global atomic)
template <int NW>
__global__ __launch_bounds__(256) void tile_fill_kernel(const float4 *__restrict__ geom, const float *__restrict__ w, long long n, BinArgs b,
                                                        float4 *__restrict__ out_geom, float *__restrict__ out_w,
                                                        const long long *__restrict__ tile_base, int *__restrict__ tile_cursor,
                                                        unsigned long long *__restrict__ odd_flag) {
    constexpr int PER = TSP_BIN_PER;
    extern __shared__ int s_tile[];        // [strips of the window] counts, then as many bases
    const int tiles_x = b.tiles_x;
    const int row0 = (int)blockIdx.y * b.win_rows, row1 = min(b.tiles_y, row0 + b.win_rows) - 1, t0 = row0 * tiles_x, n_win = (row1 - row0 + 1) * tiles_x;
    const bool lds = n_win <= G_LDS_TILES;
    int *s_cnt = s_tile, *s_base = s_tile + n_win;
    if (lds) {
        for (int t = threadIdx.x; t < n_win; t += 256) s_cnt[t] = 0;
        __syncthreads();
    }
    const long long first = ((long long)blockIdx.x * 256 + threadIdx.x) * PER;
    float4 g[PER];
    TileSpan sp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        sp[k].x0 = sp[k].y0 = 1; sp[k].x1 = sp[k].y1 = 0;
        if (first + k < n) {
            g[k] = geom[first + k];
            sp[k] = tile_span(g[k], b);
            sp[k].y0 = max(sp[k].y0, row0); sp[k].y1 = min(sp[k].y1, row1);      // (the rows of this window)
            if (lds)
                for (int ty = sp[k].y0; ty <= sp[k].y1; ++ty)
                    for (int tx = sp[k].x0; tx <= sp[k].x1; ++tx)
                        if (strip_hit(g[k], tx, ty, b)) atomicAdd(&s_cnt[ty * tiles_x + tx - t0], 1);
        }
    }
    if (lds) {
        __syncthreads();
        for (int t = threadIdx.x; t < n_win; t += 256) {
            const int c = s_cnt[t];
            s_base[t] = c ? atomicAdd(&tile_cursor[t0 + t], c) : 0;
            s_cnt[t] = 0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (first + k >= n || sp[k].y0 > sp[k].y1 || sp[k].x0 > sp[k].x1) continue;
        const float w0 = w[(first + k) * NW], w1 = (NW == 2) ? w[(first + k) * NW + 1] : 0.0f;
        // kernel N's records: (pcx, pcy, P, 1 / P) + every weight in the weight array -- the IEEE division once per record copy
        // here instead of once per record and LANE there (kernel N holds a record's parameters in all 16 lanes of its slot)
        const float4 gn = make_float4(g[k].x, g[k].y, g[k].z, 1.0f / g[k].z);
        // (kernel N draws four records per step: a weight that is not finite makes it draw every step slot by slot -- see there)
        if (b.narrow && odd_flag && !(__builtin_fabsf(g[k].w) < __builtin_inff() && __builtin_fabsf(w0) < __builtin_inff() && __builtin_fabsf(w1) < __builtin_inff()))
            *odd_flag = 1ull;
        for (int ty = sp[k].y0; ty <= sp[k].y1; ++ty)
            for (int tx = sp[k].x0; tx <= sp[k].x1; ++tx) {
                if (!strip_hit(g[k], tx, ty, b)) continue;
                const int t = ty * tiles_x + tx;
                const long long slot = tile_base[t] + (lds ? s_base[t - t0] + atomicAdd(&s_cnt[t - t0], 1) : atomicAdd(&tile_cursor[t], 1));
                if (b.narrow) {
                    out_geom[slot] = gn;
                    // (the weights as ONE store)
                    if (NW == 2) reinterpret_cast<float4 *>(out_w)[slot] = make_float4(g[k].w, w0, w1, 0.0f);
                    else reinterpret_cast<float2 *>(out_w)[slot] = make_float2(g[k].w, w0);
                } else {
                    out_geom[slot] = g[k];
                    out_w[slot * NW] = w0;
                    if (NW == 2) out_w[slot * NW + 1] = w1;
                }
            }
    }
}

// bins the mid list by strip and builds the work items; sets ta.{geom, w, hband_count (records per strip), hband_base, item_*}
template <int NW>
static int bin_mid_records(tsp_context *ctx, TileArgs &ta, const float4 *mid_geom, const float *mid_w, long long n_mid, int tw, int th, float pmin,
                           float pmax, bool exact, int *n_items_out, hipStream_t st) {
    Workspace &ws = ctx->ws;
    const int tiles_x = (ctx->R + tw - 1) / tw, tiles_y = (ctx->R + th - 1) / th, n_tiles = tiles_x * tiles_y;
    BinArgs ba;
    ba.R = ctx->R; ba.tw = tw; ba.th = th; ba.tiles_x = tiles_x; ba.tiles_y = tiles_y; ba.pmin = pmin; ba.pmax = pmax;
    ba.disc_k2 = ta.disc_k2; ba.exact = exact ? 1 : 0; ba.narrow = exact ? 1 : 0;
    ba.win_rows = std::max(1, G_WIN_TILES / tiles_x);
    const int n_win = (tiles_y + ba.win_rows - 1) / ba.win_rows, win_tiles = std::min(tiles_y, ba.win_rows) * tiles_x;
    if (ws.mtile_capacity < n_tiles) {
        void *olds[] = {ws.mband_count, ws.mband_base, ws.mitem_base};
        for (void *q : olds)
            if (q) TSP_HIP(hipFree(q));
        ws.mband_count = ws.mitem_base = nullptr; ws.mband_base = nullptr;
        ws.mtile_capacity = n_tiles;
        TSP_HIP(hipMalloc((void **)&ws.mband_count, 2 * (size_t)ws.mtile_capacity * sizeof(int)));               // counts | fill cursors
        TSP_HIP(hipMalloc((void **)&ws.mband_base, ((size_t)ws.mtile_capacity + 1) * sizeof(long long)));
        TSP_HIP(hipMalloc((void **)&ws.mitem_base, ((size_t)ws.mtile_capacity + 1) * sizeof(int)));
    }
    // records per item: short items balance a short list over the device, long ones amortise the LUT load and the final flush
    int item_records = ctx->mid_item_records;
    if (item_records <= 0) {
        const double want = ctx->mid_item_scale * std::sqrt((double)n_mid);
        item_records = 64;
        while (item_records < GCHUNK_MAX && (double)item_records * 1.41 < want) item_records *= 2;
    }
    TSP_HIP(hipMemsetAsync(ws.mband_count, 0, 2 * (size_t)ws.mtile_capacity * sizeof(int), st));
    const bool lds = win_tiles <= G_LDS_TILES;
    if (!(ctx->kernel_attr_done & (1u << (8 + NW)))) {      // (more than 64 KB of dynamic LDS needs the attribute)
        TSP_HIP(hipFuncSetAttribute((const void *)tile_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_TILES * (int)sizeof(int)));
        TSP_HIP(hipFuncSetAttribute((const void *)tile_fill_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_LDS_TILES * (int)sizeof(int)));
        ctx->kernel_attr_done |= 1u << (8 + NW);
    }
    const unsigned grid = (unsigned)((n_mid + 256 * TSP_BIN_PER - 1) / (256 * TSP_BIN_PER));
    hipLaunchKernelGGL(tile_count_kernel, dim3(grid, n_win), dim3(256), lds ? win_tiles * sizeof(int) : 0, st, mid_geom, n_mid, ba, ws.mband_count);
    // the sizes of the bins are known on the device only: the prefix pass runs once without the item table to size it, the
    // host reads the two totals (one small copy; the pipeline already synchronises once per frame for the record counts) and grows
    // the bins when needed, then the pass runs again and writes the table
    hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(1024), 0, st, (const int *)ws.mband_count, n_tiles, ws.mband_base, ws.mitem_base, (int *)nullptr, 0, item_records);
    long long total_records = 0; int total_items = 0;
    TSP_HIP(hipMemcpyAsync(&total_records, ws.mband_base + n_tiles, sizeof(long long), hipMemcpyDeviceToHost, st));
    TSP_HIP(hipMemcpyAsync(&total_items, ws.mitem_base + n_tiles, sizeof(int), hipMemcpyDeviceToHost, st));
    TSP_HIP(hipStreamSynchronize(st));
    if (ws.mband_capacity < total_records || ws.mitem_capacity < total_items) {
        void *olds[] = {ws.mband_geom, ws.mband_w, ws.mitem_tile};
        for (void *q : olds)
            if (q) TSP_HIP(hipFree(q));
        ws.mband_geom = ws.mband_w = nullptr; ws.mitem_tile = nullptr;
        ws.mband_capacity = std::max<int64_t>(ws.mband_capacity, total_records + total_records / 4 + 1024);
        ws.mitem_capacity = std::max<int64_t>(ws.mitem_capacity, (int64_t)total_items + total_items / 4 + 1024);
        TSP_HIP(hipMalloc(&ws.mband_geom, (size_t)ws.mband_capacity * sizeof(float4)));
        TSP_HIP(hipMalloc(&ws.mband_w, (size_t)ws.mband_capacity * 4 * sizeof(float)));      // (kernel N keeps a float2 / float4 of weights per record here)
        TSP_HIP(hipMalloc((void **)&ws.mitem_tile, (size_t)ws.mitem_capacity * sizeof(int)));
    }
    hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(1024), 0, st, (const int *)ws.mband_count, n_tiles, ws.mband_base, ws.mitem_base, ws.mitem_tile, total_items, item_records);
    hipLaunchKernelGGL((tile_fill_kernel<NW>), dim3(grid, n_win), dim3(256), lds ? 2 * win_tiles * sizeof(int) : 0, st, mid_geom, mid_w, n_mid, ba,
                       (float4 *)ws.mband_geom, (float *)ws.mband_w, (const long long *)ws.mband_base, ws.mband_count + ws.mtile_capacity,
                       &ctx->counters->mid_odd_weights);
    TSP_HIP(hipGetLastError());
    ta.geom = (const float4 *)ws.mband_geom; ta.w = (const float *)ws.mband_w;
    ta.hband_count = ws.mband_count; ta.hband_stride = 0; ta.hband_base = ws.mband_base;
    ta.item_tile = ws.mitem_tile; ta.item_base = ws.mitem_base; ta.n_tiles = n_tiles; ta.item_records = item_records;
    ta.tiles_x = tiles_x;
    *n_items_out = total_items;
    return TSP_OK;
}

template <int MODE, int NACC, int HR, int OCC>
static int launch_mid_gather_kernel(tsp_context *ctx, TileArgs ta, const float4 *mid_geom, const float *mid_w, long long n_mid, float pmin, hipStream_t st) {
    const bool quad = ctx->lut_mirror_symmetric && !ctx->debug_gather_full_lut;
    const size_t smem = (size_t)(quad ? MIPQ_TOTAL : MIP_TOTAL) * sizeof(float) + (H2T / 64) * 64 * sizeof(int);
    int rc, n_items = 0;
    ta.n_records = n_mid;
    if ((rc = bin_mid_records<(MODE == TSP_MODE_RGB) ? 2 : 1>(ctx, ta, mid_geom, mid_w, n_mid, 64, HR, pmin, __builtin_inff(), false, &n_items, st))) return rc;
    if (n_items == 0) return TSP_OK;
    const dim3 grid((n_items + H2T / 64 - 1) / (H2T / 64));
    if (quad) {
        if (ta.count_frag) hipLaunchKernelGGL((splat_mid_gather_kernel<MODE, NACC, HR, OCC, true, true>), grid, dim3(H2T), smem, st, ta);
        else hipLaunchKernelGGL((splat_mid_gather_kernel<MODE, NACC, HR, OCC, true, false>), grid, dim3(H2T), smem, st, ta);
    } else {
        if (ta.count_frag) hipLaunchKernelGGL((splat_mid_gather_kernel<MODE, NACC, HR, OCC, false, true>), grid, dim3(H2T), smem, st, ta);
        else hipLaunchKernelGGL((splat_mid_gather_kernel<MODE, NACC, HR, OCC, false, false>), grid, dim3(H2T), smem, st, ta);
    }
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

// kernel N for the records below `pmax` px (its own bins: 16-column strips, only the records that reach a strip)
template <int MODE, int NACC, int HR, int OCC>
static int launch_narrow_gather_kernel(tsp_context *ctx, TileArgs ta, const float4 *mid_geom, const float *mid_w, long long n_mid, float pmax, hipStream_t st) {
    const bool quad = ctx->lut_mirror_symmetric && !ctx->debug_gather_full_lut;
    const size_t smem = (size_t)(quad ? NQ_LINES * 32 : MIP_TOTAL + 64) * sizeof(float) + (H2T / 64) * 4 * HR * sizeof(int);
    int rc, n_items = 0;
    ta.n_records = n_mid;
    if ((rc = bin_mid_records<(MODE == TSP_MODE_RGB) ? 2 : 1>(ctx, ta, mid_geom, mid_w, n_mid, NSW, HR, 0.0f, pmax, true, &n_items, st))) return rc;
    if (n_items == 0) return TSP_OK;
    const dim3 grid((n_items + H2T / 64 - 1) / (H2T / 64));
    if (quad) {
        if (ta.count_frag) hipLaunchKernelGGL((splat_narrow_gather_kernel<MODE, NACC, HR, OCC, true, true>), grid, dim3(H2T), smem, st, ta);
        else hipLaunchKernelGGL((splat_narrow_gather_kernel<MODE, NACC, HR, OCC, true, false>), grid, dim3(H2T), smem, st, ta);
    } else {
        if (ta.count_frag) hipLaunchKernelGGL((splat_narrow_gather_kernel<MODE, NACC, HR, OCC, false, true>), grid, dim3(H2T), smem, st, ta);
        else hipLaunchKernelGGL((splat_narrow_gather_kernel<MODE, NACC, HR, OCC, false, false>), grid, dim3(H2T), smem, st, ta);
    }
    TSP_HIP(hipGetLastError());
    return TSP_OK;
}

template <int MODE>
static int launch_mid_gather_mode(tsp_context *ctx, TileArgs ta, bool second_channel, const float4 *mid_geom, const float *mid_w, long long n_mid, hipStream_t st) {
    TSP_REQUIRE(n_mid < (1ll << 28), TSP_EINVAL, "%lld mid footprints in one launch (kernel G indexes its work items with 32 bits; run_pipeline slices the list)", n_mid);
    // the mid list is drawn in two passes over it: footprints below mid_narrow_px by kernel N (four records per wave step on
    // 16-column strips), the rest by kernel G (one record per wave step on 64-column strips); 0 = everything by kernel G
    const float split = ctx->mid_narrow_px;
    int rc;
    if (split > 0.0f) {
        if (MODE == TSP_MODE_RGB) rc = launch_narrow_gather_kernel<MODE, 3, 16, TSP_G_OCC3>(ctx, ta, mid_geom, mid_w, n_mid, split, st);
        else if (second_channel) rc = launch_narrow_gather_kernel<MODE, 2, 16, TSP_G_OCC2>(ctx, ta, mid_geom, mid_w, n_mid, split, st);
        else rc = launch_narrow_gather_kernel<MODE, 1, TSP_G_HR1, TSP_G_OCC1>(ctx, ta, mid_geom, mid_w, n_mid, split, st);
        if (rc) return rc;
        if (split >= P_BILINEAR) return TSP_OK;        // (every mid footprint is below 64 px: nothing is left for kernel G)
    }
    if (MODE == TSP_MODE_RGB) return launch_mid_gather_kernel<MODE, 3, 16, TSP_G_OCC3>(ctx, ta, mid_geom, mid_w, n_mid, split, st);
    if (second_channel) return launch_mid_gather_kernel<MODE, 2, 16, TSP_G_OCC2>(ctx, ta, mid_geom, mid_w, n_mid, split, st);
    return launch_mid_gather_kernel<MODE, 1, TSP_G_HR1, TSP_G_OCC1>(ctx, ta, mid_geom, mid_w, n_mid, split, st);
}

int launch_mid_gather(tsp_context *ctx, TileArgs ta, int mode, bool second_channel, const float4 *mid_geom, const float *mid_w,
                      long long n_mid, hipStream_t st) {
    switch (mode) {
        case TSP_MODE_WEIGHTED: return launch_mid_gather_mode<TSP_MODE_WEIGHTED>(ctx, ta, second_channel, mid_geom, mid_w, n_mid, st);
        case TSP_MODE_DEPTH: return launch_mid_gather_mode<TSP_MODE_DEPTH>(ctx, ta, true, mid_geom, mid_w, n_mid, st);
        case TSP_MODE_RGB: return launch_mid_gather_mode<TSP_MODE_RGB>(ctx, ta, true, mid_geom, mid_w, n_mid, st);
    }
    set_error("bad mode %d", mode);
    return TSP_EINVAL;
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
        // Fewer, longer workgroups for shorter record lists: every workgroup loads the kernel image and each of its waves walks its
        // share of the list at memory latency (64 records per step), so below ~2e6 records the scan outweighs the balance that many
        // short workgroups buy.  Round 5, 1024^2, workgroups per 128x64 tile (64x32 strips): 4.2e6 records 192 / 256 / 384 -> 33.2 /
        // 33.05 / 32.8 ms; 1.2e6: 128 / 192 / 256 / 384 -> 10.29 / 10.05 / 10.37 / 11.1; 5.3e5 (one of 8 shards of the 1e9 snapshot):
        // 64 / 96 / 128 / 192 / 256 -> 5.32 / 4.90 / 4.76 / 4.85 / 5.38; 3.4e5: 64 / 128 / 256 -> 4.24 / 3.42 / 4.45
        if (n_huge < (1ll << 16)) split = std::max(32, (int)((long long)split * n_huge >> 16));      // a small render block: in proportion
        else if (HR == 32 && NACC == 1) split = n_huge >= 2000000 ? split : (n_huge >= 1000000 ? (split * 3) / 4 : split / 2);
        else if (HR == 16 && NACC == 1) split = std::max(1, split / 2);      // (64x16 strips serve < 2.5e5 records: 32 / 64 / 128 per 128x32 tile -> 2.64 / 2.18 / 2.38 ms at 1.5e5)
        else if (HR == 16 && NACC == 2 && n_huge < 1000000) split = std::max(1, split / 2);      // two channels, 3.4e5 records: 64 / 128 / 192 / 256 -> 4.49 / 5.16 / 6.49 / 8.16 ms
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
// host side: strip shape and occupancy by mode and record count
// ---------------------------------------------------------------------------------------------
template <int MODE>
static int launch_gather_mode(tsp_context *ctx, TileArgs ta, bool second_channel, const float4 *huge_geom, const float *huge_w,
                              long long n_huge) {
    hipStream_t st = ctx->stream;
    int rc = TSP_OK;
    if (n_huge > 0) {
        ta.geom = huge_geom; ta.w = huge_w; ta.n_records = n_huge;
        TSP_REQUIRE(n_huge < (1ll << 31), TSP_EINVAL, "%lld deferred footprints in one render block (the tile-gather kernel indexes them with 32 bits)", n_huge);
        if ((rc = bin_huge_records<(MODE == TSP_MODE_RGB) ? 2 : 1>(ctx, ta, huge_geom, huge_w, n_huge))) return rc;
        if (MODE == TSP_MODE_RGB) {
            // three accumulator sets: 96 VGPRs at 5 waves/SIMD (11.5 against 12.5 ms at 4 for the 64-128 px band of config 4)
            if (ctx->huge_variant == 4) rc = launch_huge2<MODE, 3, 1, 16, 4>(ctx, ta, n_huge);
            else rc = launch_huge2<MODE, 3, 1, 16, 5>(ctx, ta, n_huge);
        } else if (second_channel) {
            // 72 VGPRs at 7 waves/SIMD (24 B of scratch outside the row loop): 9.48 against 9.78 ms at 6 (80 VGPRs), 10.8 at 8 (spills)
            if (ctx->huge_variant == 4) rc = launch_huge2<MODE, 2, 1, 16, 4>(ctx, ta, n_huge);
            else rc = launch_huge2<MODE, 2, 1, 16, 7>(ctx, ta, n_huge);
        }
        else if (ctx->huge_variant == 2) rc = launch_huge2<MODE, 1, 1, 32, 6>(ctx, ta, n_huge);
        else if (ctx->huge_variant == 4) rc = launch_huge2<MODE, 1, 1, 16, 7>(ctx, ta, n_huge);
        else if (ctx->huge_variant == 5) rc = launch_huge2<MODE, 1, 1, 16, 8>(ctx, ta, n_huge);
        else if (ctx->huge_variant == 6) rc = launch_huge2<MODE, 1, 1, 32, 7>(ctx, ta, n_huge);
        // Density: 64x32 strips with the row factors fetched group by group -- half as many (footprint, strip) pairs to set up --
        // at 8 waves/SIMD (64 VGPRs): this kernel is latency-bound per wave, occupancy is what pays.  1.25e8 particles, records
        // 64-768 px: 10.20 / 9.86 / 9.59 ms at 6 / 7 / 8 waves (1e9: 28.7 / 26.9 / 26.9); 64x16 strips at 8: 10.7.  Round 5 re-measured
        // the cross-over with the workgroup count tuned per strip shape (launch_huge2): 64x32 strips win from ~2.5e5 records
        // (5.3e5: 4.76 against 5.47 ms; 3.4e5: 3.42 against 3.81; 1.5e5: 2.22 against 2.18)
        else if (ctx->huge_variant == 7 || (ctx->huge_variant == 1 && n_huge >= 250000)) rc = launch_huge2<MODE, 1, 1, 32, 8>(ctx, ta, n_huge);
        else rc = launch_huge2<MODE, 1, 1, 16, 8>(ctx, ta, n_huge);
        if (rc) return rc;
    }
    TSP_HIP(hipEventRecord(ctx->ev[10], st));
    return TSP_OK;
}

int launch_gather_kernels(tsp_context *ctx, TileArgs ta, int mode, bool second_channel, const float4 *huge_geom, const float *huge_w,
                          long long n_huge) {
    switch (mode) {
        case TSP_MODE_WEIGHTED: return launch_gather_mode<TSP_MODE_WEIGHTED>(ctx, ta, second_channel, huge_geom, huge_w, n_huge);
        case TSP_MODE_DEPTH: return launch_gather_mode<TSP_MODE_DEPTH>(ctx, ta, true, huge_geom, huge_w, n_huge);
        case TSP_MODE_RGB: return launch_gather_mode<TSP_MODE_RGB>(ctx, ta, true, huge_geom, huge_w, n_huge);
    }
    set_error("bad mode %d", mode);
    return TSP_EINVAL;
}

}  // namespace tsp
