// tsp_pipeline.hip -- the splat pipeline (default render path): streaming kernel S and the host side of a frame; gfx950.
//
// What it computes is exactly vertex_* + fragment_* + additive blend of the reference
// (src/topsy/shaders/sph.wgsl:54-91,139-165; src/topsy/sph.py:31-42) in the canonical arithmetic of
// tsp_math.h.  How it is scheduled is MI355X-specific and driven by the footprint distribution
// (DESIGN.md section 4): 80 % of particles cover < 8 px but > 90 % of all pixel updates come from the
// ~1 % of particles wider than 64 px.
//
//   kernel S  splat_stream_kernel  streams the SoA particle arrays once (coalesced, 512-particle
//             chunks; persistent workgroups take batches of consecutive chunks from a shared counter,
//             so load-time spatial order gives screen locality and the expensive chunks do not set the
//             kernel's time).  Footprints < p_small px that fit the LDS window following the chunks
//             are rasterised at once (ds_add_f64), flushed with one global atomic per touched
//             pixel.  All other footprints are not rasterised here: their projected records
//             (pcx, pcy, P, weights) are appended to the MID list or the HUGE list (P >= 64 px).
//   kernel G  (tsp_gather.hip) draws the MID list (nearest sampling): the records binned per 64-px
//             pixel strip, equal work items, one per wave, accumulators in registers.
//   kernel H2 (tsp_gather.hip) takes the footprints >= 64 px (bilinear sampling): per-wave pixel strips held in
//             registers, records scanned per wave, no atomics in the loop.
//
// All of them add into the float64 render target with device-scope atomics only at flush time.
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "tsp_pipeline.h"

namespace tsp {

// ---------------------------------------------------------------------------------------------
// block-wide helpers (256 threads = 4 waves)
// ---------------------------------------------------------------------------------------------
// Wave-wide reductions and scans on the VALU's DPP path (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15
// and row_bcast:31 across the rows): kernel S does ten of them per 512-particle chunk, and as __shfl (ds_bpermute_b32) they
// were 40 % of its LDS instructions -- on the pipe its LDS atomics need.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_or(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);     // lanes without a source get `identity`
}
template <typename Op> __device__ __forceinline__ int wave_scan_bits(int v, int identity, Op op) {
    v = op(v, dpp_or<0x111, 0xf>(identity, v));     // row_shr:1
    v = op(v, dpp_or<0x112, 0xf>(identity, v));     // row_shr:2
    v = op(v, dpp_or<0x114, 0xf>(identity, v));     // row_shr:4
    v = op(v, dpp_or<0x118, 0xf>(identity, v));     // row_shr:8
    v = op(v, dpp_or<0x142, 0xa>(identity, v));     // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_or<0x143, 0xc>(identity, v));     // row_bcast:31 into rows 2 and 3
    return v;                                       // inclusive scan; lane 63 holds the reduction
}
__device__ __forceinline__ float wave_min(float v) {
    const int r = wave_scan_bits(__float_as_int(v), __float_as_int(__builtin_inff()),
                                 [](int a, int b) { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); });
    return __int_as_float(__builtin_amdgcn_readlane(r, 63));
}
__device__ __forceinline__ float wave_max(float v) {
    const int r = wave_scan_bits(__float_as_int(v), __float_as_int(-__builtin_inff()),
                                 [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); });
    return __int_as_float(__builtin_amdgcn_readlane(r, 63));
}
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {       // both 16-bit halves at once (v_pk_min_u16)
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned wave_pk_min_u16(unsigned v) {
    const int r = wave_scan_bits((int)v, -1, [](int a, int b) { return (int)pk_min_u16((unsigned)a, (unsigned)b); });
    return (unsigned)__builtin_amdgcn_readlane(r, 63);
}
__device__ __forceinline__ unsigned wave_pk_max_u16(unsigned v) {
    const int r = wave_scan_bits((int)v, 0, [](int a, int b) { return (int)pk_max_u16((unsigned)a, (unsigned)b); });
    return (unsigned)__builtin_amdgcn_readlane(r, 63);
}
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/) {
    return wave_scan_bits(v, 0, [](int a, int b) { return a + b; });
}

// Sum of v over the lanes of the wave (every lane must take part; inactive contributions are passed as 0)
__device__ __forceinline__ float wave_sum(float v) {
    const int r = wave_scan_bits(__float_as_int(v), 0, [](int a, int b) { return __float_as_int(__int_as_float(a) + __int_as_float(b)); });
    return __int_as_float(__builtin_amdgcn_readlane(r, 63));
}

struct StreamArgs {
    Particles p;
    const int64_t *ranges;     // starts[n] | lens[n] | chunk_prefix[n+1]
    int n_ranges;
    int n_chunks;
    int chunks_per_block;      // chunks of a workgroup's first (static) batch = the largest batch it takes from the shared counter
    int dynamic;               // 1: workgroups are persistent and take further batches from Counters::next_chunk (see the kernel)
    Camera cam;
    const float *mips;
    double *img;
    float4 *mid_geom;  float *mid_w;   int64_t mid_capacity;
    float4 *huge_geom; float *huge_w;  int64_t huge_capacity;
    Counters *cnt;
    float p_small;
    int count_frag;
    int emit_small;            // 0: records only (replay after a record-list overflow)
    const int *alive;          // chunk culling: the chunks that may reach the view, nullptr = every chunk
    const unsigned long long *cull_info;   // [0] = number of entries of `alive`
};

static_assert(CHUNK == BOUNDS_BLOCK, "block bounds are kept per chunk-sized block");
constexpr int T23_FLOATS = 17 * 17 + 9 * 9;      // kernel S's LDS copy of mip levels 2 and 3, zero-padded (see the kernel)

// ---------------------------------------------------------------------------------------------
// chunk culling: which chunks can reach the view at all
// ---------------------------------------------------------------------------------------------
// A chunk is 512 consecutive particles; with a load-time spatial order (tsp_reorder_spatial) they fill a small box.  The box of a
// chunk (union of the one or two bounds blocks it overlaps) is mapped through the camera: if it lies outside the clip cube by more
// than the widest footprint's half-width plus a generous rounding allowance (1e-4 of the magnitudes involved + 2 pixels), none of
// its particles passes kernel S's own tests -- the chunk is never read.  Every comparison is written so that NaN / infinite
// bounds keep the chunk.
struct CullArgs {
    const int64_t *ranges; int n_ranges; int n_chunks;
    Camera cam;
    const float4 *bounds; int64_t n_particles;
    int *alive; unsigned long long *info;
    int *wg_count;             // per 256 chunks: survivors, then (after the scan) their exclusive prefix
};

__device__ __forceinline__ bool box_outside_view(const Camera &c, float4 lo, float4 hi) {
    if (!(c.sf > 0.0f)) return false;
    const float bx = 0.5f * lo.x + 0.5f * hi.x, by = 0.5f * lo.y + 0.5f * hi.y, bz = 0.5f * lo.z + 0.5f * hi.z;
    const float ex = 0.5f * hi.x - 0.5f * lo.x, ey = 0.5f * hi.y - 0.5f * lo.y, ez = 0.5f * hi.z - 0.5f * lo.z;
    const float s = (c.sf * lo.w) * 2.0f;           // clip-space half-width of the widest footprint (project(): P / 2 pixels = s clip units)
    const float px = 4.0f / c.Rf;                   // two pixels in clip units
    bool out = false;
#pragma unroll
    for (int row = 0; row < 3; ++row) {
        const float a = c.m[4 * row] * bx, b = c.m[4 * row + 1] * by, d = c.m[4 * row + 2] * bz, t = c.m[4 * row + 3];
        const float centre = ((a + b) + d) + t;
        const float ext = (__builtin_fabsf(c.m[4 * row]) * ex + __builtin_fabsf(c.m[4 * row + 1]) * ey) + __builtin_fabsf(c.m[4 * row + 2]) * ez;
        const float mag = (((__builtin_fabsf(a) + __builtin_fabsf(b)) + __builtin_fabsf(d)) + __builtin_fabsf(t)) + ext;
        if (row < 2) {
            const float reach = (ext + s) + (1e-4f * (mag + s) + px);
            out = out || (centre - reach > 1.0f) || (centre + reach < -1.0f);
        } else {                                    // fixed-function clip 0 <= z <= 1: the footprint has no depth
            const float reach = ext + (1e-4f * mag + 1e-6f);
            out = out || (centre - reach > 1.0f) || (centre + reach < 0.0f);
        }
    }
    return out;
}

// Ordered compaction in three small launches (the order of the chunks matters to kernel S: a list in arrival order cost it
// 1 ms of 17 at 1e9 particles): PASS 0 counts the survivors of every 256 chunks, the scan turns the counts into offsets,
// PASS 1 repeats the test and writes the list.
__device__ __forceinline__ bool chunk_alive(const CullArgs &a, int c, int &cnt) {
    const int n_ranges = a.n_ranges;
    const int64_t *starts = a.ranges, *lens = starts + n_ranges, *cprefix = starts + 2 * n_ranges;
    int lo = 0, hi = n_ranges - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cprefix[mid] <= c) lo = mid; else hi = mid - 1;
    }
    const int64_t in_range = (int64_t)(c - cprefix[lo]) * CHUNK;
    const int64_t first = starts[lo] + in_range;
    cnt = (int)min((int64_t)CHUNK, lens[lo] - in_range);
    const int64_t b0 = first / BOUNDS_BLOCK, b1 = (first + cnt - 1) / BOUNDS_BLOCK;
    float4 blo = a.bounds[2 * b0], bhi = a.bounds[2 * b0 + 1];
    if (b1 != b0) {
        const float4 l1 = a.bounds[2 * b1], h1 = a.bounds[2 * b1 + 1];
        blo = make_float4(fminf(blo.x, l1.x), fminf(blo.y, l1.y), fminf(blo.z, l1.z), fmaxf(blo.w, l1.w));
        bhi = make_float4(fmaxf(bhi.x, h1.x), fmaxf(bhi.y, h1.y), fmaxf(bhi.z, h1.z), 0.0f);
    }
    return !box_outside_view(a.cam, blo, bhi);
}

template <int PASS>
__global__ __launch_bounds__(256) void chunk_cull_kernel(CullArgs a) {
    __shared__ int s_cnt[4];
    const int c = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int cnt = 0;
    const bool alive = (c < a.n_chunks) && chunk_alive(a, c, cnt);
    const unsigned long long m = __ballot(alive);
    if (lane == 0) s_cnt[wv] = __popcll(m);
    if (PASS == 0) {
        unsigned long long dropped = alive ? 0ull : (unsigned long long)cnt;
        for (int o = 32; o; o >>= 1) dropped += __shfl_xor((long long)dropped, o);
        if (lane == 0 && dropped) atomicAdd(&a.info[1], dropped);
    }
    __syncthreads();
    if (PASS == 0) {
        if (threadIdx.x == 0) a.wg_count[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    } else if (alive) {
        int pos = a.wg_count[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));      // (now the exclusive prefix)
        for (int w = 0; w < wv; ++w) pos += s_cnt[w];
        a.alive[pos] = c;
    }
}

// exclusive prefix sum of the per-workgroup counts in place (one workgroup; <= a few thousand entries), total -> info[0]
__global__ __launch_bounds__(1024) void chunk_cull_scan_kernel(int *wg_count, int n, unsigned long long *info) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? wg_count[i] : 0;
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wv; ++w) before += s_wave[w];
        if (i < n) wg_count[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) info[0] = (unsigned long long)s_carry;
}

// WC = channels kept in the LDS window: 1 for a density-only render (channel 1 is identically 0:
// half the LDS and half the atomics), else the image's channel count.
template <int MODE, int WC>
__global__ __launch_bounds__(SBLOCK, TSP_S_OCC) void splat_stream_kernel(StreamArgs /*read through the kernarg pointer*/) {
    // The ~45 arguments are NOT held in scalar registers for the whole kernel (they overflowed the scalar file: > 100 SGPRs
    // spilled into VGPR lanes and came back through ~380 v_readlane -- a fifth of the VALU instructions of the raster phase):
    // every phase reads the few it needs from the kernarg segment again (s_load through a laundered pointer).
    typedef const __attribute__((address_space(4))) StreamArgs CArgs;
    auto KA = []() -> CArgs * {
        CArgs *p = (CArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return p;
    };
    constexpr int C = (MODE == TSP_MODE_RGB) ? 4 : 2;
    constexpr int NW = (MODE == TSP_MODE_RGB) ? 2 : 1;      // extra weights per record
    constexpr int WIN = WinSize<WC>::value;
    extern __shared__ __attribute__((aligned(16))) double smem_d[];
    double *win = smem_d;                                            // [WC][WIN*WIN]
    // mip levels 2 (16 x 16) and 3 (8 x 8), each padded with one ZERO row and one ZERO column (17 x 17 and 9 x 9 floats): what a
    // lane reads for a pixel its footprint does not cover (phase 4)
    float *T23 = reinterpret_cast<float *>(win + WC * WIN * WIN);
    constexpr int T23_L3 = 17 * 17;
    __shared__ unsigned s_red[SWAVES][2];
    __shared__ int s_cnt[SWAVES];
    __shared__ long long s_base[2];
    __shared__ int s_batch;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Work distribution.  The list of chunks to draw (all of them, or the survivors of the culling pass, whose number only the device
    // knows) is cut into BATCHES of consecutive chunks.  Workgroup b starts on batch b; with `dynamic` the launch holds only about as
    // many workgroups as the device keeps resident, and each takes further batches from a shared counter (Counters::next_chunk)
    // until the list is exhausted -- chunks differ in cost by an order of magnitude (a chunk of the dense core rasterises a few
    // hundred pixels, one of the outskirts tens of thousands), and with static shares the last workgroups set the kernel's time:
    // one shard of the 1e9-particle snapshot took 2.5 ms where its share of the whole snapshot's time is 1.8.  Batches shrink
    // towards the end of the list (a batch is 1 / (2 gridDim) of what is left, between S_BATCH_MIN chunks and chunks_per_block).
    // A batch is taken one batch AHEAD (while the first chunk of the current one is drawn), so the attribute prefetch of the
    // next chunk runs across batch borders.  [b0, b1) = current batch, [n0, n1) = the next one (NO_CHUNK: none / not yet known).
    constexpr int NO_CHUNK = 0x7fffffff, S_BATCH_MIN = 4;
    int R, b0, b1, n0 = NO_CHUNK, n1 = NO_CHUNK;
    auto list_size = [&]() -> int {
        CArgs *ap = KA();
        return ap->alive ? (int)ap->cull_info[0] : ap->n_chunks;
    };
    {
        CArgs *ap = KA();
        R = ap->cam.R;
        const int per = ap->chunks_per_block, n_todo = list_size();
        b0 = blockIdx.x * per;
        b1 = min(b0 + per, n_todo);
        if (b0 >= b1) return;                              // a workgroup without a batch leaves before it touches LDS
        const float *mips = ap->mips;
        for (int i = tid; i < WC * WIN * WIN; i += SBLOCK) win[i] = 0.0;
        for (int i = tid; i < T23_FLOATS; i += SBLOCK) {
            float t = 0.0f;
            if (i < 17 * 17) { const int r = i / 17, c = i - r * 17; if (r < 16 && c < 16) t = mips[5120 + r * 16 + c]; }
            else { const int q = i - 17 * 17, r = q / 9, c = q - r * 9; if (r < 8 && c < 8) t = mips[5376 + r * 8 + c]; }
            T23[i] = t;
        }
    }
    __syncthreads();

    // window state (uniform): origin and the dirty rectangle (window coordinates, inclusive)
    int wox = 0, woy = 0;
    int dx0 = WIN, dy0 = WIN, dx1 = -1, dy1 = -1;
    unsigned long long n_small = 0, n_cull = 0, n_frag = 0;
#ifdef TSP_S_DEBUG
    unsigned long long dbg_slots = 0;
#endif


    auto flush = [&]() {
        // one global atomic per touched pixel and channel; only the dirty rectangle is visited
        if (dx1 >= dx0) {
            // a wave per window row, a lane per column (the window is at most 60 pixels wide): no index division (as one linear
            // index over the rectangle, its quotient and remainder were ~25 of the ~35 instructions per visited pixel)
            static_assert(WIN <= 64, "one lane per window column");
            double *img = KA()->img;
            const int wx = dx0 + lane;
            if (wx <= dx1) {
                const int gx = wox + wx;
                for (int wy = dy0 + wv; wy <= dy1; wy += SWAVES) {
                    const int gy = woy + wy;
                    const int o = __mul24(wy, WIN) + wx;
#pragma unroll
                    for (int c = 0; c < WC; ++c) {
                        const double v = win[c * WIN * WIN + o];
                        if (v != 0.0) {
                            if (gx < R && gy < R) gatomic_add(img + ((size_t)gy * R + gx) * C + c, v);
                            win[c * WIN * WIN + o] = 0.0;
                        }
                    }
                }
            }
        }
        dx0 = WIN; dy0 = WIN; dx1 = -1; dy1 = -1;
    };

    // locate chunk c: range r, particles [first, first + cnt)
    auto chunk_id = [&](int c) -> int {            // entry c of the list of surviving chunks (or c itself)
        const int *alive = KA()->alive;
        return alive ? alive[c] : c;
    };
    auto locate = [&](int c /* chunk id */, int64_t &first, int &cnt) {
        CArgs *ap = KA();
        const int n_ranges = ap->n_ranges;
        const int64_t *starts = ap->ranges, *lens = starts + n_ranges, *cprefix = starts + 2 * n_ranges;
        int r = 0;
        if (n_ranges > 1) {
            int lo = 0, hi = n_ranges - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cprefix[mid] <= c) lo = mid; else hi = mid - 1;
            }
            r = lo;
        }
        const int64_t in_range = (int64_t)(c - cprefix[r]) * CHUNK;
        first = starts[r] + in_range;
        cnt = (int)min((int64_t)CHUNK, lens[r] - in_range);
    };
    // The attributes of a chunk are loaded one chunk AHEAD: the loads of chunk c + 1 are issued when chunk c has been
    // classified and are in flight while it is rasterised (the kernel is latency-bound: 60 % of its wave time was s_waitcnt).
    constexpr int NATTR = (MODE == TSP_MODE_RGB) ? 7 : 6;      // x y z h + (m, q) or (r, g, b)
    float L[KPT][NATTR];
    auto load_chunk = [&](int64_t first, int cnt) {
        CArgs *ap = KA();
        const float *qx = ap->p.x, *qy = ap->p.y, *qz = ap->p.z, *qh = ap->p.h;
        // the camera-independent weights m / h^2 (rgb: r, g, b over h^2) are formed once per upload (ensure_weights): the
        // same float32 operations on the same operands as sph.wgsl:76-83, done once instead of every frame
        const float *q4 = (MODE == TSP_MODE_RGB) ? ap->p.wr : ap->p.wm;
        const float *q5 = (MODE == TSP_MODE_RGB) ? ap->p.wg : ((MODE != TSP_MODE_DEPTH) ? ap->p.q : nullptr);
        const float *q6 = (MODE == TSP_MODE_RGB) ? ap->p.wb : nullptr;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            // lanes past the end of a short chunk load its last particle again (cnt >= 1): no predication, no zero fill;
            // the classification masks them out
            const int64_t i = first + min(k * SBLOCK + tid, cnt - 1);
            L[k][0] = qx[i]; L[k][1] = qy[i]; L[k][2] = qz[i]; L[k][3] = qh[i];
            L[k][4] = q4[i];
            if (MODE == TSP_MODE_RGB) { L[k][5] = q5[i]; L[k][NATTR - 1] = q6[i]; }
            else L[k][5] = q5 ? q5[i] : 0.0f;
        }
    };
    int64_t first = 0, first_next = 0;
    int cnt = 0, cnt_next = 0;
    // (the list entry of chunk c + 1 is fetched a chunk early, behind the attribute loads: nothing waits for it)
    int id_next = 0, id_next_of = NO_CHUNK;        // list entry id_next belongs to list position id_next_of
    // the chunk after list position x, as far as this workgroup knows it now
    auto successor = [&](int x) -> int {
        if (x >= b0 && x < b1) return (x + 1 < b1) ? x + 1 : n0;
        return (x + 1 < n1) ? x + 1 : NO_CHUNK;     // x lies in the next batch: the batch after that is not known yet
    };
    locate(chunk_id(b0), first, cnt); load_chunk(first, cnt);
    if (b0 + 1 < b1) { id_next = chunk_id(b0 + 1); id_next_of = b0 + 1; }
    bool batch_head = true;                         // c is the first chunk of its batch: the next batch is taken now
    for (int c = b0;;) {
        // the size of the batch taken now: a share of what is left of the list (uniform)
        int grab = 0;
        if (batch_head) {
            CArgs *ap = KA();
            if (ap->dynamic) {
                const int n_todo = list_size();
                // (a batch that ends the list: nothing is left to take)
                if (b1 < n_todo) grab = max(S_BATCH_MIN, min(ap->chunks_per_block, (n_todo - b0) / (2 * (int)gridDim.x)));
            }
        }

        // ---- phase 1: projection, exact covered pixel ranges, classification ------------------------------------
        // Bounding boxes are kept as PACKED 16-bit pixel pairs (x | y << 16; the image has <= 16384 pixels per side):
        // one v_pk_min_u16 / v_pk_max_u16 per reduction step does both axes.
        float pcx[KPT], pcy[KPT], PP[KPT], w0[KPT], w1[KPT], w2[KPT];
        int cls[KPT];
        unsigned xr[KPT], yr[KPT];     // first covered pixel | (number of covered pixels << 16), clipped to the image
        unsigned s_lo = 0xffffffffu, s_hi = 0u;      // covered-pixel bounding box of the small footprints: (ilo | jlo << 16), (ihi | jhi << 16)
        int my_counts = 0;                           // mid records | huge records << 16 of this lane
        Camera cam;                                  // (16 scalar registers, live in this phase only)
        float p_small;
        {
            CArgs *ap = KA();
#pragma unroll
            for (int i = 0; i < 12; ++i) cam.m[i] = ap->cam.m[i];
            cam.sf = ap->cam.sf; cam.Rf = ap->cam.Rf; cam.halfR = ap->cam.halfR; cam.R = R;
            p_small = ap->p_small;
        }
        // Branch-free: every lane runs the whole classification on whatever its registers hold (lanes past the end of the chunk
        // re-read the chunk's last particle) and the outcome is masked at the end -- nothing below reads pcx .. w2, xr, yr of a
        // particle whose class is CLS_NONE.  (As nested ifs with zeroed defaults the compiler re-materialised the fourteen
        // defaults at every nesting level: a fifth of this phase's instructions were v_mov.)
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int li = k * SBLOCK + tid;
            const bool in_chunk = li < cnt;
            const float h = L[k][3];
            const Proj pr = project<false>(cam, L[k][0], L[k][1], L[k][2], h);
            int ilo = 1, ihi = 0, jlo = 1, jhi = 0;
            if (__ballot(in_chunk && pr.keep) != 0ull) {        // (a chunk wholly outside the z-slab skips the range arithmetic)
                // any pixel centre covered?  (exact test via the canonical interval)
                cover_range(pr.pcx, pr.half, R, ilo, ihi);
                cover_range(pr.pcy, pr.half, R, jlo, jhi);
            }
            const bool vis = in_chunk && pr.keep && (ilo <= ihi) && (jlo <= jhi);
            // a small footprint covers <= 16 pixels per axis and R <= 16384: both fit 16 bits
            xr[k] = (unsigned)ilo | ((unsigned)min(ihi - ilo + 1, 0xffff) << 16);
            yr[k] = (unsigned)jlo | ((unsigned)min(jhi - jlo + 1, 0xffff) << 16);
            const unsigned lo = (unsigned)ilo | ((unsigned)jlo << 16), hi = (unsigned)ihi | ((unsigned)jhi << 16);
            if (MODE == TSP_MODE_RGB) {
                w0[k] = L[k][4]; w1[k] = L[k][5]; w2[k] = L[k][NATTR - 1];
            } else {
                w0[k] = L[k][4];
                w1[k] = (MODE == TSP_MODE_DEPTH) ? pr.cz : L[k][5];
                w2[k] = 0.0f;
            }
            pcx[k] = pr.pcx; pcy[k] = pr.pcy; PP[k] = pr.P;      // (1 / P is formed in phase 4, by the waves that rasterise)
            const int c_of_p = (pr.P < p_small) ? CLS_SMALL : ((pr.P < P_BILINEAR) ? CLS_MID : CLS_HUGE);
            cls[k] = vis ? c_of_p : CLS_NONE;
            const bool is_small = cls[k] == CLS_SMALL, is_mid = cls[k] == CLS_MID;
            s_lo = is_small ? pk_min_u16(s_lo, lo) : s_lo; s_hi = is_small ? pk_max_u16(s_hi, hi) : s_hi;
            my_counts += is_mid ? 1 : ((cls[k] == CLS_HUGE) ? (1 << 16) : 0);
            n_cull += (in_chunk && !vis) ? 1ull : 0ull;
        }

        // the next chunk's attributes start to load now; phases 2 - 5 of this chunk hide their latency
        // (a batch holds >= 2 chunks unless it ends the list, and the next batch is known from the first chunk's phase 5 on:
        // when c is the last chunk of its batch, n0 is final)
        const int c_next = successor(c);
        if (c_next != NO_CHUNK) {
            const int id = (id_next_of == c_next) ? id_next : chunk_id(c_next);
            locate(id, first_next, cnt_next); load_chunk(first_next, cnt_next);
            const int c_next2 = successor(c_next);
            if (c_next2 != NO_CHUNK) { id_next = chunk_id(c_next2); id_next_of = c_next2; }
        }

        // ---- phase 2: the chunk's small-footprint bounding box places the LDS window (uniform) ------
        // (each exchange has its own LDS scratch, so one barrier per exchange suffices: the barriers of
        // the following exchanges order its readers before the next chunk's writers)
        s_lo = wave_pk_min_u16(s_lo); s_hi = wave_pk_max_u16(s_hi);
        if (lane == 0) { s_red[wv][0] = s_lo; s_red[wv][1] = s_hi; }
        __syncthreads();
        s_lo = s_red[0][0]; s_hi = s_red[0][1];
#pragma unroll
        for (int w = 1; w < SWAVES; ++w) { s_lo = pk_min_u16(s_lo, s_red[w][0]); s_hi = pk_max_u16(s_hi, s_red[w][1]); }
        if ((s_hi & 0xffffu) >= (s_lo & 0xffffu)) {          // the chunk has small footprints
            const int ix0 = (int)(s_lo & 0xffffu), iy0 = (int)(s_lo >> 16), ix1 = (int)(s_hi & 0xffffu), iy1 = (int)(s_hi >> 16);
            const bool inside = ix0 >= wox && ix1 < wox + WIN && iy0 >= woy && iy1 < woy + WIN;
            if (!inside) {
                flush();
                // centre the window on the chunk's small-footprint bounding box
                wox = (ix0 + ix1 + 1 - WIN) / 2;
                woy = (iy0 + iy1 + 1 - WIN) / 2;
                wox = max(0, min(wox, R - WIN));
                woy = max(0, min(woy, R - WIN));
            }
            // dirty rectangle grows by this chunk's bbox (clipped to the window)
            dx0 = min(dx0, max(ix0 - wox, 0)); dx1 = max(dx1, min(ix1 - wox, WIN - 1));
            dy0 = min(dy0, max(iy0 - woy, 0)); dy1 = max(dy1, min(iy1 - woy, WIN - 1));
        }

        // ---- phase 3: a small footprint that does not fit the window joins the MID list (kernel G
        //      rasterises any width with the same nearest-mip rule), so phase 4 is LDS-only -----------
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if (cls[k] != CLS_SMALL) continue;
            const int ilo = (int)(xr[k] & 0xffffu), ihi = ilo + (int)(xr[k] >> 16) - 1;
            const int jlo = (int)(yr[k] & 0xffffu), jhi = jlo + (int)(yr[k] >> 16) - 1;
            if (ilo < wox || ihi >= wox + WIN || jlo < woy || jhi >= woy + WIN) { cls[k] = CLS_MID; my_counts += 1; }
        }
        // record counts + offsets
        const int counts_incl = wave_incl_scan(my_counts, lane);       // both counters at once: <= 128 each per wave
        if (lane == 63) s_cnt[wv] = counts_incl;
        __syncthreads();
        int counts_before = 0, counts_total = 0;
#pragma unroll
        for (int w = 0; w < SWAVES; ++w) {
            if (w < wv) counts_before += s_cnt[w];
            counts_total += s_cnt[w];
        }
        const int mid_total = counts_total & 0xffff, huge_total = counts_total >> 16;
        // reserve contiguous runs in the record lists (one atomic per chunk and list).  The returning atomics are ISSUED here and
        // their results are published after phase 4: the round trip to L2 (and the wait for the next chunk's attribute loads
        // queued before it) overlaps the rasteriser instead of holding all four waves at a barrier
        long long r_mid = 0, r_huge = 0;
        unsigned r_batch = 0;
        if (tid == 0) {
            CArgs *ap = KA();
            Counters *cntp = ap->cnt;
            if (grab) r_batch = atomicAdd(reinterpret_cast<unsigned *>(&cntp->next_chunk), (unsigned)grab);
            if (mid_total) r_mid = (long long)atomicAdd(&cntp->n_mid, (unsigned long long)mid_total);
            if (huge_total) r_huge = (long long)atomicAdd(&cntp->n_huge, (unsigned long long)huge_total);
        }
        const int my_mid = my_counts & 0xffff, my_huge = my_counts >> 16;
        const int mid_before = counts_before & 0xffff, huge_before = counts_before >> 16;
        const int mid_incl = counts_incl & 0xffff, huge_incl = counts_incl >> 16;

        // ---- phase 4: rasterise small footprints (one lane per particle, mips 3 / 2 nearest) -----
        // Nearest sampling is separable: texel column tx(i) and row ty(j).  The column indices of the whole footprint are
        // computed once (4 bits each, <= 16 columns), then per pixel row: one ty, the row's LUT values fetched in a batch
        // (LDS latency paid once per row, not per pixel), and per pixel a multiply, a conversion and the ds_add_f64.
        // Loop bounds are wave-uniform (maximum over the lanes), lanes are predicated.
        const int emit_small = KA()->emit_small, count_frag = KA()->count_frag;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const bool act = (cls[k] == CLS_SMALL) && emit_small;
            if (__ballot(act) == 0ull) continue;
            const int nx = act ? (int)(xr[k] >> 16) : 0, ny = act ? (int)(yr[k] >> 16) : 0;
            const int ilo = (int)(xr[k] & 0xffffu), jlo = (int)(yr[k] & 0xffffu);
            const float half = 0.5f * PP[k], ip = 1.0f / PP[k];      // project(): invP = 1 / P (deferred records carry P, not 1 / P)
            // mip 3 (8 x 8) up to 11.3 px, mip 2 (16 x 16) above (p_small may reach 15 px)
            const bool lvl2 = PP[k] > P_L2;
            const float nf = lvl2 ? 16.0f : 8.0f;
            const int nm1 = lvl2 ? 15 : 7;
            // Texel column of footprint column c, 5 bits each, six per register.  A column this lane does not cover (c >= nx; every
            // column of a lane that rasterises nothing) points at the ZERO column of the padded level, a row it does not cover at the
            // zero row: the texel read is then exactly 0 and the one test `texel != 0` below -- there anyway, because the corner
            // texels of the kernel image are 0 -- masks the lane.  (Until round 5 every pixel step combined a row mask, a column
            // mask and that test: ~7 scalar instructions per step, with the sixteen column masks spilled into VGPR lanes.)
            const unsigned zc = lvl2 ? 16u : 8u;
            const unsigned zc6 = zc * 0x2108421u;     // the zero column in all six fields
            unsigned pk[3] = {zc6, zc6, zc6};
            const float x0f = (float)ilo + 0.5f;      // centre of the first covered column ((float)(ilo + c) + 0.5f == x0f + c exactly)
            int maxnx = 0;
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) {
                if (__ballot(ci < nx) == 0ull) break;
                maxnx = ci + 1;
                const float dx = (x0f + (float)ci) - pcx[k];
                const int tx = floor_clamp_v(((dx + half) * ip) * nf, nm1);
                pk[ci / 6] ^= ((ci < nx) ? ((unsigned)tx ^ zc) : 0u) << (5 * (ci % 6));
            }
            const float y0f = (float)jlo + 0.5f;
            const float *lut = T23 + (lvl2 ? 0 : T23_L3);
            const int lstride = lvl2 ? 17 : 9;
            double *wbase = win + __mul24(jlo - woy, WIN) + (ilo - wox);
            if (MODE != TSP_MODE_RGB && WC == 1 && maxnx == 1) {
                // Every footprint of this wave is one pixel COLUMN wide: the dense core of the snapshot, where thousands of
                // sub-pixel particles share a pixel and the lanes of a step mostly hit the SAME pixel -- a same-address
                // ds_add_f64 costs ~11 clk per extra lane (most of such a wave's lanes draw nothing: a footprint narrower than a
                // pixel rarely covers a pixel centre).  When they are single pixels, the lanes that hit one pixel are summed
                // across the wave first (DPP tree) and one lane issues the one atomic.
                if (__ballot(act && ny != 1) == 0ull) {
                    const float dy = y0f - pcy[k];
                    const int ty = floor_clamp_v(((dy + half) * ip) * nf, nm1);
                    const float kv = act ? lut[__mul24(ty, lstride) + (int)(pk[0] & 31u)] : 0.0f;
                    const float val = kv * w0[k];
                    const int key = (int)(wbase - win);
                    unsigned long long todo = __ballot(act && val != 0.0f);
                    // up to four distinct pixels are summed across the wave (float32 tree sums of <= 64 terms), one atomic
                    // each; whatever is left after that (a sparse wave) adds lane by lane as before
                    for (int round = 0; round < 4 && todo != 0ull; ++round) {
                        const int src = __ffsll((long long)todo) - 1;
                        const int k0 = __builtin_amdgcn_readlane(key, src);
                        const bool match = act && val != 0.0f && key == k0;
                        const unsigned long long mm = __ballot(match);
                        todo &= ~mm;
                        if (__popcll(mm) == 1) {
                            if (lane == src) latomic_add(win + key, val);
                        } else {
                            const float sum = wave_sum(match ? val : 0.0f);
                            if (lane == src) latomic_add(win + key, sum);
                        }
                    }
                    if ((todo >> lane) & 1ull) latomic_add(win + key, val);
                    n_small += act ? 1 : 0;
                    if (count_frag) n_frag += act ? 1ull : 0ull;
                    continue;
                }
            }
            for (int r = 0; r < 16; ++r) {
                const bool rowact = r < ny;
                if (__ballot(rowact) == 0ull) break;
                const float dy = (y0f + (float)r) - pcy[k];
                const int ty = floor_clamp_v(((dy + half) * ip) * nf, nm1);
                const float *lrow = lut + __mul24(rowact ? ty : (int)zc, lstride);      // (24-bit multiply: full rate; v_mul_lo_u32 runs at a quarter)
                double *wrow = wbase + r * WIN;
                // a group = six footprint columns (one register of texel fields); its pixel steps are unrolled for 2, 4 or 6
                // columns -- the wave's widest footprint rounded up to even -- with no test between them: a column beyond the
                // wave's widest footprint reads the zero column in every lane
                auto group = [&](int g, auto nc_c) {
                    constexpr int NC = decltype(nc_c)::value;
                    float v[NC];
#pragma unroll
                    for (int j = 0; j < NC; ++j) v[j] = lrow[(pk[g] >> (5 * j)) & 31u];
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        const int ci = 6 * g + j;
#ifdef TSP_S_DEBUG       // analysis build: lane slots the raster loop spends (64 per executed pixel step), reported as the mid fragment count
                        if (count_frag && lane == 0) dbg_slots += 64ull;
#endif
                        double *d = wrow + ci;
                        if (MODE == TSP_MODE_RGB) {      // (the count channel also counts the fragments whose texel is exactly 0)
                            if (rowact && ci < nx) {
                                latomic_add(d, v[j] * w0[k]); latomic_add(d + WIN * WIN, v[j] * w1[k]);
                                latomic_add(d + 2 * WIN * WIN, v[j] * w2[k]); latomic_add(d + 3 * WIN * WIN, 1.0f);
                            }
                        } else if (v[j] != 0.0f) {       // covered, and not one of the kernel image's zero corner texels
                            const float val = v[j] * w0[k];
                            latomic_add(d, val);
                            if (WC > 1) latomic_add(d + WIN * WIN, val * w1[k]);
                        }
                    }
                };
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int nc = maxnx - 6 * g;                // (wave-uniform)
                    if (nc <= 0) break;
                    if (g == 2 || nc <= 4) {                     // (group 2 holds columns 12 .. 15)
                        if (nc <= 2) group(g, std::integral_constant<int, 2>());
                        else group(g, std::integral_constant<int, 4>());
                    } else group(g, std::integral_constant<int, 6>());
                }
            }
            if (act) {
                ++n_small;
                if (count_frag) n_frag += (unsigned long long)(nx * ny);
            }
        }

        // ---- phase 5: append the deferred footprints -----------------------------------------------
        if (tid == 0) {
            s_base[0] = r_mid; s_base[1] = r_huge;
            if (grab) s_batch = (int)r_batch;
        }
        __syncthreads();
        if (grab) {
            // the counter counts the chunks handed out beyond the static first batches
            CArgs *ap = KA();
            const int n_todo = list_size(), start = (int)gridDim.x * ap->chunks_per_block + s_batch;
            if (start < n_todo) { n0 = start; n1 = min(start + grab, n_todo); }
        }
        const long long mid_base = s_base[0], huge_base = s_base[1];
        {
            CArgs *ap = KA();
            float4 *const mid_geom = ap->mid_geom, *const huge_geom = ap->huge_geom;
            float *const mid_w = ap->mid_w, *const huge_w = ap->huge_w;
            const long long mid_capacity = ap->mid_capacity, huge_capacity = ap->huge_capacity;
            long long mpos = mid_base + mid_before + (mid_incl - my_mid);
            long long hpos = huge_base + huge_before + (huge_incl - my_huge);
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                if (cls[k] == CLS_MID) {
                    if (mpos < mid_capacity) {
                        mid_geom[mpos] = make_float4(pcx[k], pcy[k], PP[k], w0[k]);
                        mid_w[mpos * NW] = w1[k];
                        if (NW == 2) mid_w[mpos * NW + 1] = w2[k];
                    }
                    ++mpos;
                } else if (cls[k] == CLS_HUGE) {
                    if (hpos < huge_capacity) {
                        huge_geom[hpos] = make_float4(pcx[k], pcy[k], PP[k], w0[k]);
                        huge_w[hpos * NW] = w1[k];
                        if (NW == 2) huge_w[hpos * NW + 1] = w2[k];
                    }
                    ++hpos;
                }
            }
        }
        if (c_next == NO_CHUNK) break;
        first = first_next; cnt = cnt_next;
        batch_head = false;
        if (c + 1 == b1) { b0 = n0; b1 = n1; n0 = n1 = NO_CHUNK; batch_head = true; }
        c = c_next;
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
        Counters *cntp = KA()->cnt;
        if (n_small) atomicAdd(&cntp->n_small, n_small);
        if (n_cull) atomicAdd(&cntp->n_culled, n_cull);
        if (n_frag) { atomicAdd(&cntp->n_fragments, n_frag); atomicAdd(&cntp->n_frag_class[0], n_frag); }
#ifdef TSP_S_DEBUG
        if (dbg_slots) atomicAdd(&cntp->n_frag_class[1], dbg_slots);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// rgb fragment-counter channel of the deferred (MID and HUGE) footprints
// ---------------------------------------------------------------------------------------------
// fragment_rgb writes (k r, k g, k b, 1): channel 3 counts the footprint SQUARES covering a pixel, also where
// the kernel value is exactly 0.  Per footprint that is the indicator of a pixel rectangle, so instead of one add
// per fragment kernels G and H2 leave the channel alone (they may then skip whatever lies outside the kernel's disc
// and carry three accumulator sets instead of four) and the rectangles are summed exactly in integers: +-1 at the four corners of each rectangle, then a 2-D prefix
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

static int add_rect_counts(tsp_context *ctx, const float4 *mid_geom, long long n_mid, const float4 *huge_geom, long long n_huge) {
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
    // the list of chunks that survive culling (+ the per-workgroup counts of the culling pass behind it)
    if (ws.chunk_capacity < n_chunks) {
        if (ws.alive_list) TSP_HIP(hipFree(ws.alive_list));
        ws.alive_list = nullptr;
        ws.chunk_capacity = (int64_t)n_chunks + n_chunks / 4 + 64;
        TSP_HIP(hipMalloc((void **)&ws.alive_list, (ws.chunk_capacity + ws.chunk_capacity / 256 + 2) * sizeof(int)));
    }
    // record lists: start modest, grow to the exact need when a frame overflows (rare)
    int rc;
    if ((rc = ensure_weights(ctx, MODE == TSP_MODE_RGB))) return rc;      // m / h^2 (rgb / h^2): once per upload, not per frame
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
    const bool second_channel = (MODE == TSP_MODE_DEPTH) || (MODE == TSP_MODE_RGB) || (ctx->p.q != nullptr && ctx->use_quantity);
    const int WCr = (MODE == TSP_MODE_RGB) ? 4 : (second_channel ? 2 : 1);
    const int WIN = (WCr == 1) ? WinSize<1>::value : WinSize<C>::value;
    const size_t smem_s = (size_t)WCr * WIN * WIN * sizeof(double) + T23_FLOATS * sizeof(float);
    if (!(ctx->kernel_attr_done & (1u << MODE))) {
        TSP_HIP(hipFuncSetAttribute((const void *)splat_stream_kernel<MODE, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)C * WinSize<C>::value * WinSize<C>::value * sizeof(double) + T23_FLOATS * sizeof(float))));
        TSP_HIP(hipFuncSetAttribute((const void *)splat_stream_kernel<MODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)WinSize<1>::value * WinSize<1>::value * sizeof(double) + T23_FLOATS * sizeof(float))));
        ctx->kernel_attr_done |= 1u << MODE;
    }

    // Chunk culling (option chunk_cull, on by default): one small kernel lists the chunks whose bounds can reach the view; kernel S
    // shares out that list.  Not for small render blocks (an interactive first block is ~200 chunks: the extra launch would
    // cost more than the chunks it could drop).
    const bool cull = ctx->chunk_cull && n_chunks >= 4096;
    unsigned long long cull_info_h[2] = {0, 0};
    ctx->chunk_culled_particles = 0;
    TSP_HIP(hipEventRecord(ctx->ev[7], st));      // (the culling passes, and the block bounds when they are stale, count as kernel S's time)
    if (cull) {
        if ((rc = ensure_block_bounds(ctx))) return rc;
        if (!ws.cull_info) TSP_HIP(hipMalloc((void **)&ws.cull_info, 2 * sizeof(unsigned long long)));
        TSP_HIP(hipMemsetAsync(ws.cull_info, 0, 2 * sizeof(unsigned long long), st));
        CullArgs ca;
        ca.ranges = ws.range_prefix; ca.n_ranges = n_ranges; ca.n_chunks = n_chunks; ca.cam = cam;
        ca.bounds = ws.block_bounds; ca.n_particles = ctx->p.n; ca.alive = ws.alive_list; ca.info = ws.cull_info;
        ca.wg_count = ws.alive_list + ws.chunk_capacity;          // (the list's allocation carries the per-workgroup counts behind it)
        const int n_wg = (n_chunks + 255) / 256;
        hipLaunchKernelGGL(chunk_cull_kernel<0>, dim3(n_wg), dim3(256), 0, st, ca);
        hipLaunchKernelGGL(chunk_cull_scan_kernel, dim3(1), dim3(1024), 0, st, ca.wg_count, n_wg, ws.cull_info);
        hipLaunchKernelGGL(chunk_cull_kernel<1>, dim3(n_wg), dim3(256), 0, st, ca);
        TSP_HIP(hipGetLastError());
    }

    Counters hc, carry;
    memset(&carry, 0, sizeof(carry));
    for (int attempt = 0; attempt < 2; ++attempt) {
        StreamArgs sa;
        sa.p = parts;
        sa.alive = cull ? ws.alive_list : nullptr; sa.cull_info = ws.cull_info;
        sa.ranges = ws.range_prefix; sa.n_ranges = n_ranges; sa.n_chunks = n_chunks;
        // Small render blocks (an interactive first block of 1e5 particles is 196 chunks): one static batch per workgroup, as
        // few chunks as it takes to give every CU two workgroups.  Otherwise the launch holds the workgroups the device keeps
        // resident (a few more are harmless: they start late and take what is left) and they share out the list dynamically.
        const int min_cpb = std::max(1, std::min(8, n_chunks / (ctx->cu_count * 2)));
        int per_cu = ctx->stream_blocks_per_cu;
        if (per_cu <= 0) {
            int &occ = ctx->stream_occ[MODE][WCr == 1 ? 0 : 1];      // (per context: occupancy is a property of its device, and contexts render concurrently)
            if (occ <= 0) {
                const void *fn = (WCr == 1) ? (const void *)splat_stream_kernel<MODE, 1> : (const void *)splat_stream_kernel<MODE, C>;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, SBLOCK, smem_s) != hipSuccess || occ <= 0) { (void)hipGetLastError(); occ = 4; }
            }
            per_cu = occ;
        }
        const int resident = ctx->cu_count * per_cu;
        sa.dynamic = (min_cpb >= 8 && n_chunks > resident * 8) ? 1 : 0;
        sa.chunks_per_block = sa.dynamic ? std::max(4, ctx->stream_batch_chunks) : min_cpb;
        const int grid_s = sa.dynamic ? resident : (n_chunks + sa.chunks_per_block - 1) / sa.chunks_per_block;
        sa.cam = cam; sa.mips = ctx->mips; sa.img = ctx->image64;
        sa.mid_geom = (float4 *)ws.mid_geom; sa.mid_w = (float *)ws.mid_w; sa.mid_capacity = ws.mid_capacity;
        sa.huge_geom = (float4 *)ws.huge_geom; sa.huge_w = (float *)ws.huge_w; sa.huge_capacity = ws.huge_capacity;
        sa.cnt = ctx->counters; sa.p_small = ctx->p_small;
        sa.count_frag = ctx->count_fragments ? 1 : 0;
        sa.emit_small = (attempt == 0 && !ctx->debug_no_raster) ? 1 : 0;
        TSP_HIP(hipEventRecord(ctx->ev[2], st));
        if (WCr == 1) hipLaunchKernelGGL((splat_stream_kernel<MODE, 1>), dim3(grid_s), dim3(SBLOCK), smem_s, st, sa);
        else hipLaunchKernelGGL((splat_stream_kernel<MODE, C>), dim3(grid_s), dim3(SBLOCK), smem_s, st, sa);
        TSP_HIP(hipGetLastError());
        TSP_HIP(hipEventRecord(ctx->ev[3], st));
        // the record counts size the two tile launches (and reveal a list overflow)
        TSP_HIP(hipMemcpyAsync(&hc, ctx->counters, sizeof(hc), hipMemcpyDeviceToHost, st));
        if (cull && attempt == 0) TSP_HIP(hipMemcpyAsync(cull_info_h, ws.cull_info, sizeof(cull_info_h), hipMemcpyDeviceToHost, st));
        TSP_HIP(hipStreamSynchronize(st));
        ctx->chunk_culled_particles = (int64_t)cull_info_h[1];
        const bool mid_over = (int64_t)hc.n_mid > ws.mid_capacity, huge_over = (int64_t)hc.n_huge > ws.huge_capacity;
        if (!mid_over && !huge_over) break;
        TSP_REQUIRE(attempt == 0, TSP_ENOMEM, "record lists overflowed twice");
        // enlarge and replay kernel S in records-only mode (its small footprints are already in the image)
        if (mid_over) {
            if ((rc = grow(&ws.mid_geom, &ws.mid_capacity, (int64_t)hc.n_mid + (int64_t)hc.n_mid / 8 + 1024, sizeof(float4)))) return rc;
            if (ws.mid_w) TSP_HIP(hipFree(ws.mid_w));
            TSP_HIP(hipMalloc(&ws.mid_w, (size_t)ws.mid_capacity * 2 * sizeof(float)));
        }
        if (huge_over) {
            if ((rc = grow(&ws.huge_geom, &ws.huge_capacity, (int64_t)hc.n_huge + (int64_t)hc.n_huge / 8 + 1024, sizeof(float4)))) return rc;
            if (ws.huge_w) TSP_HIP(hipFree(ws.huge_w));
            TSP_HIP(hipMalloc(&ws.huge_w, (size_t)ws.huge_capacity * 2 * sizeof(float)));
        }
        TSP_HIP(hipMemsetAsync(ctx->counters, 0, sizeof(Counters), st));
        carry = hc;
    }

    if (carry.n_small || carry.n_fragments) {   // statistics of the first attempt (its small footprints stand)
        hc.n_small += carry.n_small;
        hc.n_fragments += carry.n_fragments;
        hc.n_frag_class[0] += carry.n_frag_class[0];
        TSP_HIP(hipMemcpyAsync(ctx->counters, &hc, sizeof(hc), hipMemcpyHostToDevice, st));
        TSP_HIP(hipStreamSynchronize(st));
    }
    TileArgs ta;
    ta.cam = cam; ta.mips = ctx->mips; ta.img = ctx->image64; ta.cnt = ctx->counters; ta.tiles_x = 0; ta.split = 1;
    ta.count_frag = ctx->count_fragments ? 1 : 0;
    ta.hband_count = nullptr; ta.hband_stride = 0; ta.hband_base = nullptr; ta.n_tiles = 0; ta.item_tile = nullptr; ta.item_base = nullptr;
    // corner culling is exact for the value channels; the rgb counter channel (which also counts zero-valued
    // fragments) is not touched by kernel H2 at all: add_rect_counts() sums the footprint rectangles instead
    ta.disc_k2 = (ctx->lut_zero_outside_disc && !ctx->count_fragments) ? 0.5235f * 0.5235f : 0.0f;
    // Kernels G and H2 only depend on kernel S and add into the float64 image with atomics: option overlap_mid_huge runs them on
    // two streams (no gain measured: both are bound by the vector units).
    hipStream_t st_mid = ctx->overlap_mid_huge ? ctx->stream2 : st;
    if (ctx->overlap_mid_huge) {
        TSP_HIP(hipEventRecord(ctx->ev[8], st));
        TSP_HIP(hipStreamWaitEvent(st_mid, ctx->ev[8], 0));
    }
    if (ctx->debug_fail_stage == 1) { ctx->debug_fail_stage = 0; TSP_REQUIRE(false, TSP_ENOMEM, "injected failure after kernel S (debug_fail_stage)"); }
    // A block of any size draws (sph.py:306-332 has no limit on a block): the tile kernels index their records and work items with
    // 32 bits, so a longer list goes through them in slices -- bin, draw, next slice; the bins of a slice are bounded with it.
    constexpr int NWr = (MODE == TSP_MODE_RGB) ? 2 : 1;            // weight floats per record
    const long long n_mid = (long long)hc.n_mid, n_huge = (long long)hc.n_huge;
    const long long mid_slice = ctx->slice_records > 0 ? ctx->slice_records : (1ll << 27);
    const long long huge_slice = ctx->slice_records > 0 ? ctx->slice_records : (1ll << 30);
    const float4 *mid_geom = (const float4 *)ws.mid_geom, *huge_geom = (const float4 *)ws.huge_geom;
    const float *mid_w = (const float *)ws.mid_w, *huge_w = (const float *)ws.huge_w;
    TSP_HIP(hipEventRecord(ctx->ev[4], st_mid));
    for (long long o = 0; o < n_mid; o += mid_slice)
        if ((rc = launch_mid_gather(ctx, ta, MODE, second_channel, mid_geom + o, mid_w + o * NWr, std::min(mid_slice, n_mid - o), st_mid))) return rc;
    TSP_HIP(hipEventRecord(ctx->ev[5], st_mid));
    if (ctx->debug_fail_stage == 2) { ctx->debug_fail_stage = 0; TSP_REQUIRE(false, TSP_ENOMEM, "injected failure after kernel G (debug_fail_stage)"); }
    TSP_HIP(hipEventRecord(ctx->ev[9], st));
    TSP_HIP(hipEventRecord(ctx->ev[10], st));      // (kernel H2's launcher records it again after its launch)
    for (long long o = 0; o < n_huge; o += huge_slice)
        if ((rc = launch_gather_kernels(ctx, ta, MODE, second_channel, huge_geom + o, huge_w + o * NWr, std::min(huge_slice, n_huge - o)))) return rc;
    if (MODE == TSP_MODE_RGB && (n_mid > 0 || n_huge > 0)) {
        if (ctx->overlap_mid_huge) TSP_HIP(hipStreamWaitEvent(st, ctx->ev[5], 0));
        // (the rectangle counts are 32-bit: one pass while both lists are single slices, else one pass per slice)
        if (n_mid <= mid_slice && n_huge <= huge_slice) {
            if ((rc = add_rect_counts(ctx, mid_geom, n_mid, huge_geom, n_huge))) return rc;
        } else {
            for (long long o = 0; o < n_mid; o += mid_slice)
                if ((rc = add_rect_counts(ctx, mid_geom + o, std::min(mid_slice, n_mid - o), nullptr, 0))) return rc;
            for (long long o = 0; o < n_huge; o += huge_slice)
                if ((rc = add_rect_counts(ctx, nullptr, 0, huge_geom + o, std::min(huge_slice, n_huge - o)))) return rc;
        }
    }
    TSP_HIP(hipEventRecord(ctx->ev[6], st));
    if (ctx->overlap_mid_huge) TSP_HIP(hipStreamWaitEvent(st, ctx->ev[5], 0));     // join: later work on `st` sees both
    TSP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[cull ? 7 : 2], ctx->ev[3])); ctx->stats.ms_stream = ms;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); ctx->stats.ms_mid = ms;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[9], ctx->ev[10])); ctx->stats.ms_huge = ms;
    ctx->stats.ms_mega = 0.0;
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
