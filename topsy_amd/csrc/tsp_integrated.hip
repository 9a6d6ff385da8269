// tsp_integrated.hip -- kernel I (option `integrated_px`, off by default): bilinear footprints through their second differences.
//
// For P >= 64 px the reference samples level 0 of the kernel texture bilinearly (sph.wgsl:139-146), so a
// footprint's image is PIECEWISE BILINEAR in pixel coordinates: along either axis it is a continuous piecewise-linear
// function whose slope changes only where the texel coordinate crosses a texel centre -- 64 breakpoints per axis, P / 64
// pixels apart -- plus a jump at either edge of the square.  The mixed second difference  D2 = d2/dx2 d2/dy2 (image)  of such
// a function is therefore SPARSE: a slope change s at a non-integer position t contributes, to the second difference of
// the samples at integer pixels, s (c - t) at pixel c = ceil(t) and s (1 - (c - t)) at pixel c + 1, nothing elsewhere; a jump
// J at pixel e contributes (J, -J) at (e, e + 1).  In two dimensions the pairs of breakpoints of the two axes give 4 entries
// each, with strengths from ONE fixed 66 x 66 table  S0 = L T L^t  (T = the level-0 kernel image, L = [left jump; 64 second
// differences with clamp-to-edge ends; right jump]).  A footprint of any size is thus 4 x 66 x 66 = 17 k scattered adds
// instead of P^2 fragments (262 k at 512 px, 1 M at 1024 px), followed by ONE double prefix sum along either axis of the whole image.
//
// What it gives up (why it is an option, not the default): the adds cancel to the footprint's values only up to float64
// rounding, and the texel coordinates are exact instead of the float32-rounded ones of the canonical arithmetic
// (tsp_math.h), so a pixel differs from the oracle by ~1e-7 of the footprint's PEAK value (not of the pixel's own value),
// and what the adds leave where a footprint draws nothing is ~1e-10 of its peak, of either sign (the last prefix-sum kernel snaps
// everything below 1e-8 of the pass's largest peak contribution to the exact zero it stands for).  In a dense
// scene (every pixel under thousands of such footprints) that is < 1e-6 relative per pixel; where a pixel holds only the
// far tail of one footprint it is not within 1e-5 of the oracle.  tests/test_gpu_integrated.py states the contract.
//
// Layout: a 1024-thread workgroup owns a TILE of BH image rows x <= 1024 columns in LDS (float64; 17 rows at 1024^2) plus a ghost
// row / column on every side -- a breakpoint's two rows and two columns then always lie inside, and what falls on a ghost
// cell is the neighbouring tile's to add -- and a share of the mega records.  Per (footprint, tile) its 64 lanes ARE the 64
// x-breakpoints (column, two weights, computed once), and a scalar loop walks the y-breakpoints whose rows meet the band: one
// coalesced 528-byte row of S0 (the next one already in flight), 6 multiplies and 4 ds_add_f64 per lane and y-breakpoint.
// Breakpoints left of / above the viewport all land on columns / rows 0 and 1: their SUM is a linear function of prefix tables of
// S0 (PA, PB along x; PAy, PBy along y; M for the corner), formed by one lane / one virtual breakpoint -- no same-address
// pile-up, no 64-step walk in the top band.  The tile is then added to the global D2 image, and three small kernels integrate
// it twice along x and twice along y into the float64 render target.  Several channels (weighted, depth, rgb): one pass each
// (blockIdx.y), the channel's weight riding in the record's w.
#include "tsp_pipeline.h"
#include "tsp_math.h"

namespace tsp {

constexpr int IT = 1024;                 // threads per workgroup of kernel I (one workgroup per CU: the tile fills its LDS)
constexpr int IWAVES = IT / 64;
#ifndef TSP_INT_LDS_KB
#define TSP_INT_LDS_KB 156
#endif
constexpr size_t INT_LDS_BYTES = TSP_INT_LDS_KB * 1024;   // LDS of a tile: 17 + 2 rows of 1024 + 2 doubles (images wider than 1024 px are cut into column parts)

struct IntArgs {
    const float4 *geom;
    const float *wq;                     // per record: q (two-channel modes) or (g, b) / h^2 (rgb), as in the tile-gather kernels
    int wmode;                           // 0: density only, 1: second channel = w q, 2: rgb
    long long n_records;
    const double *T;                     // the breakpoint tables (integrated_tables)
    unsigned long long edge_lo, edge_hi; // bit q (of 66): row q of S0 has a non-zero edge jump (S0[q][0] or S0[q][65])
    double *D2;                          // [channels][R][R] second-difference images (blockIdx.y = channel)
    unsigned int *wmax;                  // [channels] largest |weight| of the pass (float bits): scales the zero threshold of the prefix sums
    Counters *cnt;
    int R, BH, STR, TW, nx, split, count_frag;   // tile = BH rows x TW columns (nx column parts per band), LDS row stride STR = TW + 2
    float p_lo;                          // records narrower than this belong to the matrix-core kernels
};

__device__ __forceinline__ void ladd64(double *addr, double v) {
    __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__global__ __launch_bounds__(IT) void splat_integrated_kernel(IntArgs a) {
    // rows by0 - 1 .. by0 + BH of the image: a breakpoint's two rows always lie inside (no per-row tests); the first and the last
    // are ghost rows that belong to the neighbouring bands (which add them themselves) and are not written back
    extern __shared__ __attribute__((aligned(16))) double itile[];     // [BH + 2][STR]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile_id = blockIdx.x / a.split, sp = blockIdx.x % a.split, ch = blockIdx.y;
    const int band = tile_id / a.nx, part = tile_id % a.nx;
    const int R = a.R, STR = a.STR;
    const int by0 = band * a.BH, by1 = min(by0 + a.BH, R), bh = by1 - by0;
    const int tx0 = part * a.TW, tx1 = min(tx0 + a.TW, R);   // the tile's columns; LDS also holds a ghost column either side
    for (int i = tid; i < (a.BH + 2) * STR; i += IT) itile[i] = 0.0;
    __syncthreads();
    unsigned long long n_frag = 0;
    bool touched = false;
    float wmax = 0.0f;

    // the workgroup's share of the records is dealt to its waves in runs of HDEAL, like the tile-gather kernels
    const long long vsplit = (long long)a.split * IWAVES, vsp = (long long)sp * IWAVES + wv;
    const long long n_runs = (a.n_records + HDEAL - 1) / HDEAL;
    for (long long run0 = 0; run0 * vsplit < n_runs; run0 += 64 / HDEAL) {
        const long long ri = ((run0 + lane / HDEAL) * vsplit + vsp) * HDEAL + (lane & (HDEAL - 1));
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ri < a.n_records) {
            g = a.geom[ri];
            // this workgroup's channel: the record's weight in it rides in g.w
            if (ch == 1) g.w = (a.wmode == 2) ? a.wq[ri * 2] : g.w * a.wq[ri];
            else if (ch == 2) g.w = a.wq[ri * 2 + 1];
        }
        int ilo = 1, ihi = 0, jlo = 1, jhi = 0;
        bool hit = false;
        if (g.z >= a.p_lo && g.z > 0.0f) {
            wmax = fmaxf(wmax, __builtin_fabsf(g.w));
            const float hf = 0.5f * g.z;
            cover_range(g.x, hf, R, ilo, ihi);
            cover_range(g.y, hf, R, jlo, jhi);
            // rows that receive entries: jlo .. jhi + 2 (the bottom jump sits on rows jhi + 1, jhi + 2)
            // ... and columns ilo .. ihi + 2
            hit = ilo <= ihi && jlo <= jhi && jlo < by1 && jhi + 2 >= by0 && ilo < tx1 && ihi + 2 >= tx0;
            if (hit && a.count_frag && ch == 0) {      // the fragments of this record inside this tile (every record meets every tile once)
                const int rows = min(jhi, by1 - 1) - max(jlo, by0) + 1, cols = min(ihi, tx1 - 1) - max(ilo, tx0) + 1;
                if (rows > 0 && cols > 0) n_frag += (unsigned long long)rows * (unsigned long long)cols;
            }
            if (hit && g.z > 64.0f * (float)(a.BH + 2)) {
                // breakpoints further apart than the band is tall: most bands hold none of them.  A breakpoint at t lands on rows
                // ceil(t), ceil(t) + 1; it concerns this band when by0 - 2 < t <= by1 - 1 (band 0: also everything above the image).
                // Conservative float32 test (the exact one follows per hit); the jumps sit on rows jlo and jhi + 1.
                const float st = g.z * (1.0f / 64.0f), t0 = 0.5f * st - hf - 0.5f + g.y;
                const float blo = by0 == 0 ? 0.0f : __builtin_floorf(((float)(by0 - 2) - t0) / st - 0.01f) + 1.0f;
                const float bhi = __builtin_floorf(((float)(by1 - 1) - t0) / st + 0.01f);
                const bool ramp = bhi >= fmaxf(blo, 0.0f) && blo <= 63.0f;
                const bool jump = (jlo >= by0 - 1 && jlo < by1) || (jhi + 1 >= by0 - 1 && jhi + 1 < by1);
                hit = ramp || jump;
            }
        }
        unsigned long long hits = __ballot(hit);
        while (hits) {
            const int src = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            touched = true;
            const float pcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), src));
            const float pcy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.y), src));
            const float Pf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.z), src));
            const float wf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.w), src));
            const int colL = __builtin_amdgcn_readlane(ilo, src), colR = __builtin_amdgcn_readlane(ihi, src) + 1;
            const int rowL = __builtin_amdgcn_readlane(jlo, src), rowR = __builtin_amdgcn_readlane(jhi, src) + 1;
            const double P = (double)Pf, w = (double)wf;
            const double step = P * (1.0 / 64.0);                        // pixels per texel
            double rp = __builtin_amdgcn_rcp(P);                        // texels per pixel: 64 / P by two Newton steps (P is a normal float32)
            rp = __builtin_fma(__builtin_fma(-P, rp, 1.0), rp, rp);
            rp = __builtin_fma(__builtin_fma(-P, rp, 1.0), rp, rp);
            const double alpha = 64.0 * rp;
            const double base = 0.5 * step - 0.5 * P - 0.5;
            const double t0x = base + (double)pcx, t0y = base + (double)pcy;   // pixel position of texel centre 0
            // ---- x breakpoints: lane a = slope change at texel centre a ----
            const double tx = __builtin_fma((double)lane, step, t0x);
            const double cxr = __builtin_ceil(tx);
            const bool xcl = cxr < 0.0;                                 // left of the viewport: lands on columns 0 / 1
            const double cxc = xcl ? 0.0 : (cxr > (double)R ? (double)R : cxr);
            const double fx = cxc - tx;
            int col = (int)cxc;
            const double A0 = alpha * fx, A1 = alpha - A0;
            const int a_c = __popcll(__ballot(xcl));                    // texel centres 0 .. a_c - 1 are left of the viewport
            // both of the lane's columns lie in the tile or its ghost columns (which the neighbouring tile fills itself)
            bool vx = col >= tx0 - 1 && col < tx1;
            if (xcl && lane != 0) vx = false;                           // lane 0 carries their sum (PA / PB below)
            col -= tx0 - 1;
            // ---- y breakpoints: lane b ----
            const double ty = __builtin_fma((double)lane, step, t0y);
            const double cyr = __builtin_ceil(ty);
            const double cyc = cyr < 0.0 ? 0.0 : (cyr > (double)R ? (double)R : cyr);
            const double fy = cyc - ty;
            const int row = (int)cyc;
            const double B0 = (alpha * w) * fy, B1 = alpha * w - B0;
            // y breakpoints above the viewport all land on rows 0 / 1: band 0 takes their SUM as one virtual breakpoint (prefix
            // tables PAy / PBy, corner table M) instead of walking up to 64 of them
            const unsigned long long yclm = __ballot(cyr < 0.0);
            const int b_c = __popcll(yclm);
            const unsigned long long qmask = __ballot(row >= by0 - 1 && row < by1 && row < R) & ~yclm;
            const int c2g = lane ? colR : colL;                         // lanes 0 / 1: the jump at the left / right edge
            const bool vs = c2g >= tx0 - 1 && c2g < tx1;
            const int c2 = c2g - (tx0 - 1);
            const bool allv = __ballot(vx) == ~0ull;

            // one y breakpoint: the lane's four entries (row rq: e00 at col, e10 at col + 1; row rq + 1: e01, e11) and,
            // on lanes 0 / 1, the edge jumps (u0s, -u0s on row rq; u1s, -u1s on row rq + 1)
            auto emit = [&](int rq, double e00, double e10, double e01, double e11, bool steps, double u0s, double u1s) {
                const int roff = (rq - by0 + 1) * STR;
                double *d = itile + roff + col;
                if (allv || vx) {                                       // (allv: the whole footprint width is in the tile, no lane mask)
                    ladd64(d, e00); ladd64(d + 1, e10); ladd64(d + STR, e01); ladd64(d + STR + 1, e11);
                }
                if (steps && lane < 2 && vs) {
                    double *e = itile + roff + c2;
                    ladd64(e, u0s); ladd64(e + 1, -u0s); ladd64(e + STR, u1s); ladd64(e + STR + 1, -u1s);
                }
            };
            if (b_c > 0 && by0 == 0) {
                const double *pay_r = a.T + INT_O_PAY + b_c * INT_S0_STRIDE, *pby_r = a.T + INT_O_PBY + b_c * INT_S0_STRIDE;
                const double pay = pay_r[1 + lane], pby = pby_r[1 + lane];
                double pas = 0.0, pbs = 0.0;
                if (lane < 2) { pas = pay_r[lane ? 65 : 0]; pbs = pby_r[lane ? 65 : 0]; }
                const double aw = alpha * w;
                const double my = t0y * pay + step * pby;
                const double Y0 = -(aw * my), Y1 = aw * (pay + my);
                double e00 = A0 * Y0, e10 = A1 * Y0, e01 = A0 * Y1, e11 = A1 * Y1;
                if (a_c > 0) {
                    const double *m = a.T + INT_O_M + (b_c * 65 + a_c) * 4;
                    const double M00 = m[0], M10 = m[1], M01 = m[2], M11 = m[3];
                    const double TA = t0x * M00 + step * M10, TB = t0y * M00 + step * M01;
                    const double TT = (t0x * t0y) * M00 + (t0x * step) * M01 + (t0y * step) * M10 + (step * step) * M11;
                    const double k = alpha * aw;
                    if (lane == 0) { e00 = k * TT; e10 = -(k * (TB + TT)); e01 = -(k * (TA + TT)); e11 = k * (((M00 + TA) + TB) + TT); }
                }
                const double mys = t0y * pas + step * pbs;
                const bool steps = __ballot(lane < 2 && (pas != 0.0 || pbs != 0.0)) != 0ull;
                emit(0, e00, e10, e01, e11, steps, -(aw * mys), aw * (pas + mys));
            }
            // Row q of S0 (lane a: the strength of the pair (q, a)) and one more value per lane: lanes 0 / 1 the edge jumps
            // S0[q][0], S0[q][65], lanes 2 / 3 the prefix sums PA / PB [q][a_c] of the breakpoints left of the viewport (all tables
            // share the row stride).  Both loads are unconditional, so that the next breakpoint's pair can be in flight while
            // this one is scattered (the compiler counts them: s_waitcnt vmcnt(2)).
            const int xoff = lane == 1 ? 65 : (lane == 2 ? INT_O_PA + a_c : (lane == 3 ? INT_O_PB + a_c : 0));
            auto load_q = [&](int q, double &s, double &x) {
                const double *S = a.T + q * INT_S0_STRIDE;
                s = S[1 + lane];
                x = S[xoff];
            };
            auto scatter_q = [&](int rq, double B0q, double B1q, double s, double x, bool steps) {
                const double U0 = s * B0q, U1 = s * B1q;
                double e00 = A0 * U0, e10 = A1 * U0, e01 = A0 * U1, e11 = A1 * U1;
                if (a_c > 0) {
                    const double pa = readlane_f64(x, 2), pb = readlane_f64(x, 3);
                    const double m = t0x * pa + step * pb;
                    const double X0 = -(alpha * m), X1 = alpha * (pa + m);
                    if (lane == 0) { e00 = X0 * B0q; e10 = X1 * B0q; e01 = X0 * B1q; e11 = X1 * B1q; }
                }
                emit(rq, e00, e10, e01, e11, steps, x * B0q, x * B1q);
            };
            // the jumps at the top / bottom edge of the square (one band each per footprint: not pipelined)
            if (rowL >= by0 - 1 && rowL < by1) { double s, x; load_q(0, s, x); scatter_q(rowL, w, -w, s, x, (a.edge_lo & 1ull) != 0ull); }
            if (rowR >= by0 - 1 && rowR < by1 && rowR < R) { double s, x; load_q(65, s, x); scatter_q(rowR, w, -w, s, x, (a.edge_hi & 2ull) != 0ull); }
            if (qmask) {
                unsigned long long rest = qmask;
                int b = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                double s, x;
                load_q(1 + b, s, x);
                for (;;) {
                    const int bn = rest ? __ffsll((long long)rest) - 1 : b;
                    double sn, xn;
                    load_q(1 + bn, sn, xn);
                    scatter_q(__builtin_amdgcn_readlane(row, b), readlane_f64(B0, b), readlane_f64(B1, b), s, x,
                              (((b < 63 ? a.edge_lo >> (1 + b) : a.edge_hi) & 1ull) != 0ull));
                    if (!rest) break;
                    rest &= rest - 1;
                    b = bn; s = sn; x = xn;
                }
            }
        }
    }
    const int any = __syncthreads_or(touched ? 1 : 0);
    if (any) {
        const int tw = tx1 - tx0;
        for (int i = tid; i < bh * tw; i += IT) {
            const int r = i / tw, c = i - r * tw;
            const double v = itile[(r + 1) * STR + c + 1];
            if (v != 0.0) gatomic_add(a.D2 + ((size_t)ch * R + (by0 + r)) * R + tx0 + c, v);
        }
    }
    if (a.count_frag) {
        for (int o = 32; o; o >>= 1) n_frag += __shfl_xor((long long)n_frag, o);
        if (lane == 0 && n_frag) { atomicAdd(&a.cnt->n_fragments, n_frag); atomicAdd(&a.cnt->n_frag_class[3], n_frag); }
    }
    if (tile_id == 0) {      // (every tile sees every record: one of them reports)
        for (int o = 32; o; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
        if (lane == 0 && wmax > 0.0f) atomicMax(&a.wmax[ch], __float_as_uint(wmax));     // non-negative floats order like their bits
    }
}

// ---- the double prefix sums -------------------------------------------------------------------------------------------------
// A run of L values with first sum a1 = sum(v) and second sum a2 = sum of the running first sums composes as
//   (L1, a1, a2) (+) (L2, b1, b2) = (L1 + L2, a1 + b1, a2 + b2 + L2 a1).

// along x, in place: one workgroup per image row
__global__ __launch_bounds__(256) void integrate_rows_kernel(double *__restrict__ D2, int R) {
    __shared__ double s1[256], s2[256];
    __shared__ int sl[256];
    const int tid = threadIdx.x;
    double *row = D2 + ((size_t)blockIdx.y * R + blockIdx.x) * R;
    const int per = (R + 255) / 256, b = min(tid * per, R), e = min(b + per, R);
    double r1 = 0.0, r2 = 0.0;
    for (int i = b; i < e; ++i) { r1 += row[i]; r2 += r1; }
    s1[tid] = r1; s2[tid] = r2; sl[tid] = e - b;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        double l1 = 0.0, l2 = 0.0; int ll = 0;
        if (tid >= o) { l1 = s1[tid - o]; l2 = s2[tid - o]; ll = sl[tid - o]; }
        __syncthreads();
        if (tid >= o) {
            const double m1 = s1[tid], m2 = s2[tid]; const int ml = sl[tid];
            s1[tid] = l1 + m1; s2[tid] = l2 + m2 + (double)ml * l1; sl[tid] = ll + ml;
        }
        __syncthreads();
    }
    double c1 = 0.0, c2 = 0.0;
    if (tid > 0) { c1 = s1[tid - 1]; c2 = s2[tid - 1]; }
    for (int i = b; i < e; ++i) { c1 += row[i]; c2 += c1; row[i] = c2; }
}

constexpr int ISEG = 32;                 // rows per segment of the column pass
// along y, pass 1: (a1, a2) of every (segment, column)
__global__ __launch_bounds__(256) void integrate_cols_partial_kernel(const double *__restrict__ D2, int R, double2 *__restrict__ part) {
    const int i = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y;
    if (i >= R) return;
    D2 += (size_t)blockIdx.z * R * R;
    part += (size_t)blockIdx.z * gridDim.y * R;
    const int j0 = seg * ISEG, j1 = min(j0 + ISEG, R);
    double r1 = 0.0, r2 = 0.0;
    for (int j = j0; j < j1; ++j) { r1 += D2[(size_t)j * R + i]; r2 += r1; }
    part[(size_t)seg * R + i] = make_double2(r1, r2);
}
// pass 2: carry in the segments above, finish, add into the channel of the render target and clear D2 for the next block
// What is left of the cancelling adds where no footprint (or only its exactly-zero corner) reaches is ~1e-10 of a peak
// contribution, of either sign: values below 1e-8 of the pass's largest peak contribution count as the exact zero they stand for
// (no negative dust in a density image: the reference's autorange picks the linear scale on ANY negative value).
// ONE thread takes a pixel column segment in every channel of the pass and decides the snap once per pixel: on the density
// channel for the two-channel modes (weighted, depth: channel 1 = density x quantity must vanish exactly where the density does,
// or the colormap's g / r would read +-inf there), on any channel for rgb.
template <int NCH>
__global__ __launch_bounds__(256) void integrate_cols_apply_kernel(double *__restrict__ D2, int R, int nseg, const double2 *__restrict__ part,
                                                                   double *__restrict__ img, int C, const unsigned int *__restrict__ wmax,
                                                                   float peak, bool density_decides) {
    const int i = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y;
    if (i >= R) return;
    double thr[NCH], c1[NCH], c2[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        thr[ch] = 1e-8 * (double)peak * (double)__uint_as_float(wmax[ch]);
        c1[ch] = c2[ch] = 0.0;
        const double2 *pc = part + (size_t)ch * nseg * R;
        for (int s = 0; s < seg; ++s) {
            const double2 p = pc[(size_t)s * R + i];
            const int len = min((s + 1) * ISEG, R) - s * ISEG;
            c2[ch] += p.y + (double)len * c1[ch];
            c1[ch] += p.x;
        }
    }
    const int j0 = seg * ISEG, j1 = min(j0 + ISEG, R);
    for (int j = j0; j < j1; ++j) {
        const size_t k = (size_t)j * R + i;
        bool keep = false;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            double *d = D2 + (size_t)ch * R * R + k;
            c1[ch] += *d; c2[ch] += c1[ch];
            *d = 0.0;
            if (ch == 0 || !density_decides) keep = keep || (__builtin_fabs(c2[ch]) > thr[ch]);
        }
        if (keep) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) gatomic_add(img + k * C + ch, c2[ch]);
        }
    }
}

bool integrated_supported(const tsp_context *ctx) {
    return ctx->int_tables != nullptr && ctx->R >= 64 && ctx->R <= 65536;
}

int launch_integrated(tsp_context *ctx, const TileArgs &ta, const float4 *geom, const float *wq, int wmode, long long n_records, float p_lo) {
    Workspace &ws = ctx->ws;
    hipStream_t st = ctx->stream;
    const int R = ctx->R;
    TSP_REQUIRE(integrated_supported(ctx), TSP_ESTATE, "kernel I is not available for this context");
    const int nch = wmode == 0 ? 1 : (wmode == 1 ? 2 : 3);
    const int nseg = (R + ISEG - 1) / ISEG;
    if (ws.int_dirty) integrated_release(ctx);      // an earlier pass was never seen to finish: its D2 cannot be trusted to be zero
    if (ws.int_channels < nch) {
        integrated_release(ctx);
        const size_t need = (size_t)nch * R * R * sizeof(double) + (size_t)nch * nseg * R * sizeof(double2);
        size_t free_b = 0, total_b = 0;
        TSP_HIP(hipMemGetInfo(&free_b, &total_b));
        TSP_REQUIRE(need < free_b, TSP_ENOMEM, "option integrated_px: the second-difference images of a %d-channel pass at %d^2 need %.1f GB, "
                    "%.1f GB are free", nch, R, need / 1e9, free_b / 1e9);
        TSP_HIP(hipMalloc((void **)&ws.int_d2, (size_t)nch * R * R * sizeof(double)));
        TSP_HIP(hipMalloc((void **)&ws.int_part, (size_t)nch * nseg * R * sizeof(double2)));
        TSP_HIP(hipMemsetAsync(ws.int_d2, 0, (size_t)nch * R * R * sizeof(double), st));
        ws.int_channels = nch;
    }
    if (!ws.int_wmax) TSP_HIP(hipMalloc((void **)&ws.int_wmax, 4 * sizeof(unsigned int)));
    TSP_HIP(hipMemsetAsync(ws.int_wmax, 0, 4 * sizeof(unsigned int), st));
    IntArgs ia;
    ia.geom = geom; ia.wq = wq; ia.wmode = wmode; ia.n_records = n_records;
    ia.T = ctx->int_tables; ia.p_lo = p_lo;
    ia.D2 = ws.int_d2; ia.wmax = ws.int_wmax; ia.cnt = ta.cnt; ia.R = R; ia.count_frag = ta.count_frag;
    // tiles of at most 1024 columns (+ a ghost column either side): 17 rows of them fit the LDS, and a footprint meets P / 17 + 1 bands
    ia.nx = (R + 1023) / 1024;
    ia.TW = (R + ia.nx - 1) / ia.nx;
    ia.STR = ia.TW + 2;
    const int bh = (int)(INT_LDS_BYTES / ((size_t)ia.STR * sizeof(double))) - 2;     // + two ghost rows
    ia.BH = bh;
    ia.edge_lo = ctx->int_edge[0]; ia.edge_hi = ctx->int_edge[1];
    const int n_bands = (R + bh - 1) / bh;
    const size_t smem = (size_t)(bh + 2) * ia.STR * sizeof(double);
    if (!(ctx->kernel_attr_done & (1u << 16))) {
        TSP_HIP(hipFuncSetAttribute((const void *)splat_integrated_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)INT_LDS_BYTES));
        ctx->kernel_attr_done |= 1u << 16;
    }
    // two workgroups per CU and channel-set: bands differ in work, and one round of 256 long workgroups measured 1.7x slower
    int split = ctx->mega_split;
    if (split <= 0) split = std::max(1, (ctx->cu_count * 2 + n_bands * ia.nx * nch - 1) / (n_bands * ia.nx * nch));
    const long long runs = (n_records + HDEAL - 1) / HDEAL;
    split = (int)std::min<long long>(split, std::max<long long>((runs + IWAVES - 1) / IWAVES, 1));
    ia.split = split;
    // From the scatter on, D2 is only all-zero again when the apply kernel has run: any failure in between drops the buffers
    // (the next pass allocates and clears fresh ones) instead of leaving stale second differences for every later frame.
    hipLaunchKernelGGL(splat_integrated_kernel, dim3(n_bands * ia.nx * split, nch), dim3(IT), smem, st, ia);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) {
        hipLaunchKernelGGL(integrate_rows_kernel, dim3(R, nch), dim3(256), 0, st, ws.int_d2, R);
        hipLaunchKernelGGL(integrate_cols_partial_kernel, dim3((R + 255) / 256, nseg, nch), dim3(256), 0, st, ws.int_d2, R, (double2 *)ws.int_part);
        const dim3 ag((R + 255) / 256, nseg), ab(256);
        const bool density_decides = wmode == 1;          // weighted / depth: channel 0 is the density
        if (nch == 1) hipLaunchKernelGGL(integrate_cols_apply_kernel<1>, ag, ab, 0, st, ws.int_d2, R, nseg, (const double2 *)ws.int_part, ctx->image64, ctx->C, ws.int_wmax, ctx->int_peak, density_decides);
        else if (nch == 2) hipLaunchKernelGGL(integrate_cols_apply_kernel<2>, ag, ab, 0, st, ws.int_d2, R, nseg, (const double2 *)ws.int_part, ctx->image64, ctx->C, ws.int_wmax, ctx->int_peak, density_decides);
        else hipLaunchKernelGGL(integrate_cols_apply_kernel<3>, ag, ab, 0, st, ws.int_d2, R, nseg, (const double2 *)ws.int_part, ctx->image64, ctx->C, ws.int_wmax, ctx->int_peak, density_decides);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        ws.int_dirty = true;
        set_error("kernel I launch failed: %s", hipGetErrorString(e));
        return TSP_EHIP;
    }
    ws.int_dirty = true;        // cleared by tsp_render once the stream has drained without error
    return TSP_OK;
}

// Free the second-difference images (option switched off, or their content is not known to be zero any more).
void integrated_release(tsp_context *ctx) {
    Workspace &ws = ctx->ws;
    if (ws.int_d2) (void)hipFree(ws.int_d2);
    if (ws.int_part) (void)hipFree(ws.int_part);
    ws.int_d2 = nullptr; ws.int_part = nullptr; ws.int_channels = 0; ws.int_dirty = false;
}

// The tables of kernel I, float64, one allocation (offsets INT_O_*):
//   S0  [66][68]   S0 = L T L^t over the 66 breakpoints of an axis (left jump, 64 slope changes with clamp-to-edge ends, right jump)
//   PA, PB [66][68]   prefix sums along x:  PA[q][n] = sum_{a < n} S0[q][1 + a],  PB[q][n] = sum_{a < n} a S0[q][1 + a]
//   PAy, PBy [65][68] prefix sums along y:  PAy[n][p] = sum_{b < n} S0[1 + b][p], PBy[n][p] = sum_{b < n} b S0[1 + b][p]
//   M   [65][65][4]   both: sums over b < n_b, a < n_a of S0[1 + b][1 + a] times (1, a, b, a b)
void integrated_tables(const float *mip0, std::vector<double> &out) {
    std::vector<double> L(66 * 64, 0.0), LT(66 * 64, 0.0);
    L[0] = 1.0;
    for (int a = 0; a < 64; ++a) {
        L[(1 + a) * 64 + std::max(a - 1, 0)] += 1.0;
        L[(1 + a) * 64 + a] -= 2.0;
        L[(1 + a) * 64 + std::min(a + 1, 63)] += 1.0;
    }
    L[65 * 64 + 63] = -1.0;
    for (int q = 0; q < 66; ++q)                    // LT = L T  (66 x 64); every sum has <= 3 float32 terms: exact in float64
        for (int x = 0; x < 64; ++x) {
            double s = 0.0;
            for (int k = 0; k < 64; ++k) if (L[q * 64 + k] != 0.0) s += L[q * 64 + k] * (double)mip0[k * 64 + x];
            LT[q * 64 + x] = s;
        }
    out.assign((size_t)INT_TABLE_DOUBLES, 0.0);
    double *S0 = out.data(), *PA = S0 + INT_O_PA, *PB = S0 + INT_O_PB, *PAy = S0 + INT_O_PAY, *PBy = S0 + INT_O_PBY, *M = S0 + INT_O_M;
    for (int q = 0; q < 66; ++q)
        for (int p = 0; p < 66; ++p) {
            double s = 0.0;
            for (int k = 0; k < 64; ++k) if (L[p * 64 + k] != 0.0) s += LT[q * 64 + k] * L[p * 64 + k];
            S0[q * INT_S0_STRIDE + p] = s;
        }
    for (int q = 0; q < 66; ++q) {
        double pa = 0.0, pb = 0.0;
        for (int n = 0; n <= 64; ++n) {
            PA[q * INT_S0_STRIDE + n] = pa; PB[q * INT_S0_STRIDE + n] = pb;
            if (n < 64) { pa += S0[q * INT_S0_STRIDE + 1 + n]; pb += (double)n * S0[q * INT_S0_STRIDE + 1 + n]; }
        }
    }
    for (int p = 0; p < 66; ++p) {
        double pa = 0.0, pb = 0.0;
        for (int n = 0; n <= 64; ++n) {
            PAy[n * INT_S0_STRIDE + p] = pa; PBy[n * INT_S0_STRIDE + p] = pb;
            if (n < 64) { pa += S0[(1 + n) * INT_S0_STRIDE + p]; pb += (double)n * S0[(1 + n) * INT_S0_STRIDE + p]; }
        }
    }
    for (int nb = 0; nb <= 64; ++nb)
        for (int na = 0; na <= 64; ++na) {
            double *m = M + (nb * 65 + na) * 4;
            if (nb == 0 || na == 0) { m[0] = m[1] = m[2] = m[3] = 0.0; continue; }
            // M[nb][na] = M[nb - 1][na] + (row nb - 1 of the x prefix sums)
            const double *up = M + ((nb - 1) * 65 + na) * 4;
            const double pa = PA[nb * INT_S0_STRIDE + na], pb = PB[nb * INT_S0_STRIDE + na];          // row q = 1 + (nb - 1) = nb
            m[0] = up[0] + pa; m[1] = up[1] + pb; m[2] = up[2] + (double)(nb - 1) * pa; m[3] = up[3] + (double)(nb - 1) * pb;
        }
}

}  // namespace tsp
