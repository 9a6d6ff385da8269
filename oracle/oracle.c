/* CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates topsy's SPH splat + colormap hot path on the CPU so the HIP kernels can be checked
 * against it at sizes the numpy twin (oracle_np.py) cannot reach, and serves as the timed
 * `cpu_baseline` ("port") in bench.py.  The product (topsy_amd/) never links, loads or calls it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks it against every known-answer
 * vector of the reference's own tests (tests/golden/reference_kats.npz) and bit-for-bit against
 * oracle_np.py.
 *
 * Reference lines followed (relative to /root/reference):
 *   vertex stage ............ src/topsy/shaders/sph.wgsl:54-83   (orc_project)
 *   rasteriser + fragment ... src/topsy/shaders/sph.wgsl:139-146,161-165 ; blend src/topsy/sph.py:31-42
 *   kernel texture sampler .. src/topsy/sph.py:396-426 (4 mips, mag linear / min+mip nearest)
 *   colormap ................ src/topsy/shaders/colormap.wgsl:75-159
 *   range blocks ............ src/topsy/particle_buffers.py:70-82 (first_instance, instance_count)
 *
 * The canonical float32 operation order is documented at the top of oracle_np.py; this file must
 * be compiled with -ffp-contract=off and without -ffast-math.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define P_BILINEAR 64.0f
#define P_L0 45.254833995939045f
#define P_L1 22.627416997969522f
#define P_L2 11.313708498984761f

static const int MIP_N[4] = {64, 32, 16, 8};
static const int MIP_OFF[4] = {0, 4096, 5120, 5376};

typedef struct {
    float pcx, pcy, cz, P, half, invP;
    int keep;
} proj_t;

/* sph.wgsl:54-66 in pixel units */
static inline proj_t orc_project(const float *M, float sf, float Rf, float x, float y, float z, float h) {
    proj_t r;
    float cx = ((M[0] * x + M[1] * y) + M[2] * z) + M[3];
    float cy = ((M[4] * x + M[5] * y) + M[6] * z) + M[7];
    r.cz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11];
    float s = (sf * h) * 2.0f;
    float halfR = 0.5f * Rf;
    r.P = s * Rf;
    r.half = 0.5f * r.P;
    r.pcx = (cx + 1.0f) * halfR;
    r.pcy = (1.0f - cy) * halfR;
    r.invP = 1.0f / r.P;
    r.keep = (r.cz >= 0.0f) && (r.cz <= 1.0f) && isfinite(r.P) && (r.P > 0.0f) && isfinite(r.pcx) && isfinite(r.pcy);
    return r;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* candidate pixel interval (conservative), the exact test is |d| < half */
static inline void cand(float pc, float half, int R, int *lo, int *hi) {
    float a = floorf(pc - half - 0.5f) - 1.0f;
    float b = ceilf(pc + half - 0.5f) + 1.0f;
    if (a < 0.0f) a = 0.0f;
    if (b > (float)(R - 1)) b = (float)(R - 1);
    *lo = (int)a;
    *hi = (int)b;
}

static inline int level_for(float P) {
    if (P >= P_BILINEAR) return -1;
    if (P > P_L0) return 0;
    if (P > P_L1) return 1;
    if (P > P_L2) return 2;
    return 3;
}

/* One axis of texture addressing, precomputed per covered pixel column/row. */
typedef struct {
    int i0, i1;
    float f;
} axis_t;

static inline axis_t axis_bilinear_n(float u, int n) {
    axis_t a;
    float tu = u * (float)n - 0.5f;
    float x0 = floorf(tu);
    a.f = tu - x0;
    a.i0 = clampi((int)x0, 0, n - 1);
    a.i1 = clampi((int)x0 + 1, 0, n - 1);
    return a;
}
static inline axis_t axis_bilinear(float u) { return axis_bilinear_n(u, 64); }

static inline int axis_nearest(float u, int n) { return clampi((int)floorf(u * (float)n), 0, n - 1); }

/* Sampling rule.  0 = "O1", the rule that reproduces all of the reference's golden
 * vectors: bilinear on mip 0 when P >= 64, else the NEAREST texel of the mip chosen by the rounded LOD.
 * The two alternatives are kept for diagnosis against other WebGPU drivers: 1 = bilinear on mip 0 always,
 * 2 = bilinear within the chosen mip.
 *
 * Work decomposition (round 4): the image is cut into TS x TS pixel tiles; every tile owns the list of the
 * particles whose candidate rectangle reaches it, in drawing order, and ONE thread adds them into a tile-sized
 * double accumulator that stays in its cache.  A pixel therefore receives its terms in exactly the order a
 * single thread would add them: the result does not depend on the thread count, and the working set per
 * thread is 16-32 KiB instead of a whole image (the per-thread images of rounds 1-3 made 256 threads slower
 * than 128).  The per-fragment float32 arithmetic is untouched; texture addressing of a pixel column / row is
 * computed once per (particle, tile) instead of once per pixel -- the same operations on the same operands. */
#define TS 32

typedef struct { int i0, i1; float f, g; int tx; int covered; } col_t;

/* Add the part of one particle's footprint that lies in the tile [x0, x1] x [y0, y1] (inclusive pixel bounds)
 * to the tile accumulator `acc` (TS * TS * C doubles, row stride TS).  nch: C = 2 (w[0]=m/h^2, w[1]=q), 4 = rgb + count. */
static void splat_tile(double *acc, int x0, int x1, int y0, int y1, int R, int C, const float *mips, proj_t pr,
                       const float *w, long *nfrag, int rule) {
    int ilo, ihi, jlo, jhi;
    cand(pr.pcx, pr.half, R, &ilo, &ihi);
    cand(pr.pcy, pr.half, R, &jlo, &jhi);
    if (ilo < x0) ilo = x0;
    if (ihi > x1) ihi = x1;
    if (jlo < y0) jlo = y0;
    if (jhi > y1) jhi = y1;
    if (ihi < ilo || jhi < jlo) return;
    int lvl = level_for(pr.P);
    int bn = 64, boff = 0;                  /* mip sampled bilinearly when lvl < 0 */
    if (rule == 1) lvl = -1;
    else if (rule == 2 && lvl >= 0) { bn = MIP_N[lvl]; boff = MIP_OFF[lvl]; lvl = -1; }
    col_t cols[TS];
    int any = 0;
    for (int i = ilo; i <= ihi; ++i) {
        col_t *c = &cols[i - x0];
        float dx = ((float)i + 0.5f) - pr.pcx;
        c->covered = fabsf(dx) < pr.half;
        if (!c->covered) continue;
        any = 1;
        float u = (dx + pr.half) * pr.invP;
        if (lvl < 0) {
            axis_t ax = axis_bilinear_n(u, bn);
            c->i0 = ax.i0; c->i1 = ax.i1; c->f = ax.f; c->g = 1.0f - ax.f;
        } else {
            c->tx = axis_nearest(u, MIP_N[lvl]);
        }
    }
    if (!any) return;
    long nf = 0;
    for (int j = jlo; j <= jhi; ++j) {
        float dy = ((float)j + 0.5f) - pr.pcy;
        if (!(fabsf(dy) < pr.half)) continue;
        float v = (dy + pr.half) * pr.invP;
        double *row = acc + (size_t)(j - y0) * TS * C;
        const float *T0 = 0, *T1 = 0;
        float fy = 0.f, gy = 0.f;
        if (lvl < 0) {
            axis_t ay = axis_bilinear_n(v, bn);
            T0 = mips + boff + ay.i0 * bn;
            T1 = mips + boff + ay.i1 * bn;
            fy = ay.f; gy = 1.0f - ay.f;
        } else {
            int n = MIP_N[lvl];
            T0 = mips + MIP_OFF[lvl] + axis_nearest(v, n) * n;
        }
        for (int i = ilo; i <= ihi; ++i) {
            const col_t *c = &cols[i - x0];
            if (!c->covered) continue;
            float k;
            if (lvl < 0) {
                float top = T0[c->i0] * c->g + T0[c->i1] * c->f;
                float bot = T1[c->i0] * c->g + T1[c->i1] * c->f;
                k = top * gy + bot * fy;
            } else {
                k = T0[c->tx];
            }
            double *px = row + (size_t)(i - x0) * C;
            if (C == 2) {
                float val = k * w[0];
                px[0] += val;
                px[1] += (float)(val * w[1]);
            } else {
                px[0] += (float)(k * w[0]);
                px[1] += (float)(k * w[1]);
                px[2] += (float)(k * w[2]);
                px[3] += 1.0;
            }
            ++nf;
        }
    }
    *nfrag += nf;
}

static inline void particle_weights(float *w, int mode, proj_t pr, float hp, const float *a, const float *b,
                                    const float *c, int64_t p) {
    float hh = hp * hp;
    w[0] = w[1] = w[2] = 0.f;
    if (mode == 2) {
        w[0] = a[p] / hh; w[1] = b[p] / hh; w[2] = c[p] / hh;
    } else {
        w[0] = a[p] / hh;
        w[1] = (mode == 1) ? pr.cz : (b ? b[p] : 0.0f);
    }
}

/* mode: 0 = mass + quantity (vertex_weighting), 1 = depth (vertex_depth: w[1] = clip z), 2 = rgb.
 * a,b,c: mode 0: a=mass, b=qty (may be NULL -> 0), c unused; mode 1: a=mass; mode 2: a,b,c = r,g,b.
 * starts/lens: particle index ranges (NULL -> all), drawn in the order given (particle_buffers.py:70-82).
 * out: R*R*C float32 (C = 2 or 4).  accumulate != 0 adds to `out` instead of overwriting.
 * Returns the total fragment count. */
#ifndef ORC_BATCH
#define ORC_BATCH 50000000      /* most particles per batch of the tile-parallel splat */
#endif
#ifndef ORC_ENTRY_BUDGET
#define ORC_ENTRY_BUDGET 200000000LL   /* most (particle, tile) entries per batch (8 bytes each): a batch ends early when wide footprints fill it */
#endif

long orc_splat_rule(long n, const float *x, const float *y, const float *z, const float *h,
                    const float *a, const float *b, const float *c, int mode,
                    const float *M, float sf, int R, const float *mips,
                    const int64_t *starts, const int64_t *lens, int nranges,
                    int accumulate, int nthreads, float *out, int rule) {
    const int C = (mode == 2) ? 4 : 2;
    int64_t s0 = 0, l0 = n;
    if (!starts) { starts = &s0; lens = &l0; nranges = 1; }
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    const float Rf = (float)R;
    const int nt = (R + TS - 1) / TS;                      /* tiles per image side */
    /* the drawing sequence: ranges in the order given, clipped to [0, n) */
    int64_t *rbeg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nranges + 1) * 2);
    if (!rbeg) return -1;
    int64_t *roff = rbeg + nranges + 1;
    int64_t nsel = 0;
    for (int r = 0; r < nranges; ++r) {
        int64_t beg = starts[r], end = starts[r] + lens[r];
        if (beg < 0) beg = 0;
        if (end > n) end = n;
        if (end < beg) end = beg;
        rbeg[r] = beg;
        roff[r] = nsel;
        nsel += end - beg;
    }
    roff[nranges] = nsel;
    /* The drawing sequence is processed in BATCHES of at most ORC_BATCH particles and ORC_ENTRY_BUDGET (particle, tile) entries
     * (a 1000-px footprint spans ~1000 tiles of 32 x 32 px: the entry list of a batch stays bounded whatever the footprints).  Every tile's terms are still added in
     * drawing order into ONE float64 accumulator image that lives across the batches, so the sequence of float64 additions per
     * pixel -- and with it the result, bit for bit -- is that of a single pass (and independent of the thread count).
     * A failed allocation returns -1 (the Python wrapper raises MemoryError). */
    typedef struct { uint16_t x0, x1, y0, y1; } box_t;
    const int64_t batch_cap = nsel < ORC_BATCH ? (nsel > 0 ? nsel : 1) : ORC_BATCH;
    box_t *box = (box_t *)malloc(sizeof(box_t) * (size_t)batch_cap);
    int64_t *tcount = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nt * nt + 1));
    int64_t *tstart = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nt * nt + 1));
    double *accimg = (double *)calloc((size_t)R * R * C, sizeof(double));
    int64_t *entries = NULL;
    int64_t entries_cap = 0;
    long total_frag = 0;
    int failed = (!rbeg || !box || !tcount || !tstart || !accimg);
    int64_t nb = 0;
    for (int64_t f0 = 0; f0 < nsel && !failed; f0 += nb) {
        nb = (nsel - f0 < batch_cap) ? nsel - f0 : batch_cap;
        /* pass A: tile rectangle of every particle of the batch (x0 > x1 = draws nothing) */
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int64_t k = 0; k < nb; ++k) {
            const int64_t f = f0 + k;
            int r = 0;                                           /* range of sequence position f (binary search) */
            { int lo = 0, hi = nranges - 1; while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (roff[mid] <= f) lo = mid; else hi = mid - 1; } r = lo; }
            const int64_t p = rbeg[r] + (f - roff[r]);
            box_t q = {1, 0, 1, 0};
            proj_t pr = orc_project(M, sf, Rf, x[p], y[p], z[p], h[p]);
            if (pr.keep) {
                int ilo, ihi, jlo, jhi;
                cand(pr.pcx, pr.half, R, &ilo, &ihi);
                cand(pr.pcy, pr.half, R, &jlo, &jhi);
                if (ihi >= ilo && jhi >= jlo) {
                    q.x0 = (uint16_t)(ilo / TS); q.x1 = (uint16_t)(ihi / TS);
                    q.y0 = (uint16_t)(jlo / TS); q.y1 = (uint16_t)(jhi / TS);
                }
            }
            box[k] = q;
        }
        {   /* the batch ends where its entry list would exceed the budget (at least one particle) */
            int64_t e = 0, k = 0;
            for (; k < nb; ++k) {
                const box_t q = box[k];
                if (q.x0 <= q.x1) e += (int64_t)(q.x1 - q.x0 + 1) * (q.y1 - q.y0 + 1);
                if (e > ORC_ENTRY_BUDGET && k > 0) break;
            }
            nb = k;
        }
        /* pass B: per tile, the positions (in the drawing sequence) of the batch's particles that reach it, ascending.
         * One thread per tile ROW scans the boxes twice (count, fill): no atomics, order preserved. */
        memset(tcount, 0, sizeof(int64_t) * ((size_t)nt * nt + 1));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
        for (int ty = 0; ty < nt; ++ty) {
            int64_t *cn = tcount + (size_t)ty * nt;
            for (int64_t k = 0; k < nb; ++k) {
                const box_t q = box[k];
                if (q.x0 > q.x1 || ty < q.y0 || ty > q.y1) continue;
                for (int tx = q.x0; tx <= q.x1; ++tx) ++cn[tx];
            }
        }
        int64_t total = 0;
        for (int t = 0; t < nt * nt; ++t) { tstart[t] = total; total += tcount[t]; }
        tstart[nt * nt] = total;
        if (total > entries_cap) {
            free(entries);
            entries_cap = total + total / 4 + 1024;
            entries = (int64_t *)malloc(sizeof(int64_t) * (size_t)entries_cap);
            if (!entries) { failed = 1; break; }
        }
        memcpy(tcount, tstart, sizeof(int64_t) * (size_t)nt * nt);      /* reused as the fill cursors */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
        for (int ty = 0; ty < nt; ++ty) {
            int64_t *cur = tcount + (size_t)ty * nt;
            for (int64_t k = 0; k < nb; ++k) {
                const box_t q = box[k];
                if (q.x0 > q.x1 || ty < q.y0 || ty > q.y1) continue;
                for (int tx = q.x0; tx <= q.x1; ++tx) entries[cur[tx]++] = f0 + k;
            }
        }
        /* pass C: one thread per tile, terms added in drawing order into the tile's part of the float64 accumulator image */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total_frag)
        for (int t = 0; t < nt * nt; ++t) {
            if (tstart[t] == tstart[t + 1]) continue;
            const int ty = t / nt, tx = t % nt;
            const int x0 = tx * TS, y0 = ty * TS;
            const int x1 = (x0 + TS - 1 < R - 1) ? x0 + TS - 1 : R - 1;
            const int y1 = (y0 + TS - 1 < R - 1) ? y0 + TS - 1 : R - 1;
            double acc[TS * TS * 4];
            for (int j = y0; j <= y1; ++j)
                for (int i = x0; i <= x1; ++i)
                    for (int k = 0; k < C; ++k)
                        acc[((size_t)(j - y0) * TS + (i - x0)) * C + k] = accimg[((size_t)j * R + i) * C + k];
            long nf = 0;
            int r = 0;
            for (int64_t e = tstart[t]; e < tstart[t + 1]; ++e) {
                const int64_t f = entries[e];
                while (f >= roff[r + 1]) ++r;                    /* entries ascend: the range index only moves forward */
                const int64_t p = rbeg[r] + (f - roff[r]);
                proj_t pr = orc_project(M, sf, Rf, x[p], y[p], z[p], h[p]);
                float w[3];
                particle_weights(w, mode, pr, h[p], a, b, c, p);
                splat_tile(acc, x0, x1, y0, y1, R, C, mips, pr, w, &nf, rule);
            }
            total_frag += nf;
            for (int j = y0; j <= y1; ++j)
                for (int i = x0; i <= x1; ++i)
                    for (int k = 0; k < C; ++k)
                        accimg[((size_t)j * R + i) * C + k] = acc[((size_t)(j - y0) * TS + (i - x0)) * C + k];
        }
    }
    if (!failed) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int64_t o = 0; o < (int64_t)R * R * C; ++o) out[o] = accumulate ? (float)((double)out[o] + accimg[o]) : (float)accimg[o];
    }
    free(entries);
    free(accimg);
    free(tstart);
    free(tcount);
    free(box);
    free(rbeg);
    return failed ? -1 : total_frag;
}

/* the reference's sampling (rule "O1") */
long orc_splat(long n, const float *x, const float *y, const float *z, const float *h,
               const float *a, const float *b, const float *c, int mode,
               const float *M, float sf, int R, const float *mips,
               const int64_t *starts, const int64_t *lens, int nranges,
               int accumulate, int nthreads, float *out) {
    return orc_splat_rule(n, x, y, z, h, a, b, c, mode, M, sf, R, mips, starts, lens, nranges, accumulate, nthreads, out, 0);
}

/* ------------------------------------------------------------------------------------------
 * canonical float32 log / exp (see oracle_np.py) and the colormap (colormap.wgsl:75-159)
 * ------------------------------------------------------------------------------------------ */
static inline float as_f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t as_i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

float orc_logf(float x) {
    if (x != x || x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (isinf(x)) return INFINITY;
    int eadj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; eadj = -23; }
    int32_t bits = as_i(x);
    int e = ((bits >> 23) & 0xff) - 127 + eadj;
    float m = as_f((bits & 0x007fffff) | 0x3f800000);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = 0.18181819f;
    p = p * z + 0.22222222f;
    p = p * z + 0.2857143f;
    p = p * z + 0.4f;
    p = p * z + 0.6666667f;
    float lm = (s + s) + (s * z) * p;
    float ef = (float)e;
    return (ef * 0.693359375f + lm) + ef * -2.12194440e-4f;
}

float orc_expf(float y) {
    if (y != y) return NAN;
    if (y > 88.72f) return INFINITY;
    if (y < -103.9f) return 0.0f;
    float nf = floorf(y * 1.44269504f + 0.5f);
    float r = (y - nf * 0.693359375f) - nf * -2.12194440e-4f;
    float p = 1.9841270e-4f;
    p = p * r + 1.3888889e-3f;
    p = p * r + 8.3333333e-3f;
    p = p * r + 4.1666667e-2f;
    p = p * r + 0.16666667f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    int n = (int)nf;
    int n1 = clampi(n, -126, 127);
    int n2 = clampi(n - n1, -126, 127);
    return (p * as_f((n1 + 127) << 23)) * as_f((n2 + 127) << 23);
}

static inline float orc_log10f(float x) { return orc_logf(x) / 2.30258509f; }

float orc_powf(float x, float g) {
    if (g == 1.0f) return x;
    if (x == 0.0f && g > 0.0f) return 0.0f;
    if (g == 0.0f && x == x) return 1.0f;
    return orc_expf(g * orc_logf(x));
}

static inline uint8_t unorm8(float c) {
    if (c != c) c = 0.0f;
    if (c < 0.0f) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)floorf(c * 255.0f + 0.5f);
}

/* fragment_main, non-bivariate (colormap.wgsl:113-127).  img: npix x C floats (C >= 2). */
void orc_colormap_scalar(const float *img, long npix, int C, const float *lut, int nlut,
                         float vmin, float vmax, int log_scale, int weighted, uint8_t *out) {
    float range = vmax - vmin;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p) {
        float v = weighted ? img[p * C + 1] / img[p * C] : img[p * C];
        if (log_scale) v = orc_log10f(v);
        float t = (v - vmin) / range;
        if (t != t) t = 0.0f;
        if (t < 0.0f) t = 0.0f;
        if (t > 1.0f) t = 1.0f;
        float c = t * (float)nlut - 0.5f;
        float c0 = floorf(c);
        float f = c - c0;
        int i0 = clampi((int)c0, 0, nlut - 1), i1 = clampi((int)c0 + 1, 0, nlut - 1);
        float g = 1.0f - f;
        for (int k = 0; k < 4; ++k) out[p * 4 + k] = unorm8(lut[i0 * 4 + k] * g + lut[i1 * 4 + k] * f);
    }
}

/* fragment_main_tri + gamma_map (colormap.wgsl:131-159), LOG_SCALE on.  outf (optional) gets the
 * unclamped float RGBA (HDR path), out8 (optional) the unorm8 store. */
void orc_colormap_rgb(const float *img, long npix, int C, float vmin, float vmax, float gamma,
                      uint8_t *out8, float *outf) {
    float range = vmax - vmin;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p) {
        for (int k = 0; k < 3; ++k) {
            float v = orc_log10f(img[p * C + k]);
            float xx = (v - vmin) / range;
            if (xx != xx) xx = 0.0f;
            if (xx < 0.0f) xx = 0.0f;
            float c = orc_powf(xx, gamma);
            if (out8) out8[p * 4 + k] = unorm8(c);
            if (outf) outf[p * 4 + k] = c;
        }
        if (out8) out8[p * 4 + 3] = 255;
        if (outf) outf[p * 4 + 3] = 1.0f;
    }
}

/* fragment_main, BIVARIATE branch (colormap.wgsl:91-111).  lut2d: n x n x RGBA, [y (value)][x (density)]. */
void orc_colormap_bivariate(const float *img, long npix, int C, const float *lut2d, int n, float vmin, float vmax,
                            float dvmin, float dvmax, int log_scale, int weighted, uint8_t *out) {
    const float range = vmax - vmin, drange = dvmax - dvmin;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p) {
        float x = (orc_log10f(img[p * C]) - dvmin) / drange;
        float y = weighted ? img[p * C + 1] / img[p * C] : img[p * C];
        if (log_scale) y = orc_log10f(y);
        y = (y - vmin) / range;
        if (x != x || x < 0.0f) x = 0.0f;
        if (x > 1.0f) x = 1.0f;
        if (y != y || y < 0.0f) y = 0.0f;
        if (y > 1.0f) y = 1.0f;
        float cx = x * (float)n - 0.5f, cy = y * (float)n - 0.5f;
        float x0 = floorf(cx), y0 = floorf(cy);
        float fx = cx - x0, fy = cy - y0, gx = 1.0f - fx, gy = 1.0f - fy;
        int i0 = clampi((int)x0, 0, n - 1), i1 = clampi((int)x0 + 1, 0, n - 1);
        int j0 = clampi((int)y0, 0, n - 1), j1 = clampi((int)y0 + 1, 0, n - 1);
        for (int k = 0; k < 4; ++k) {
            float top = lut2d[((size_t)j0 * n + i0) * 4 + k] * gx + lut2d[((size_t)j0 * n + i1) * 4 + k] * fx;
            float bot = lut2d[((size_t)j1 * n + i0) * 4 + k] * gx + lut2d[((size_t)j1 * n + i1) * 4 + k] * fx;
            out[p * 4 + k] = unorm8(top * gy + bot * fy);
        }
    }
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
