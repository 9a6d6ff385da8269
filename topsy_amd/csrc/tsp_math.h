// tsp_math.h -- device-side arithmetic of the splat + colormap path (gfx950).
//
// The semantics follow the reference's shaders: vertex stage src/topsy/shaders/sph.wgsl:54-83,
// fragment stage :139-146,161-165, sampler src/topsy/sph.py:396-426, colormap
// src/topsy/shaders/colormap.wgsl:75-159.  Coverage and nearest-texel decisions use ONE fixed
// float32 operation order (documented in DESIGN.md "canonical arithmetic") so that discrete
// decisions are reproducible; this translation unit must be built with -ffp-contract=off, and
// fused multiply-adds appear only where written explicitly (__builtin_fmaf) in value-only code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsp {

constexpr float P_BILINEAR = 64.0f;               // P >= 64 px  -> LOD <= 0 -> mag filter (bilinear, mip 0)
constexpr float P_L0 = 45.254833995939045f;       // 64 / 2^0.5 : P >  -> nearest, mip 0
constexpr float P_L1 = 22.627416997969522f;       // 64 / 2^1.5 : P >  -> mip 1
constexpr float P_L2 = 11.313708498984761f;       // 64 / 2^2.5 : P >  -> mip 2, else mip 3
constexpr int MIP_TOTAL = 5440;                   // 64^2 + 32^2 + 16^2 + 8^2

struct Camera {          // passed by value to kernels (lives in SGPRs / kernarg)
    float m[12];         // first three rows of the row-major 4x4: clip = M * (x,y,z,1)
    float sf;            // 1/scale
    float Rf;            // resolution as float
    float halfR;         // 0.5 * R
    int R;
};

struct Proj {            // a particle in pixel units
    float pcx, pcy;      // centre: column / row coordinate (pixel i centre = i + 0.5)
    float P;             // quad width in pixels
    float half;          // 0.5 * P
    float invP;          // 1 / P
    float cz;            // clip-space z
    bool keep;
};

template <bool WITH_INVP = true>
__device__ __forceinline__ Proj project(const Camera &c, float x, float y, float z, float h) {
    Proj r;
    const float cx = ((c.m[0] * x + c.m[1] * y) + c.m[2] * z) + c.m[3];
    const float cy = ((c.m[4] * x + c.m[5] * y) + c.m[6] * z) + c.m[7];
    r.cz = ((c.m[8] * x + c.m[9] * y) + c.m[10] * z) + c.m[11];
    const float s = (c.sf * h) * 2.0f;               // sph.wgsl:58  scale_factor * pos.w * 2
    r.P = s * c.Rf;
    r.half = 0.5f * r.P;
    r.pcx = (cx + 1.0f) * c.halfR;
    r.pcy = (1.0f - cy) * c.halfR;
    r.invP = WITH_INVP ? 1.0f / r.P : 0.0f;      // (kernel S forms it only in the waves that rasterise)
    // fixed-function clip: 0 <= z <= 1; non-finite geometry draws nothing
    r.keep = (r.cz >= 0.0f) && (r.cz <= 1.0f) && (r.P > 0.0f) && (r.P < __builtin_inff()) &&
             (__builtin_fabsf(r.pcx) < __builtin_inff()) && (__builtin_fabsf(r.pcy) < __builtin_inff());
    return r;
}

// Exact interval [lo, hi] of covered pixels along one axis, clipped to the image; lo > hi when empty.
// Coverage is the canonical test |(i + 0.5) - pc| < half evaluated in float32; it holds on a contiguous
// run of pixels whose ends lie within one pixel of the real-arithmetic bounds, so three candidate
// tests per side decide it without loops or branches.
__device__ __forceinline__ bool covers(float pc, float half, float i) {
    return __builtin_fabsf((i + 0.5f) - pc) < half;
}
__device__ __forceinline__ void cover_range(float pc, float half, int R, int &lo, int &hi) {
    const float l0 = __builtin_floorf(pc - half - 0.5f) + 1.0f;    // first covered pixel in real arithmetic
    const float h0 = __builtin_ceilf(pc + half - 0.5f) - 1.0f;     // last covered pixel in real arithmetic
    float l = covers(pc, half, l0 - 1.0f) ? l0 - 1.0f : (covers(pc, half, l0) ? l0 : l0 + 1.0f);
    float h = covers(pc, half, h0 + 1.0f) ? h0 + 1.0f : (covers(pc, half, h0) ? h0 : h0 - 1.0f);
    // clip to [0, R-1] while still in float (the bounds may be far outside the int range)
    l = l < 0.0f ? 0.0f : l;
    const float rm = (float)(R - 1);
    h = h > rm ? rm : h;
    // an empty or fully clipped run must come out as lo > hi
    const bool empty = !(l <= h);
    lo = empty ? 1 : (int)l;
    hi = empty ? 0 : (int)h;
}

__device__ __forceinline__ int level_for(float P) {   // -1 = bilinear on mip 0
    return P >= P_BILINEAR ? -1 : (P > P_L0 ? 0 : (P > P_L1 ? 1 : (P > P_L2 ? 2 : 3)));
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ int nearest_index(float u, int n) {
    return clampi((int)__builtin_floorf(u * (float)n), 0, n - 1);
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libtopsy_splat is written for gfx950 (MI355X) only: its inline assembly uses that target's instructions (v_cvt_flr_i32_f32, DPP, ds_add_f64)"
#endif
// clampi((int)floorf(x), 0, hi) in two instructions instead of four: v_cvt_flr_i32_f32 converts with floor rounding (and saturates
// like the plain conversion), v_med3_i32 clamps.  The same integers for every x; `hi` wave-uniform (_s) or per lane (_v).
__device__ __forceinline__ int floor_clamp_s(float x, int hi) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1\n\tv_med3_i32 %0, %0, 0, %2" : "=&v"(r) : "v"(x), "s"(hi));
    return r;
}
__device__ __forceinline__ int floor_clamp_v(float x, int hi) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1\n\tv_med3_i32 %0, %0, 0, %2" : "=&v"(r) : "v"(x), "v"(hi));
    return r;
}

__device__ __forceinline__ int mip_offset(int lvl) {  // 0, 4096, 5120, 5376
    return lvl == 0 ? 0 : (lvl == 1 ? 4096 : (lvl == 2 ? 5120 : 5376));
}

struct AxisB {  // bilinear addressing along one axis
    int i0, i1;
    float f;
};
__device__ __forceinline__ AxisB axis_bilinear(float u) {
    AxisB a;
    const float tu = u * 64.0f - 0.5f;
    const float x0 = __builtin_floorf(tu);
    a.f = tu - x0;
    const int ix = (int)x0;
    a.i0 = clampi(ix, 0, 63);
    a.i1 = clampi(ix + 1, 0, 63);
    return a;
}

// Kernel sample for pixel offset (dx, dy) from the particle centre.  T = mip pyramid (LDS or global).
template <typename LUT>
__device__ __forceinline__ float sample_kernel(const LUT &T, const Proj &p, int lvl, float dx, float dy) {
    const float u = (dx + p.half) * p.invP;
    const float v = (dy + p.half) * p.invP;
    if (lvl < 0) {
        const AxisB ax = axis_bilinear(u), ay = axis_bilinear(v);
        const float gx = 1.0f - ax.f, gy = 1.0f - ay.f;
        const float top = T[ay.i0 * 64 + ax.i0] * gx + T[ay.i0 * 64 + ax.i1] * ax.f;
        const float bot = T[ay.i1 * 64 + ax.i0] * gx + T[ay.i1 * 64 + ax.i1] * ax.f;
        return top * gy + bot * ay.f;
    }
    const int n = 64 >> lvl;
    return T[mip_offset(lvl) + nearest_index(v, n) * n + nearest_index(u, n)];
}

// The same with a selectable sampling rule (TSP_SAMPLE_*): 0 = "O1" as above; 1 = bilinear on
// mip 0 whatever the footprint width; 2 = bilinear within the mip the rounded LOD selects.  Diagnostic only.
template <typename LUT>
__device__ __forceinline__ float sample_kernel_rule(const LUT &T, const Proj &p, float dx, float dy, int rule) {
    int lvl = level_for(p.P);
    if (rule == 0 || (rule != 1 && lvl < 0)) return sample_kernel(T, p, lvl, dx, dy);
    if (rule == 1) return sample_kernel(T, p, -1, dx, dy);
    const int n = 64 >> lvl, off = mip_offset(lvl);
    const float tu = ((dx + p.half) * p.invP) * (float)n - 0.5f, tv = ((dy + p.half) * p.invP) * (float)n - 0.5f;
    const float x0 = __builtin_floorf(tu), y0 = __builtin_floorf(tv);
    const float fx = tu - x0, fy = tv - y0, gx = 1.0f - fx, gy = 1.0f - fy;
    const int ix0 = clampi((int)x0, 0, n - 1), ix1 = clampi((int)x0 + 1, 0, n - 1);
    const int iy0 = clampi((int)y0, 0, n - 1), iy1 = clampi((int)y0 + 1, 0, n - 1);
    const float top = T[off + iy0 * n + ix0] * gx + T[off + iy0 * n + ix1] * fx;
    const float bot = T[off + iy1 * n + ix0] * gx + T[off + iy1 * n + ix1] * fx;
    return top * gy + bot * fy;
}

// ------------------------------------------------------------------------------------------
// canonical float32 log / exp / pow (colormap).  WGSL leaves log()/pow() precision
// implementation-defined; the path fixes one algorithm so the uint8 image is reproducible.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float canon_logf(float x) {
    if (x != x || x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int eadj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; eadj = -23; }
    const int bits = __float_as_int(x);
    int e = ((bits >> 23) & 0xff) - 127 + eadj;
    float m = __int_as_float((bits & 0x007fffff) | 0x3f800000);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = 0.18181819f;
    p = p * z + 0.22222222f;
    p = p * z + 0.2857143f;
    p = p * z + 0.4f;
    p = p * z + 0.6666667f;
    const float lm = (s + s) + (s * z) * p;
    const float ef = (float)e;
    return (ef * 0.693359375f + lm) + ef * -2.12194440e-4f;
}

__device__ __forceinline__ float canon_expf(float y) {
    if (y != y) return y;
    if (y > 88.72f) return __builtin_inff();
    if (y < -103.9f) return 0.0f;
    const float nf = __builtin_floorf(y * 1.44269504f + 0.5f);
    const float r = (y - nf * 0.693359375f) - nf * -2.12194440e-4f;
    float p = 1.9841270e-4f;
    p = p * r + 1.3888889e-3f;
    p = p * r + 8.3333333e-3f;
    p = p * r + 4.1666667e-2f;
    p = p * r + 0.16666667f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    const int n = (int)nf;
    const int n1 = clampi(n, -126, 127);
    const int n2 = clampi(n - n1, -126, 127);
    return (p * __int_as_float((n1 + 127) << 23)) * __int_as_float((n2 + 127) << 23);
}

__device__ __forceinline__ float canon_log10f(float x) { return canon_logf(x) / 2.30258509f; }  // colormap.wgsl:75-77

__device__ __forceinline__ float canon_powf(float x, float g) {
    if (g == 1.0f) return x;
    if (x == 0.0f && g > 0.0f) return 0.0f;
    if (g == 0.0f && x == x) return 1.0f;
    return canon_expf(g * canon_logf(x));
}

__device__ __forceinline__ uint32_t unorm8(float c) {   // rgba8unorm store: round(255 * clamp(c))
    c = (c != c) ? 0.0f : c;
    c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
    return (uint32_t)__builtin_floorf(c * 255.0f + 0.5f);
}

}  // namespace tsp
