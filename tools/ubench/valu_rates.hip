// Microbenchmark: VALU issue rates on gfx950 (cycles per wave-instruction per SIMD) for scalar vs packed fp32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 m = {s, s * 0.5f}, c = {1e-3f, 2e-3f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) {          // 8 independent scalar fma
                a0 = __builtin_fmaf(a0, s, 1e-3f); a1 = __builtin_fmaf(a1, s, 1e-3f); a2 = __builtin_fmaf(a2, s, 1e-3f); a3 = __builtin_fmaf(a3, s, 1e-3f);
                a4 = __builtin_fmaf(a4, s, 1e-3f); a5 = __builtin_fmaf(a5, s, 1e-3f); a6 = __builtin_fmaf(a6, s, 1e-3f); a7 = __builtin_fmaf(a7, s, 1e-3f);
            } else if (OP == 1) {   // 8 independent packed fma
                p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c); p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
                p4 = __builtin_elementwise_fma(p4, m, c); p5 = __builtin_elementwise_fma(p5, m, c); p6 = __builtin_elementwise_fma(p6, m, c); p7 = __builtin_elementwise_fma(p7, m, c);
            } else if (OP == 2) {   // packed mul
                p0 = p0 * m; p1 = p1 * m; p2 = p2 * m; p3 = p3 * m; p4 = p4 * m; p5 = p5 * m; p6 = p6 * m; p7 = p7 * m;
            } else if (OP == 3) {   // scalar mul
                a0 *= s; a1 *= s; a2 *= s; a3 *= s; a4 *= s; a5 *= s; a6 *= s; a7 *= s;
            } else if (OP == 4) {   // packed add
                p0 = p0 + c; p1 = p1 + c; p2 = p2 + c; p3 = p3 + c; p4 = p4 + c; p5 = p5 + c; p6 = p6 + c; p7 = p7 + c;
            } else if (OP == 5) {   // readlane feeding a scalar-operand fma
                const float r = __builtin_amdgcn_readlane(a7, u);
                a0 = __builtin_fmaf(a0, r, 1e-3f); a1 = __builtin_fmaf(a1, r, 1e-3f); a2 = __builtin_fmaf(a2, r, 1e-3f); a3 = __builtin_fmaf(a3, r, 1e-3f);
                a4 = __builtin_fmaf(a4, r, 1e-3f); a5 = __builtin_fmaf(a5, r, 1e-3f); a6 = __builtin_fmaf(a6, r, 1e-3f);
            } else if (OP == 7) {   // packed fma, three distinct varying register-pair operands
                p0 = __builtin_elementwise_fma(p1, p2, p0); p1 = __builtin_elementwise_fma(p2, p3, p1); p2 = __builtin_elementwise_fma(p3, p4, p2); p3 = __builtin_elementwise_fma(p4, p5, p3);
                p4 = __builtin_elementwise_fma(p5, p6, p4); p5 = __builtin_elementwise_fma(p6, p7, p5); p6 = __builtin_elementwise_fma(p7, p0, p6); p7 = __builtin_elementwise_fma(p0, p1, p7);
            } else if (OP == 8) {   // scalar fma, three distinct varying operands
                a0 = __builtin_fmaf(a1, a2, a0); a1 = __builtin_fmaf(a2, a3, a1); a2 = __builtin_fmaf(a3, a4, a2); a3 = __builtin_fmaf(a4, a5, a3);
                a4 = __builtin_fmaf(a5, a6, a4); a5 = __builtin_fmaf(a6, a7, a5); a6 = __builtin_fmaf(a7, a0, a6); a7 = __builtin_fmaf(a0, a1, a7);
            } else if (OP == 9) {   // packed mul, two distinct varying operands
                p0 = p1 * p2; p1 = p2 * p3; p2 = p3 * p4; p3 = p4 * p5; p4 = p5 * p6; p5 = p6 * p7; p6 = p7 * p0; p7 = p0 * p1;
            } else if (OP == 10) {  // packed fma with a splat (op_sel) middle operand taken from a varying scalar VGPR
                p0 = __builtin_elementwise_fma(p1, (f2){a0, a0}, p0); p1 = __builtin_elementwise_fma(p2, (f2){a1, a1}, p1); p2 = __builtin_elementwise_fma(p3, (f2){a2, a2}, p2); p3 = __builtin_elementwise_fma(p4, (f2){a3, a3}, p3);
                p4 = __builtin_elementwise_fma(p5, (f2){a4, a4}, p4); p5 = __builtin_elementwise_fma(p6, (f2){a5, a5}, p5); p6 = __builtin_elementwise_fma(p7, (f2){a6, a6}, p6); p7 = __builtin_elementwise_fma(p0, (f2){a7, a7}, p7);
            } else if (OP == 11) {  // packed add, two distinct varying operands
                p0 = p1 + p2; p1 = p2 + p3; p2 = p3 + p4; p3 = p4 + p5; p4 = p5 + p6; p5 = p6 + p7; p6 = p7 + p0; p7 = p0 + p1;
            } else if (OP == 6) {   // floor + cvt + med3 (transcendental-free "slow?" ops)
                a0 = __builtin_floorf(a0); a1 = (float)(int)a1; a2 = __builtin_amdgcn_fmed3f(a2, 0.f, s); a3 = __builtin_floorf(a3);
                a4 = (float)(int)a4; a5 = __builtin_amdgcn_fmed3f(a5, 0.f, s); a6 = __builtin_floorf(a6); a7 = __builtin_amdgcn_fmed3f(a7, 0.f, s);
            }
        }
    }
    long long t1 = clock64();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 123.456f) out[1] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 64.0f);
}

template <int OP>
int run(const char *name, int waves_per_simd) {
    float *d; CHECK(hipMalloc(&d, 64));
    const int iters = 4000, grid = 256 * waves_per_simd;     // 256-thread blocks: 1 wave per SIMD each
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, 10, 0.999f);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(b); CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    float h[2]; CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const double winst_per_simd = (double)waves_per_simd * iters * 64;
    printf("%-28s waves/SIMD %d: %6.2f clk per wave-instr per SIMD (wall @2.4GHz)  [clock64/inst in-wave %.2f]\n", name, waves_per_simd,
           ms * 1e-3 * 2.4e9 / winst_per_simd, h[0]);
    hipFree(d); return 0;
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<3>("v_mul_f32", w); run<2>("v_pk_mul_f32", w); run<4>("v_pk_add_f32", w);
        run<5>("readlane + 7 fma(sgpr)", w); run<8>("v_fma_f32 3 distinct", w); run<7>("v_pk_fma_f32 3 distinct", w); run<10>("v_pk_fma_f32 splat operand", w);
        run<9>("v_pk_mul_f32 2 distinct", w); run<11>("v_pk_add_f32 2 distinct", w);
    }
    return 0;
}
