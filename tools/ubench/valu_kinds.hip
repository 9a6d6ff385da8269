// Microbenchmark: issue cost of the instruction kinds in kernel H's inner loop on gfx950, with 1/2/4 waves per SIMD.
// Each kernel body is inline asm on 8 independent register sets so the compiler cannot fold anything.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a[8], b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 0.5f + i * 0.01f; c[i] = 0.25f * i; }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 1) asm volatile("v_fma_f32 %0, %1, 4.0, -0.5" : "=v"(a[i]) : "v"(b[i]));
                else if (OP == 2) asm volatile("v_med3_f32 %0, %1, 0, 1.0" : "=v"(a[i]) : "v"(b[i]));
                else if (OP == 3) asm volatile("v_cmp_lt_f32 vcc, |%1|, %2\n\tv_cndmask_b32 %0, 0, 1.0, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc");
                else if (OP == 4) asm volatile("v_floor_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                else if (OP == 5) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                else if (OP == 6) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 7) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 8) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 9) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                else if (OP == 11) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "s"(s), "v"(c[i]));
                else if (OP == 12) asm volatile("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 13) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc");
                else if (OP == 14) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(b[i]), "v"(c[i]) : "vcc");
                else if (OP == 15) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]), "v"(b[(i + 1) & 7]));
                else if (OP == 16) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 18) asm volatile("v_mul_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 19) asm volatile("v_mul_f32_dpp %0, %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 20) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 21) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]));
                else if (OP == 22) asm volatile("v_add_u32_dpp %0, %1, %2 row_ror:12 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                else if (OP == 23) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(c[i]));
                else if (OP == 24) { int t; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t) : "v"(b[i])); asm volatile("" :: "s"(t)); }
                else if (OP == 25) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\ts_bitcmp0_b32 %3, 5" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]), "s"(iters) : "scc");
                else if (OP == 17) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
            }
        }
    }
    long long t1 = clock64();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    if (r == 123.456f) out[1] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 64.0f);
}

template <int OP>
int run(const char *name, int per_item) {
    float *d; CHECK(hipMalloc(&d, 64));
    printf("%-34s", name);
    for (int w = 1; w <= 4; w *= 2) {
        const int iters = 4000, grid = 256 * w;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, 10, 0.999f);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, iters, 0.999f);
        (void)hipEventRecord(b); CHECK(hipDeviceSynchronize());
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("  %dw/SIMD %5.2f", w, ms * 1e-3 * 2.4e9 / ((double)w * iters * 64 * per_item));
    }
    printf("   clk per instruction per SIMD\n");
    (void)hipFree(d); return 0;
}

int main() {
    run<0>("v_fma_f32 v,v,v (acc)", 1); run<15>("v_fma_f32 v,v,v,v", 1); run<1>("v_fma_f32 v, 4.0, -0.5 (inline)", 1); run<2>("v_med3_f32 v, 0, 1.0 (inline)", 1);
    run<3>("v_cmp + v_cndmask (2 instrs)", 2); run<14>("v_cmp_lt_f32 vcc", 1); run<13>("v_cndmask_b32 vcc", 1);
    run<4>("v_floor_f32", 1); run<5>("v_cvt_i32_f32", 1); run<6>("v_add_u32", 1); run<12>("v_lshl_add_u32", 1);
    run<7>("v_mul_f32 v,v", 1); run<11>("v_mul_f32 s,v", 1); run<8>("v_fmac_f32", 1); run<9>("v_sub_f32", 1); run<16>("v_add_f32", 1); run<17>("v_max_f32", 1);
    run<10>("v_mov_b32", 1);
    run<18>("v_mul_f32_dpp quad_perm", 1); run<19>("v_mul_f32_dpp row_ror:4", 1); run<20>("v_fmac_f32_dpp quad_perm", 1); run<21>("v_mov_b32_dpp row_ror:8", 1); run<22>("v_add_u32_dpp row_ror:12", 1);
    run<23>("v_fmac_f32 s,v", 1); run<24>("v_readlane_b32", 1); run<25>("v_fmac_dpp + s_bitcmp0 (per pair)", 1);
    return 0;
}
