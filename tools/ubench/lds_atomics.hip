// Microbenchmark: LDS atomic / store throughput on gfx950 (cycles per wave-instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP, int PATTERN>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ float buf[4096];
    __shared__ unsigned ibuf[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) { buf[i] = 0.f; ibuf[i] = 0u; }
    __syncthreads();
    int idx;
    if (PATTERN == 0) idx = tid;                       // distinct consecutive addresses
    else if (PATTERN == 1) idx = (tid >> 6) * 64;      // all lanes of a wave on one address
    else if (PATTERN == 2) idx = (tid >> 6) * 64 + (lane >> 2);   // 4 lanes per address
    else idx = (tid >> 6) * 64 + (lane >> 4);          // 16 lanes per address
    float v = 1.0f + lane * 1e-3f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = (idx + u * 256) & 4095;
            if (OP == 0) __hip_atomic_fetch_add(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 1) __hip_atomic_fetch_add(&ibuf[a], (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 2) { ((volatile float *)buf)[a] = v; }
            else if (OP == 3) { v += ((volatile float *)buf)[a]; }
            else if (OP == 4) { float o = __hip_atomic_fetch_add(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); v += o * 1e-9f; }
            else if (OP == 6) { double *p = (double *)ibuf; __hip_atomic_fetch_add(&p[a >> 1], (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else if (OP == 7) { double *p = (double *)ibuf; __hip_atomic_fetch_add(&p[(a >> 1) & ~31], (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else if (OP == 5) { unsigned long long *p = (unsigned long long *)ibuf; __hip_atomic_fetch_add(&p[a >> 1], (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (tid == 0) out[blockIdx.x] = (float)(t1 - t0) / (iters * 8.0f);
    if (v == 123.0f) out[0] = buf[tid] + ibuf[tid];
}

template <int OP, int PATTERN>
int run(const char *name, int blocks_per_cu) {
    float *d; CHECK(hipMalloc(&d, 4096 * sizeof(float)));
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(grid), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    float h[8]; CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    // wave-instructions per CU: blocks_per_cu * 4 waves * iters * 8
    const double winst_per_cu = (double)blocks_per_cu * 4 * iters * 8;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-34s blocks/CU %d: %7.1f clk per wave-instr per CU (wall, @2.4GHz)   [s_memtime/inst/wave %.1f]\n", name, blocks_per_cu, cyc / winst_per_cu, h[0]);
    hipFree(d); return 0;
}

int main() {
    for (int b = 2; b <= 2; ++b) {
        run<0, 0>("ds_add_f32 distinct", b);
        run<0, 1>("ds_add_f32 same addr (64)", b);
        run<0, 2>("ds_add_f32 4 lanes/addr", b);
        run<0, 3>("ds_add_f32 16 lanes/addr", b);
        run<1, 0>("ds_add_u32 distinct", b);
        run<1, 1>("ds_add_u32 same addr (64)", b);
        run<1, 2>("ds_add_u32 4 lanes/addr", b);
        run<5, 0>("ds_add_u64 distinct", b);
        run<6, 0>("ds_add_f64 distinct", b);
        run<6, 1>("ds_add_f64 same addr (64)", b);
        run<6, 2>("ds_add_f64 4 lanes/addr", b);
        run<5, 1>("ds_add_u64 same addr (64)", b);
        run<5, 2>("ds_add_u64 4 lanes/addr", b);
        run<2, 0>("ds_write_b32 distinct", b);
        run<3, 0>("ds_read_b32 distinct", b);
        run<4, 0>("ds_add_rtn_f32 distinct", b);
    }
    return 0;
}
