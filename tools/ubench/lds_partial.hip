// Microbenchmark: ds_add_f64 cost vs number of active lanes and vs 2-way address conflicts (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int active, int ways) {
    __shared__ double buf[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int idx = (tid >> 6) * 64 + lane / ways;        // `ways` lanes share an address
    const double v = 1.0 + lane * 1e-3;
    float acc = 0.f;
    if (lane < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int a = (idx + u * 256) & 4095;
                if (MODE == 0) __hip_atomic_fetch_add(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 1) { buf[a] += v; }                               // non-atomic read-modify-write (wave-private data)
                else if (MODE == 2) { acc += ((volatile float *)buf)[a]; }          // ds_read_b32
                else if (MODE == 3) { float *fb = (float *)buf; fb[a] += (float)v; } // non-atomic f32 RMW
            }
        }
    }
    __syncthreads();
    if (acc == 123.f) out[1] = acc + (float)buf[tid];
}

template <int MODE>
int run(const char *name, int active, int ways) {
    float *d; CHECK(hipMalloc(&d, 64));
    const int iters = 2000, bpc = 2, grid = 256 * bpc;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, d, 10, active, ways);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, d, iters, active, ways);
    (void)hipEventRecord(b); CHECK(hipDeviceSynchronize());
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-28s active lanes %2d, %d lane(s)/address: %7.1f clk per wave-instr per CU\n", name, active, ways,
           ms * 1e-3 * 2.4e9 / ((double)bpc * 4 * iters * 8));
    (void)hipFree(d); return 0;
}

int main() {
    for (int act : {64, 32, 16, 8, 4, 1}) run<0>("ds_add_f64", act, 1);
    for (int w : {2, 4, 8, 16, 64}) run<0>("ds_add_f64", 64, w);
    run<1>("f64 read+add+write", 64, 1); run<3>("f32 read+add+write", 64, 1); run<2>("ds_read_b32", 64, 1);
    return 0;
}
