// Probe of the fp64 matrix pipe on MI355X: issue interval, dependent latency, sustained clock.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(1024) void mfma_loop(double* out, unsigned long long* stamps, int niter)
{
    d4 acc[U];
    for (int u = 0; u < U; ++u) acc[u] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < niter; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int u = 0; u < U; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

__global__ void tiny(unsigned long long* stamps, int spin)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
        stamps[2] = (unsigned long long)x;
    }
}

template <int U>
void run(int wgs, int threads, int niter)
{
    double* out;
    unsigned long long* st;
    hipMalloc(&out, (size_t)wgs * threads * 8);
    hipMalloc(&st, (size_t)wgs * 16);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(mfma_loop<U>, dim3(wgs), dim3(threads), 0, 0, out, st, niter);
    hipEventRecord(a);
    hipLaunchKernelGGL(mfma_loop<U>, dim3(wgs), dim3(threads), 0, 0, out, st, niter);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(2 * wgs);
    hipMemcpy(h.data(), st, (size_t)wgs * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < wgs; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= wgs; rt /= wgs;
    const double nm = (double)niter * U;
    const double waves = (double)wgs * threads / 64;
    const double flops = waves * nm * 2048.0;
    printf("U=%d wgs=%d thr=%d: %.3f ms  %.2f TFLOP/s  cycles/MFMA/wave=%.1f  clock=%.0f MHz\n", U, wgs,
           threads, ms, flops / ms / 1e9, cyc / nm, cyc / rt * 100.0);
    hipFree(out);
    hipFree(st);
}

int main()
{
    const int niter = 4000;
    printf("== one wave per SIMD (256 thr/WG, 1 WG per CU) ==\n");
    run<1>(256, 256, niter);
    run<2>(256, 256, niter);
    run<4>(256, 256, niter);
    run<8>(256, 256, niter);
    printf("== two waves per SIMD ==\n");
    run<1>(512, 256, niter);
    run<4>(512, 256, niter);
    printf("== four waves per SIMD ==\n");
    run<1>(1024, 256, niter);
    run<4>(1024, 256, niter);
    printf("== 512-thread WGs: 2, 4, 6, 8 waves per SIMD ==\n");
    run<4>(256, 512, niter);
    run<4>(512, 512, niter);
    run<4>(768, 512, niter);
    run<2>(1024, 512, niter);
    run<4>(1024, 512, niter);
    run<7>(512, 512, niter);
    run<7>(256, 1024, niter);
    printf("== long run (clock under sustained load) ==\n");
    run<4>(1024, 256, niter * 20);
    run<4>(1024, 512, niter * 10);
    // light load: tiny kernels back to back
    unsigned long long* st;
    hipMalloc(&st, 64);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, st, 2000);
        hipDeviceSynchronize();
        unsigned long long h[3];
        hipMemcpy(h, st, 24, hipMemcpyDeviceToHost);
        printf("tiny kernel chain: clock=%.0f MHz (cycles %llu, realtime ticks %llu)\n",
               (double)h[0] / h[1] * 100.0, h[0], h[1]);
    }
    return 0;
}
