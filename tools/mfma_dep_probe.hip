// What does a dependency cost on the fp64 MFMA pipe of gfx950 (tools only)?
// One workgroup of 1 / 2 / 3 waves per SIMD runs a loop of v_mfma_f64_16x16x4 in several dependency
// patterns; prints core cycles per MFMA and per SIMD (64 = the pipe is always busy).
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_dep_probe.hip -o tools/bin/mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d4 mf(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(768) void probe(double* out, long long* cyc, long long* wcyc, int iters, double* gbuf_all = nullptr)
{
    double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-4, a1 = a + 0.5, a2 = a - 0.25;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned k0 = threadIdx.x, k1 = 77;
    double f0 = a, f1 = b;
    __shared__ double lds[2 * 768];
    double* gbuf = gbuf_all ? gbuf_all + (size_t)blockIdx.x * 3 * 64 * 16 * 4096 : nullptr;   // [3 groups of 256 threads][64][16][4096]
    d2v ld_acc = {0.0, 0.0};
    __syncthreads();
    const long long w0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // 16 MFMAs, four accumulators round robin
#pragma unroll
            for (int i = 0; i < 4; ++i) { c0 = mf(a, b, c0); c1 = mf(a, b, c1); c2 = mf(a, b, c2); c3 = mf(a, b, c3); }
        } else if constexpr (MODE == 1) {   // 16 MFMAs on one accumulator
#pragma unroll
            for (int i = 0; i < 16; ++i) c0 = mf(a, b, c0);
        } else if constexpr (MODE == 2) {   // four chains of four, one after the other
#pragma unroll
            for (int i = 0; i < 4; ++i) c0 = mf(a, b, c0);
#pragma unroll
            for (int i = 0; i < 4; ++i) c1 = mf(a, b, c1);
#pragma unroll
            for (int i = 0; i < 4; ++i) c2 = mf(a, b, c2);
#pragma unroll
            for (int i = 0; i < 4; ++i) c3 = mf(a, b, c3);
        } else if constexpr (MODE == 3) {   // stage-1 shape: 12 first products, VALU combine, 4 dependent seconds
            d4 x0 = {0, 0, 0, 0}, x1 = x0, x2 = x0;
#pragma unroll
            for (int i = 0; i < 4; ++i) x0 = mf(a, b, x0);
#pragma unroll
            for (int i = 0; i < 4; ++i) x1 = mf(a1, b, x1);
#pragma unroll
            for (int i = 0; i < 4; ++i) x2 = mf(a2, b, x2);
            d4 xt;
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[e] = x0[e] + 0.5 * x1[e] + 0.25 * x2[e];
#pragma unroll
            for (int i = 0; i < 4; ++i) c0 = mf(a, xt[i], c0);
        } else if constexpr (MODE == 4) {   // as 3, the second products of the PREVIOUS iteration among the first
            d4 x0 = {0, 0, 0, 0}, x1 = x0, x2 = x0;
            d4 xp = c1;                       // last iteration's combination
#pragma unroll
            for (int i = 0; i < 4; ++i) { x0 = mf(a, b, x0); x1 = mf(a1, b, x1); x2 = mf(a2, b, x2); c0 = mf(a, xp[i], c0); }
#pragma unroll
            for (int e = 0; e < 4; ++e) c1[e] = x0[e] + 0.5 * x1[e] + 0.25 * x2[e];
        } else if constexpr (MODE == 6) {   // independent integer VALU work in the shadow of each MFMA
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i % 4 == 0) c0 = mf(a, b, c0); else if (i % 4 == 1) c1 = mf(a, b, c1); else if (i % 4 == 2) c2 = mf(a, b, c2); else c3 = mf(a, b, c3);
                asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(k0) : "v"(k1));
            }
        } else if constexpr (MODE == 7) {   // independent fp64 VALU work in the shadow of each MFMA
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i % 4 == 0) c0 = mf(a, b, c0); else if (i % 4 == 1) c1 = mf(a, b, c1); else if (i % 4 == 2) c2 = mf(a, b, c2); else c3 = mf(a, b, c3);
                asm volatile("v_add_f64 %0, %0, %1\n\tv_add_f64 %2, %2, %1" : "+v"(f0), "+v"(f1) : "v"(a1));
            }
        } else if constexpr (MODE == 8) {   // four independent integer VALU ops per MFMA
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i % 4 == 0) c0 = mf(a, b, c0); else if (i % 4 == 1) c1 = mf(a, b, c1); else if (i % 4 == 2) c2 = mf(a, b, c2); else c3 = mf(a, b, c3);
                asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(k0) : "v"(k1));
            }
        } else if constexpr (MODE == 9) {   // two LDS stores per MFMA
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i % 4 == 0) c0 = mf(a, b, c0); else if (i % 4 == 1) c1 = mf(a, b, c1); else if (i % 4 == 2) c2 = mf(a, b, c2); else c3 = mf(a, b, c3);
                lds[threadIdx.x] = a1; lds[threadIdx.x + 768] = a2;
            }
        } else if constexpr (MODE == 10) {  // one 16-byte-per-lane global store (4 rows of 256 B) per MFMA
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i % 4 == 0) c0 = mf(a, b, c0); else if (i % 4 == 1) c1 = mf(a, b, c1); else if (i % 4 == 2) c2 = mf(a, b, c2); else c3 = mf(a, b, c3);
                d2v* dst = reinterpret_cast<d2v*>(gbuf + ((size_t)(threadIdx.x >> 8) * 1024 + (size_t)(it & 63) * 16 + i) * 4096 + ((threadIdx.x & 255) >> 4) * 256 + (threadIdx.x & 15) * 2);
                __builtin_nontemporal_store(d2v{a1, a2}, dst);
            }
        } else if constexpr (MODE == 11) {  // one 16-byte-per-lane global load per MFMA (consumed at the end)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i % 4 == 0) c0 = mf(a, b, c0); else if (i % 4 == 1) c1 = mf(a, b, c1); else if (i % 4 == 2) c2 = mf(a, b, c2); else c3 = mf(a, b, c3);
                const d2v v = *reinterpret_cast<const d2v*>(gbuf + ((size_t)(threadIdx.x >> 8) * 1024 + (size_t)(it & 63) * 16 + i) * 4096 + ((threadIdx.x & 255) >> 4) * 256 + (threadIdx.x & 15) * 2);
                ld_acc.x += v.x; ld_acc.y += v.y;
            }
        } else if constexpr (MODE == 5) {   // MFMA result straight into a VALU op and back, one at a time
#pragma unroll
            for (int i = 0; i < 16; ++i) { c0 = mf(a, b, c0); b += c0[0]; }
        }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(a1), "+v"(a2));
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) wcyc[blockIdx.x * 12 + 11] = w1 - w0;     // 100 MHz
    d4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + k0 + f0 + f1 + lds[(threadIdx.x * 7) % 1536] + ld_acc.x + ld_acc.y;
    // the last wave to finish counts (the oldest wave of a SIMD wins the pipe whenever it is ready)
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)&cyc[blockIdx.x], (unsigned long long)(t1 - t0));
    if ((threadIdx.x & 63) == 0) wcyc[blockIdx.x * 12 + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
static void run(const char* what, double* out, long long* cyc, long long* wcyc, double* gbuf = nullptr)
{
    const int iters = 2000;
    for (int wps = 1; wps <= 3; ++wps)
        for (int grid : {1, 256}) {
            hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256 * wps), 0, 0, out, cyc, wcyc, iters, gbuf);
            (void)hipMemset(cyc, 0, 256 * sizeof(long long));
            hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256 * wps), 0, 0, out, cyc, wcyc, iters, gbuf);
            (void)hipDeviceSynchronize();
            long long h[256], w[12];
            (void)hipMemcpy(w, wcyc, sizeof(w), hipMemcpyDeviceToHost);
            (void)hipMemcpy(h, cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
            double mean = 0;
            for (int i = 0; i < grid; ++i) mean += (double)h[i] / grid;
            printf("%-58s %d wave(s)/SIMD, %3d workgroup(s): %6.1f cycles per MFMA of the SIMD\n", what, wps, grid,
                   mean / ((double)iters * 16 * wps));
            if (grid == 1 && wps > 1) {
                printf("      waves of SIMD 0 finish after");
                for (int k = 0; k < wps; ++k) printf(" %.0f", (double)w[4 * k] / iters);
                printf(" cycles per iteration\n");
            }
        }
}

int main()
{
    double* out;
    long long *cyc, *wcyc;
    (void)hipMalloc(&out, 256 * 768 * sizeof(double));
    (void)hipMalloc(&cyc, 256 * sizeof(long long));
    (void)hipMalloc(&wcyc, 256 * 12 * sizeof(long long));
    // core clock under a sustained fp64 MFMA load: cycle counter against the 100 MHz wall clock
    for (int grid : {1, 32, 256}) {
        const int iters = 100000;
        (void)hipMemset(cyc, 0, 256 * sizeof(long long));
        hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, out, cyc, wcyc, iters);
        (void)hipDeviceSynchronize();
        long long w[12], c;
        (void)hipMemcpy(w, wcyc, sizeof(w), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("%3d workgroups of 8 waves, MFMA back to back for %.1f ms: %.0f core cycles per microsecond\n", grid,
               w[11] / 1e5, (double)c / (w[11] / 100.0));
    }
    run<0>("four accumulators round robin", out, cyc, wcyc);
    run<1>("one accumulator", out, cyc, wcyc);
    run<2>("four chains of four in sequence", out, cyc, wcyc);
    run<3>("12 first products -> VALU combine -> 4 second products", out, cyc, wcyc);
    run<4>("same, second products of the previous slab interleaved", out, cyc, wcyc);
    run<5>("MFMA -> VALU -> MFMA, one at a time", out, cyc, wcyc);
    run<6>("two independent integer VALU ops after each MFMA", out, cyc, wcyc);
    run<8>("four independent integer VALU ops after each MFMA", out, cyc, wcyc);
    run<7>("two independent v_add_f64 after each MFMA", out, cyc, wcyc);
    run<9>("two LDS stores after each MFMA", out, cyc, wcyc);
    double* gbuf;
    (void)hipMalloc(&gbuf, (size_t)256 * 3 * 64 * 16 * 4096 * sizeof(double));
    (void)hipMemset(gbuf, 0, (size_t)256 * 3 * 64 * 16 * 4096 * sizeof(double));
    run<10>("one global store (16 B per lane, 4 rows of 256 B) per MFMA", out, cyc, wcyc, gbuf);
    run<11>("one global load (16 B per lane) per MFMA", out, cyc, wcyc, gbuf);
    return 0;
}
