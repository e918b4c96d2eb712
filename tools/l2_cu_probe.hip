// Per-CU read bandwidth from L2 / Infinity Cache for a small resident buffer (tools only).
// W workgroups of T threads stream the same S-byte buffer R times with 16-byte loads.
// Question it answers: how fast can ONE workgroup re-read a 331 x 331 fp64 matrix (876 KB)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ void stream(const d2* __restrict__ buf, long n2, int reps, double* out)
{
    double acc = 0.0;
    const int T = blockDim.x;
    for (int r = 0; r < reps; ++r) {
        long i = threadIdx.x;
        for (; i + (long)(UNROLL - 1) * T < n2; i += (long)UNROLL * T) {
            d2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(buf + i + (long)u * T) ;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
        }
        for (; i < n2; i += T) { d2 v = buf[i]; acc += v.x + v.y; }
        __syncthreads();
    }
    if (acc == 1.2345e300) out[blockIdx.x] = acc;
}

template <int UNROLL>
__global__ void stream_plain(const d2* __restrict__ buf, long n2, int reps, double* out)
{
    double acc = 0.0;
    const int T = blockDim.x;
    for (int r = 0; r < reps; ++r) {
        long i = threadIdx.x;
        for (; i + (long)(UNROLL - 1) * T < n2; i += (long)UNROLL * T) {
            d2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = buf[i + (long)u * T];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
        }
        for (; i < n2; i += T) { d2 v = buf[i]; acc += v.x + v.y; }
        __syncthreads();
    }
    if (acc == 1.2345e300) out[blockIdx.x] = acc;
}

int main()
{
    const long sizes[] = {331L * 331 * 8, 200L * 200 * 8, 128 * 1024, 4L << 20};
    double* out;
    hipMalloc(&out, 4096);
    for (long S : sizes) {
        d2* buf;
        hipMalloc(&buf, S + 64);
        hipMemset(buf, 0, S + 64);
        const long n2 = S / 16;
        for (int W : {1, 2, 8}) {
            for (int T : {256, 512, 1024}) {
                for (int mode = 0; mode < 2; ++mode) {
                    const int reps = 200;
                    hipEvent_t a, b;
                    hipEventCreate(&a); hipEventCreate(&b);
                    for (int it = 0; it < 2; ++it) {
                        hipEventRecord(a);
                        if (mode == 0) stream_plain<8><<<W, T>>>(buf, n2, reps, out);
                        else stream<8><<<W, T>>>(buf, n2, reps, out);
                        hipEventRecord(b);
                        hipEventSynchronize(b);
                    }
                    float ms; hipEventElapsedTime(&ms, a, b);
                    printf("S=%8ld B  W=%d T=%4d %s : %.1f us per pass, %.1f GB/s per workgroup\n", S, W, T,
                           mode ? "nt   " : "plain", ms * 1e3 / reps, (double)S * reps / (ms * 1e-3) / 1e9);
                }
            }
        }
        hipFree(buf);
    }
    return 0;
}
