// Range-check semantics of raw buffer stores / loads with a scalar offset on gfx950 (tools only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(double* out, double* in, double* lres, int nrec_bytes, int so)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, nrec_bytes, 0x00020000);
    const double v = 1000.0 + threadIdx.x;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, threadIdx.x * 8, so, 0);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(in, 0, nrec_bytes, 0x00020000);
    lres[threadIdx.x] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r2, threadIdx.x * 8, so, 0));
}
int main()
{
    double *out, *in, *lres;
    hipMalloc(&out, 4096 * 8); hipMalloc(&in, 4096 * 8); hipMalloc(&lres, 64 * 8);
    double h[4096];
    const int cases[][2] = {{512, 0}, {512, 256}, {512, 512}, {512, 1024}, {256, 128}, {1024, 256}};
    for (auto& c : cases) {
        hipMemset(out, 0, 4096 * 8);
        for (int i = 0; i < 4096; ++i) h[i] = i;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, in, lres, c[0], c[1]);
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("num_records %d soffset %d: stored elements:", c[0], c[1]);
        int first = -1, last = -1, n = 0;
        for (int i = 0; i < 4096; ++i) if (h[i] != 0) { if (first < 0) first = i; last = i; ++n; }
        printf(" count %d first %d (value %.0f) last %d\n", n, first, first >= 0 ? h[first] : 0, last);
        double l[64];
        hipMemcpy(l, lres, sizeof(l), hipMemcpyDeviceToHost);
        int nz = 0, lastnz = -1; for (int i = 0; i < 64; ++i) if (l[i] != 0) { ++nz; lastnz = i; }
        printf("    loads: lane0 %.0f nonzero lanes %d last nonzero lane %d (value %.0f)\n", l[0], nz, lastnz, lastnz >= 0 ? l[lastnz] : 0);
    }
    return 0;
}
