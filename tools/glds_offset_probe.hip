// Where does an LDS-DMA piece land for destination offsets beyond 64 KB?  (tools only)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glob_void;
__global__ void probe(const double* g, int dst_off_bytes, int* found, double* val)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int nd = 159 * 1024 / 8;
    for (int i = lane; i < nd; i += 64) lds[i] = -1.0;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((glob_void*)(reinterpret_cast<const char*>(g) + lane * 16),
                                     (lds_void*)(reinterpret_cast<char*>(lds) + dst_off_bytes), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // g holds 1000 + index: find where element 0 and element 127 landed
    if (lane == 0) {
        int f0 = -1, f127 = -1;
        for (int i = 0; i < nd; ++i) { if (lds[i] == 1000.0) f0 = i * 8; if (lds[i] == 1127.0) f127 = i * 8; }
        found[0] = f0; found[1] = f127;
        val[0] = lds[dst_off_bytes / 8];
    }
}
int main()
{
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1000.0 + i;
    double *g, *val; int* found;
    hipMalloc(&g, sizeof(h)); hipMalloc(&val, 8); hipMalloc(&found, 8);
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int off : {0, 32768, 64512, 65536, 66560, 98304, 131072, 150528, 160768}) {
        probe<<<1, 64, 159 * 1024>>>(g, off, found, val);
        int f[2]; double v;
        hipMemcpy(f, found, 8, hipMemcpyDeviceToHost); hipMemcpy(&v, val, 8, hipMemcpyDeviceToHost);
        printf("dst offset %7d: first element landed at byte %7d, last at %7d (expected %d / %d); lds[dst]=%g\n", off, f[0], f[1], off, off + 1016, v);
    }
    return 0;
}
