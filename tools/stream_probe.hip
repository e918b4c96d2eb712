// Stage 1 of the N = 200 evaluation (half_stream_kernel<2,10,3,true>, mirrored slabs p <= q, r <-> s
// symmetric integrals) alone, without python around it (tools only).  -DOOVQE_STREAM_PROBE=3 builds the
// kernel without its loads after the first chunks; workgroup 0 reports its core cycles and clock.
#ifndef OOVQE_STREAM_PROBE
#define OOVQE_STREAM_PROBE 1
#endif
#include "../auto_oo_amd/csrc/cas.hip"
#include <vector>
int main()
{
    const int N = 200, M = 26;
    const size_t n4 = (size_t)N * N * N * N;
    double *g, *C, *T2;
    (void)hipMalloc(&g, n4 * 8);
    (void)hipMalloc(&C, (size_t)N * N * 8);
    (void)hipMalloc(&T2, (size_t)N * N * M * M * 8);
    std::vector<double> h(1 << 20);
    for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
    for (size_t off = 0; off + h.size() <= n4; off += h.size())
        (void)hipMemcpy(g + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(C, h.data(), (size_t)N * N * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 6; ++r) {
        long long zero[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stream_cyc), zero, sizeof(zero));
        (void)hipEventRecord(e0, 0);
        int rc = half_transform_batched(g, C, N, M, T2, 1, nullptr, SYM_MIRROR, true);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long c[16];
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_stream_cyc), sizeof(c));
        if (r >= 3)
            printf("launch %.1f us; workgroup 0: %lld core cycles in %.1f us = %.0f MHz; MFMA issued by wave 0: %lld (x64 = %lld cycles; two waves per SIMD)\n",
                   ms * 1e3, c[0], c[1] / 100.0, c[0] / (c[1] / 100.0), c[2], c[2] * 64);
    }
    return 0;
}
