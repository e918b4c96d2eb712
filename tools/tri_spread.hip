// When do the 256 workgroups of the packed stage-1 kernel finish (tools only)?  (the default realisation;
// run with an argument to select another tri_mode, e.g. 1 = the operand-load kernel with its sweep / burst counters)  Builds cas.hip with
// OOVQE_TRI_PROBE: every workgroup stores the 100 MHz wall clock at its end.
#ifndef OOVQE_TRI_PROBE
#define OOVQE_TRI_PROBE 1   // 2: every geometry reads geometry 0's integrals (no HBM stream)
#endif
#include "../auto_oo_amd/csrc/cas.hip"
#include <vector>
#include <algorithm>
int main(int argc, char** argv)
{
    if (argc > 1) oovqe_debug_set_option("tri_mode", atoi(argv[1]));
    const int N = 43, M = 9, G = 256;
    const size_t psz = (size_t)oovqe_eri_packed_size(N);
    const size_t nc = (size_t)G * N * N, nj = (size_t)G * (N * (N + 1) / 2) * 48;
    double *gp, *C, *J;
    (void)hipMalloc(&gp, (size_t)G * psz * 8);
    (void)hipMalloc(&C, nc * 8);
    (void)hipMalloc(&J, nj * 8);
    std::vector<double> h(1 << 20);
    for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
    for (size_t off = 0; off + h.size() <= (size_t)G * psz; off += h.size())
        (void)hipMemcpy(gp + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(C, h.data(), nc * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 12; ++r) {
        long long zero[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tri_cyc), zero, sizeof(zero));
        (void)hipEventRecord(e0, 0);
        int rc = half_tri_batched(gp, C, N, M, J, G, nullptr, 2, true);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> t(4096);
        (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_tri_wg_end), t.size() * 8);
        std::vector<long long> v(t.begin(), t.begin() + G);
        std::sort(v.begin(), v.end());
        if (r >= 8)
            printf("launch %.1f us; workgroup end times relative to the last one (us): first %.1f, 10%% %.1f, median %.1f, 90%% %.1f\n",
                   ms * 1e3, (v[0] - v[G - 1]) / 100.0, (v[G / 10] - v[G - 1]) / 100.0, (v[G / 2] - v[G - 1]) / 100.0,
                   (v[9 * G / 10] - v[G - 1]) / 100.0);
        if (r == 11) {
            long long c[16];
            (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_tri_cyc), sizeof(c));
            printf("workgroup (0,0): %.1f us on the 100 MHz clock = %.0f core cycles per us; core cycles: whole kernel %lld, bursts %lld, sweeps by wave",
                   c[10] / 100.0, c[9] / (c[10] / 100.0), c[9], c[8]);
            for (int w = 0; w < 8; ++w) printf(" %lld", c[w]);
            printf("\n");
        }
    }
    return 0;
}
