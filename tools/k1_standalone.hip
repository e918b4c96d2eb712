// Standalone timing of the K1 contraction kernel on the N=200 quarter transforms (tools only).
#include "../auto_oo_amd/csrc/contract.hip"
#include "../auto_oo_amd/csrc/contract_pair.hip"
#include <vector>
#include <string.h>
int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "small")) {
        // p -> n step of a batched evaluation: out[g][n][xyz] = sum_p C[g][p][n] T3[g][p][xyz]
        const int N = 43, M3 = 729, G = 64;
        double *T, *C, *O;
        (void)hipMalloc(&T, (size_t)G * N * M3 * 8);
        (void)hipMalloc(&C, (size_t)G * N * N * 8);
        (void)hipMalloc(&O, (size_t)G * N * M3 * 8);
        (void)hipMemset(T, 0, (size_t)G * N * M3 * 8);
        (void)hipMemset(C, 0, (size_t)G * N * N * 8);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float tot = 0;
        const int reps = 300;
        for (int r = 0; r < reps + 20; ++r) {
            (void)hipEventRecord(e0, 0);
            int rc = oovqe_mode_contract_batched(T, C, O, 1, N, N, M3, N, 0, G, (long)N * M3, (long)N * N,
                                                 (long)N * M3, nullptr);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 20) tot += ms;
        }
        printf("p->n step, 64 geometries: %.2f us per launch\n", tot / reps * 1e3);
        return 0;
    }
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    if (argc > 2 && !strcmp(argv[2], "nopair")) oovqe_debug_set_option("k1_no_pair", 1);
    if (argc > 2 && !strncmp(argv[2], "nt=", 3)) oovqe_debug_set_option("k1_force_nt", atoi(argv[2] + 3));
    const long n = N, n2 = n * n, n3 = n2 * n, n4 = n3 * n;
    double *g, *w, *C;
    (void)hipMalloc(&g, n4 * 8);
    (void)hipMalloc(&w, n4 * 8);
    (void)hipMalloc(&C, n2 * 8);
    std::vector<double> h(1 << 20);
    for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
    for (long off = 0; off + (long)h.size() <= n4; off += h.size())
        (void)hipMemcpy(g + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(C, h.data(), n2 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const char* names[4] = {"pi,pqrs->iqrs", "qj,iqrs->ijrs", "rk,ijrs->ijks", "sl,ijks->ijkl"};
    for (int step = 0; step < 4; ++step) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            (void)hipEventRecord(e0, 0);
            int rc = 0;
            if (step == 0) rc = oovqe_mode_contract_impl(g, C, w, 1, N, N, n3, N, 0, nullptr);
            if (step == 1) rc = oovqe_mode_contract_impl(g, C, w, n, N, N, n2, N, 0, nullptr);
            if (step == 2) rc = oovqe_mode_contract_impl(g, C, w, n2, N, N, n, N, 0, nullptr);
            if (step == 3) rc = oovqe_mode_contract_impl(g, C, w, n3, N, N, 1, N, 1, nullptr);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: %.2f ms = %.1f TFLOP/s\n", names[step], best, 2.0 * n4 * N / best / 1e9);
#if OOVQE_K1_PROBE & 64
        long long marks[128];
        (void)hipMemcpyFromSymbol(marks, HIP_SYMBOL(g_k1_marks), sizeof(marks));
        if (step < 3 && !(argc > 2 && !strcmp(argv[2], "nopair")))
            (void)hipMemcpyFromSymbol(marks, HIP_SYMBOL(g_k1p_marks), sizeof(marks));
        if (step < 3 && !(argc > 2 && !strcmp(argv[2], "nopair"))) {
            long long clk[2];
            (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_k1p_clk), sizeof(clk));
            printf("   workgroup 0: %lld core cycles in %.1f us = %.0f MHz\n", clk[0], clk[1] / 100.0, clk[0] / (clk[1] / 100.0));
        }
        for (int m = 1; m < (argc > 3 ? 64 : 14); ++m)
            printf("   mark %3lld -> %3lld : %7lld\n", marks[2 * m - 2], marks[2 * m], marks[2 * m + 1] - marks[2 * m - 1]);
#endif
    }
    return 0;
}
