// Standalone timing of the product half-transform kernels on a synthetic batch (tools only):
// includes the product source and launches through the C ABI, without python/torch around it.
#include "../auto_oo_amd/csrc/cas.hip"
#include <vector>
int main()
{
    const int N = 43, M = 9, G = 64;
    const size_t ng = (size_t)G * N * N * N * N, nc = (size_t)G * N * N, nt = (size_t)G * N * N * M * M;
    double *g, *C, *T2;
    (void)hipMalloc(&g, ng * 8);
    (void)hipMalloc(&C, nc * 8);
    (void)hipMalloc(&T2, nt * 8);
    std::vector<double> h(1 << 20);
    for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
    for (size_t off = 0; off + h.size() <= ng; off += h.size())
        (void)hipMemcpy(g + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(C, h.data(), nc * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 200;
    float tot = 0, best = 1e30f;
    for (int r = 0; r < reps + 5; ++r) {
        (void)hipEventRecord(e0, 0);
        int rc = half_transform_batched(g, C, N, M, T2, G, nullptr);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 5) { tot += ms; if (ms < best) best = ms; }
    }
    {
        FusedPlan fp;
        if (!fused_plan(N, M, G, &fp)) { printf("no fused plan\n"); return 1; }
        double* Cdup;
        (void)hipMalloc(&Cdup, (size_t)G * fp.nchunk * N * N * 8 + 8);
        float tot2 = 0, best2 = 1e30f;
        for (int r = 0; r < reps + 5; ++r) {
            (void)hipEventRecord(e0, 0);
            int rc = half_transform_fused_batched(g, C, N, M, T2, Cdup, fp, G, nullptr);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 5) { tot2 += ms; if (ms < best2) best2 = ms; }
        }
        printf("fused (nchunk=%d qc=%d nbuf=%d wpg=%d lds=%zu): avg %.1f us best %.1f us -> %.2f TB/s\n", fp.nchunk,
               fp.qc, fp.nbuf, fp.wpg, fp.lds_bytes, tot2 / reps * 1e3, best2 * 1e3,
               (double)ng * 8 / (tot2 / reps * 1e-3) / 1e12);
    }
    for (int mode = SYM_MIRROR; mode <= 5; ++mode) {   // 3 = persistent packed-triangle kernel, 4 = + r<->s on the full layout, 5 = packed integrals
        // p <= q slabs only (the random g is not symmetric: timing only)
        float tot3 = 0, best3 = 1e30f, totq = 0;
        for (int r = 0; r < reps + 5; ++r) {
            (void)hipEventRecord(e0, 0);
            int rc = mode == 5 ? half_tri_batched(g, C, N, M, T2, G, nullptr, 2, true) : mode == 4 ? half_tri_batched(g, C, N, M, T2, G, nullptr, 2, false) : mode == 3 ? half_tri_batched(g, C, N, M, T2, G, nullptr) : half_transform_batched(g, C, N, M, T2, G, nullptr, mode);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 5) { tot3 += ms; if (ms < best3) best3 = ms; }
            if (mode == SYM_PACKED || mode == 3) {
                double* T3 = T2 + (size_t)G * N * (N + 1) / 2 * M * M;
                (void)hipEventRecord(e0, 0);
                rc = sym_q_contract_batched(T2, C, T3, N, M, G, nullptr);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                if (rc) { printf("error: %s\n", oovqe_last_error()); return 1; }
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 5) totq += ms;
            }
        }
        const double tb = (double)G * N * (N + 1) / 2 * N * N * 8;
        printf("sym %s: avg %.1f us best %.1f us -> %.2f TB/s of the triangle", mode == SYM_MIRROR ? "mirror" : mode == SYM_PACKED ? "packed" : mode == 3 ? "packed-persistent" : mode == 4 ? "persistent rs full-layout" : "persistent rs packed-integrals",
               tot3 / reps * 1e3, best3 * 1e3, tb / (tot3 / reps * 1e-3) / 1e12);
        if (mode == SYM_PACKED || mode == 3) printf("; q->x kernel %.1f us", totq / reps * 1e3);
        printf("\n");
    }
    const double bytes = (double)ng * 8;
    printf("half_transform batched N=%d M=%d G=%d: avg %.1f us best %.1f us -> %.2f TB/s (g_ao read)\n", N, M, G,
           tot / reps * 1e3, best * 1e3, bytes / (tot / reps * 1e-3) / 1e12);
    return 0;
}
