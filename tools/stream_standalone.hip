// Standalone timing of the streaming (N > 48) half-transform kernel (tools only): includes the
// product source; -DOOVQE_EXP_NOSTORE builds the "what if the T2 store were free" variant.
#include "../auto_oo_amd/csrc/cas.hip"
#include <vector>
int main()
{
    const int shapes[][3] = {{64, 10, 12}, {96, 12, 3}, {128, 16, 1}, {100, 40, 2}, {200, 26, 1}, {67, 20, 5}};
    std::vector<double> h(1 << 20);
    for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
    for (auto& s : shapes) {
        const int N = s[0], M = s[1], G = s[2];
        const size_t ng = (size_t)G * N * N * N * N, nc = (size_t)G * N * N, nt = (size_t)G * N * N * M * M;
        double *g, *C, *T2;
        (void)hipMalloc(&g, ng * 8);
        (void)hipMalloc(&C, nc * 8);
        (void)hipMalloc(&T2, nt * 8);
        for (size_t off = 0; off + h.size() <= ng; off += h.size())
            (void)hipMemcpy(g + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(C, h.data(), nc * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        const int reps = 50;
        std::vector<double> ref(nt), out(nt);
        for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) setenv("OOVQE_HALF_STREAM_OLD", "1", 1); else unsetenv("OOVQE_HALF_STREAM_OLD");
        (void)hipMemset(T2, 0xff, nt * 8);
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
        (void)hipMemcpy(mode ? out.data() : ref.data(), T2, nt * 8, hipMemcpyDeviceToHost);
        printf("%s N=%d M=%d G=%d: avg %.1f us best %.1f us -> %.2f TB/s (best %.2f)\n", mode ? "new" : "old", N, M, G, tot / reps * 1e3,
               best * 1e3, 8.0 * ng / (tot / reps * 1e-3) / 1e12, 8.0 * ng / (best * 1e-3) / 1e12);
        fflush(stdout);
        }
        double md = 0;
        for (size_t i = 0; i < nt; ++i) { double d = fabs(out[i] - ref[i]); if (!(d <= md)) md = d; }
        printf("   max |new - old| = %.3e\n", md);
        (void)hipFree(g); (void)hipFree(C); (void)hipFree(T2);
    }
    return 0;
}
