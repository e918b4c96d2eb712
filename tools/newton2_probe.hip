// Phase timing of the two-stage Newton-direction kernels (tools only): compiles newton.hip with
// OOVQE_NEWTON_TIMING and prints the cycles thread 0 of problem 0's first workgroup spent between the marks.
#define OOVQE_NEWTON_TIMING 1
#include <stdarg.h>
#include "../auto_oo_amd/csrc/newton.hip"
#include <vector>
#include <random>
void oovqe_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int oovqe_opt(int) { return 0; }
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 331, batch = argc > 2 ? atoi(argv[2]) : 1;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> H((size_t)batch * n * n), g((size_t)batch * n);
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < n; ++i) {
            g[(size_t)b * n + i] = nd(rng);
            for (int j = 0; j <= i; ++j) { double v = nd(rng); H[((size_t)b * n + i) * n + j] = v; H[((size_t)b * n + j) * n + i] = v; }
        }
    double *dH, *dg, *dw, *ddp, *dl;
    hipMalloc(&dH, H.size() * 8); hipMalloc(&dg, g.size() * 8); hipMalloc(&ddp, g.size() * 8); hipMalloc(&dl, batch * 8);
    hipMalloc(&dw, oovqe_newton_direction_work_size(n, batch) * 8);
    hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice);
    const char* names[16] = {"copy H", "P: load panel", "P: QR", "P: T + publish", "X: wait for V | T", "X: X0 tiles", "X: reduce + publish",
                             "U: wait for X0", "U: S0, Y, W", "U: trailing update", "Q^T b (workgroup 0)", "2: band + bounds", "2: multisection",
                             "2: Q^T b", "2: LDL^T solve", "2: Q y"};
    for (int it = 0; it < 3; ++it) {
        long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_newton2_cycles), zero, sizeof(zero));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        int rc = oovqe_newton_direction(dH, dg, n, batch, 1e-6, 1e-6, 1.1, 1, dw, ddp, dl, nullptr, nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long cyc[16];
        hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_newton2_cycles), sizeof(cyc));
        long long tot = 0; for (int k = 0; k < 16; ++k) tot += cyc[k];
        printf("rc=%d n=%d batch=%d: %.1f us; cycles total %lld\n", rc, n, batch, ms * 1e3, tot);
        if (it == 2) for (int k = 0; k < 16; ++k) printf("  %-22s %10lld  %5.1f %%\n", names[k], cyc[k], 100.0 * cyc[k] / tot);
    }
    return 0;
}
