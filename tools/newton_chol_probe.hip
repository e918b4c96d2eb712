// Phase timing of the Cholesky fast path of the Newton direction (tools only): compiles newton_chol.hip with
// OOVQE_CHOL_TIMING and prints the cycles thread 0 of workgroup 0 spent between the marks.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/newton_chol_probe.hip -o tools/bin/newton_chol_probe
#define OOVQE_CHOL_TIMING 1
#include <stdarg.h>
#include "../auto_oo_amd/csrc/newton_chol.hip"
#include <vector>
#include <random>
void oovqe_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int oovqe_opt(int) { return 0; }
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 331, batch = argc > 2 ? atoi(argv[2]) : 1;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> A((size_t)n * n), H((size_t)batch * n * n), g((size_t)batch * n);
    for (int b = 0; b < batch; ++b) {
        for (auto& v : A) v = nd(rng);
        for (int i = 0; i < n; ++i) {
            g[(size_t)b * n + i] = nd(rng);
            for (int j = 0; j <= i; ++j) {
                double s = i == j ? 0.5 : 0.0;
                for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * A[(size_t)j * n + k] / n;
                H[((size_t)b * n + i) * n + j] = s; H[((size_t)b * n + j) * n + i] = s;
            }
        }
    }
    double *dH, *dg, *dw, *ddp, *dnu, *dinfo;
    hipMalloc(&dH, H.size() * 8); hipMalloc(&dg, g.size() * 8); hipMalloc(&ddp, g.size() * 8);
    hipMalloc(&dnu, batch * 8); hipMalloc(&dinfo, batch * 8);
    hipMalloc(&dw, oovqe_newton_chol_work(n, batch) * 8);
    hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice);
    const char* names[16] = {"prologue", "2: D (wave 0; others: bulk)", "2: barrier", "3: S rows below", "3: barrier",
                             "1: last contribution (wave 0 idle)", "1: barrier", "B: 16x16 transposed solve", "B: barrier",
                             "B: block row + prefetch", "B: barrier", "", "", "", "", ""};
    for (int it = 0; it < 4; ++it) {
        long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_chol_cycles), zero, sizeof(zero));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        int rc = oovqe_newton_chol_launch(dH, dg, n, batch, 1e-6, dw, ddp, dnu, dinfo, nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long cyc[16];
        hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_chol_cycles), sizeof(cyc));
        long long tot = 0; for (int k = 0; k < 16; ++k) tot += cyc[k];
        std::vector<double> info(batch);
        hipMemcpy(info.data(), dinfo, batch * 8, hipMemcpyDeviceToHost);
        printf("rc=%d n=%d batch=%d: %.1f us; cycles total %lld; info[0] %.0f\n", rc, n, batch, ms * 1e3, tot, info[0]);
        if (it == 3) for (int k = 0; k < 11; ++k) printf("  %-30s %10lld  %5.1f %%\n", names[k], cyc[k], 100.0 * cyc[k] / tot);
    }
    return 0;
}
