// Library-level entry points of liboovqe_hip.so (include/oovqe.h).
#include "common.h"
#include <stdarg.h>
#include <vector>
#include <utility>

static thread_local char g_err[512] = "";

void oovqe_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int oovqe_version(void) { return 100; }

extern "C" const char* oovqe_last_error(void) { return g_err; }

extern "C" int oovqe_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        oovqe_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return -1;
    }
    return n;
}

// ---- optional HIP-event timing of the dominant kernel (used by bench.py for the roofline) -----
static bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static hipEvent_t g_prof_pending = nullptr;

void oovqe_profile_mark_start(hipStream_t st)
{
    if (!g_prof_on) return;
    hipEvent_t a;
    if (hipEventCreate(&a) != hipSuccess) return;
    (void)hipEventRecord(a, st);
    g_prof_pending = a;
}

void oovqe_profile_mark_stop(hipStream_t st)
{
    if (!g_prof_on || !g_prof_pending) return;
    hipEvent_t b;
    if (hipEventCreate(&b) != hipSuccess) return;
    (void)hipEventRecord(b, st);
    g_prof_events.emplace_back(g_prof_pending, b);
    g_prof_pending = nullptr;
}

extern "C" int oovqe_profile_begin(void)
{
    for (auto& p : g_prof_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    g_prof_events.clear();
    g_prof_pending = nullptr;
    g_prof_on = true;
    return 0;
}

extern "C" int oovqe_profile_end(double* total_ms, int* count)
{
    g_prof_on = false;
    double tot = 0.0;
    int n = 0;
    for (auto& p : g_prof_events) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess &&
            hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            tot += ms;
            ++n;
        }
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    g_prof_events.clear();
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    return 0;
}
