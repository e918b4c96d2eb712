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

// ---- dynamic-LDS limit of a kernel, raised at most once per (kernel, device) --------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: a process-wide `static` next to the
// launch would skip the call on a second device (or race between host threads) and the launch above 64 KB
// would fail there.
#include <mutex>
#include <map>
int oovqe_ensure_dynamic_lds(const void* kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = done[std::make_pair(kernel, dev)];
    if (bytes <= have) return 0;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        oovqe_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu): %s", bytes, hipGetErrorString(e));
        return OOVQE_ERR_HIP;
    }
    have = bytes;
    return 0;
}

// ---- test / measurement switches (common.h: oovqe_option_t) -------------------------------------
static int g_opts[OOVQE_OPT_COUNT] = {0};
static const char* const g_opt_names[OOVQE_OPT_COUNT] = {
    "half_stream_old", "gm_two_per_cu", "gm_one_per_cu", "fused_chunks", "tri_plain_w", "cas_unfused",
    "sym_no_rs", "sym_mirror", "sym_simple", "sym_two_step", "no_ride", "tri_mode", "k1_no_pair", "k1_force_wide", "gm_plain_grid",
    "newton_one_wg", "sector_unfused", "sector_probe", "hess_vk_pass", "hess_own_stage1", "panel_rows",
    "k1_force_nt", "newton_no_chol", "tiles_variant", "sector_lambda_w", "sector_rdm_r3", "gm_three_per_cu", "panel_no_w",
    "stage1_free_run", "one_stream"};
static_assert(sizeof(g_opt_names) / sizeof(g_opt_names[0]) == OOVQE_OPT_COUNT, "option name table out of step with oovqe_option_t");

int oovqe_opt(int id) { return (id >= 0 && id < OOVQE_OPT_COUNT) ? g_opts[id] : 0; }

// name (with template arguments) of the stage-1 kernel the last evaluation dispatched: bench.py quotes the
// kernel that ran, not a literal
static char g_stage1[160] = "";
void oovqe_note_stage1(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_stage1, sizeof(g_stage1), fmt, ap);
    va_end(ap);
}
extern "C" const char* oovqe_last_stage1_kernel(void) { return g_stage1; }

extern "C" int oovqe_debug_set_option(const char* name, int value)
{
    for (int i = 0; name && i < OOVQE_OPT_COUNT; ++i)
        if (strcmp(name, g_opt_names[i]) == 0) { g_opts[i] = value; return 0; }
    oovqe_set_error("oovqe_debug_set_option: unknown option '%s'", name ? name : "(null)");
    return OOVQE_ERR_ARG;
}

extern "C" int oovqe_debug_get_option(const char* name)
{
    for (int i = 0; name && i < OOVQE_OPT_COUNT; ++i)
        if (strcmp(name, g_opt_names[i]) == 0) return g_opts[i];
    return -1;
}

// ---- a ring of events and two internal streams per device (common.h) --------------------------------
namespace {
struct Internals {
    hipEvent_t ev[64] = {};
    hipStream_t st[1] = {nullptr};
    unsigned next = 0;
    bool tried = false, streams_tried = false;
};
Internals g_int[16];
std::mutex g_int_mu;

Internals* internals()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Internals& I = g_int[dev];
    if (!I.tried) {
        I.tried = true;
        for (int i = 0; i < 64; ++i)
            if (hipEventCreateWithFlags(&I.ev[i], hipEventDisableTiming) != hipSuccess) I.ev[i] = nullptr;
    }
    return &I;
}
}  // namespace

hipEvent_t oovqe_internal_event()
{
    std::lock_guard<std::mutex> lock(g_int_mu);
    Internals* I = internals();
    // (a wait enqueued on an event refers to the record that preceded it: re-recording the event later, 64 uses
    // on, does not disturb it)
    return I ? I->ev[I->next++ & 63] : nullptr;
}

hipStream_t oovqe_internal_stream(int k)
{
    if (oovqe_opt(OOVQE_OPT_ONE_STREAM) != 0 || k != 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_int_mu);
    Internals* I = internals();
    if (!I) return nullptr;
    // made on first use: HIP multiplexes its streams over a few hardware queues (four by default), and a stream that
    // exists takes its share of them whether it is used or not
    if (!I->streams_tried) {
        I->streams_tried = true;
        if (hipStreamCreateWithFlags(&I->st[0], hipStreamNonBlocking) != hipSuccess) I->st[0] = nullptr;
    }
    return I->st[k];
}

// Sweeps enqueued on different streams are ordered one after the other -- but an event record behind every sweep costs
// the launch that follows ~7 us (a barrier packet), which a one-stream caller (every small-batch evaluation: the
// trials of a line search) must not pay.  So the record is made only while sweeps actually alternate between
// streams: the first sweep that arrives on another stream orders itself behind EVERYTHING enqueued on the previous
// stream so far (an event recorded there at that moment), and from then on every sweep leaves its event; after 64
// sweeps in a row on one stream the records stop again.
static hipStream_t g_s1_stream[16];
static hipEvent_t g_s1_event[16];
static bool g_s1_seen[16];
static int g_s1_multi[16];          // > 0: sweeps have been alternating between streams (counts down on one stream)

int oovqe_stage1_enter(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (oovqe_opt(OOVQE_OPT_STAGE1_FREE_RUN) != 0) return 0;
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool seen, was_multi;
    {
        std::lock_guard<std::mutex> lock(g_int_mu);
        seen = g_s1_seen[dev];
        last = g_s1_stream[dev];
        was_multi = g_s1_multi[dev] > 0;
        ev = was_multi ? g_s1_event[dev] : nullptr;
        if (seen && last != st) g_s1_multi[dev] = 64;
        else if (g_s1_multi[dev] > 0) --g_s1_multi[dev];
    }
    if (!seen || last == st) return 0;
    if (!ev) {
        // the previous sweep left no event: one recorded on its stream now stands behind it (and behind what followed)
        ev = oovqe_internal_event();
        if (!ev) return 0;
        if (hipEventRecord(ev, last) != hipSuccess) {       // (a stream the caller has destroyed since: nothing to wait for)
            (void)hipGetLastError();
            return 0;
        }
    }
    OOVQE_CHECK_HIP(hipStreamWaitEvent(st, ev, 0), "stage 1: hipStreamWaitEvent");
    return 0;
}

int oovqe_stage1_leave(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (oovqe_opt(OOVQE_OPT_STAGE1_FREE_RUN) != 0) return 0;
    bool multi;
    {
        std::lock_guard<std::mutex> lock(g_int_mu);
        g_s1_seen[dev] = true;
        g_s1_stream[dev] = st;
        g_s1_event[dev] = nullptr;
        multi = g_s1_multi[dev] > 0;
    }
    if (!multi) return 0;
    hipEvent_t ev = oovqe_internal_event();
    if (!ev) return 0;
    OOVQE_CHECK_HIP(hipEventRecord(ev, st), "stage 1: hipEventRecord");
    std::lock_guard<std::mutex> lock(g_int_mu);
    g_s1_event[dev] = ev;
    return 0;
}

// ---- optional HIP-event timing of the evaluation kernels (bench.py: roofline + breakdown) -------
// Events come from a pool created in oovqe_profile_begin (nothing is allocated inside the timed
// region); launches beyond the pool size are simply not bracketed.  Label 0 is the dominant
// kernel (the N^4 half-transform sweep); labels 1.. are the other launches of an evaluation.
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_pool;          // pairs: [2i] start, [2i+1] stop
static std::vector<int> g_label;                // label of pair i
static size_t g_used = 0;                       // pairs handed out
static bool g_pending = false;
static bool g_detail = false;                   // also bracket labels > 0

void oovqe_profile_mark_start_l(hipStream_t st, int label)
{
    if (!g_prof_on || 2 * g_used + 1 >= g_pool.size() || (label > 0 && !g_detail)) return;
    (void)hipEventRecord(g_pool[2 * g_used], st);
    g_label[g_used] = label;
    g_pending = true;
}

void oovqe_profile_mark_start(hipStream_t st) { oovqe_profile_mark_start_l(st, 0); }

void oovqe_profile_mark_stop(hipStream_t st)
{
    if (!g_prof_on || !g_pending) return;
    (void)hipEventRecord(g_pool[2 * g_used + 1], st);
    ++g_used;
    g_pending = false;
}

extern "C" int oovqe_profile_begin(void)
{
    const size_t want = 2 * 8192;
    while (g_pool.size() < want) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_pool.push_back(e);
    }
    g_label.assign(g_pool.size() / 2, 0);
    g_used = 0;
    g_pending = false;
    g_prof_on = true;
    g_detail = false;
    return 0;
}

extern "C" int oovqe_profile_begin_detail(void)
{
    oovqe_profile_begin();
    g_detail = true;
    return 0;
}

// totals per label (label < n_labels); total_ms / count refer to label 0 for compatibility
extern "C" int oovqe_profile_end_labels(double* ms_by_label, int* count_by_label, int n_labels)
{
    g_prof_on = false;
    for (int l = 0; l < n_labels; ++l) { ms_by_label[l] = 0.0; count_by_label[l] = 0; }
    for (size_t i = 0; i < g_used; ++i) {
        float ms = 0.f;
        const int l = g_label[i];
        if (l < 0 || l >= n_labels) continue;
        if (hipEventSynchronize(g_pool[2 * i + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, g_pool[2 * i], g_pool[2 * i + 1]) == hipSuccess) {
            ms_by_label[l] += ms;
            ++count_by_label[l];
        }
    }
    g_used = 0;
    return 0;
}

extern "C" int oovqe_profile_end(double* total_ms, int* count)
{
    double ms[1];
    int cnt[1];
    oovqe_profile_end_labels(ms, cnt, 1);
    if (total_ms) *total_ms = ms[0];
    if (count) *count = cnt[0];
    return 0;
}
