// Error reporting and per-family launch timers of libvitmi_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void vitmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vitmi_hip_fail(hipError_t e, const char* what) {
    vitmi_set_error("%s: %s", what, hipGetErrorString(e));
    return VITMI_ERR_HIP;
}

extern "C" int vitmi_version(void) { return VITMI_VERSION; }
extern "C" const char* vitmi_last_error(void) { return g_err; }

// ---- launch timers ----------------------------------------------------------------------------
// Events are created on the device that is current at record time and are only ever read
// through vitmi_prof_read, which synchronises them.  Disabled by default: zero cost.
namespace {
struct Pair {
    hipEvent_t a, b;
    double work;
};
struct Family {
    std::vector<Pair> pairs;
    hipEvent_t pending = nullptr;
    double pending_work = 0;
};
std::mutex g_mu;
bool g_on = false;
int g_only = -1;  // >= 0: only this family is timed
int g_stride = 1;  // every g_stride-th launch of a timed family carries events
long long g_seen[VITMI_FAM_COUNT] = {0};
Family g_fam[VITMI_FAM_COUNT];
std::vector<hipEvent_t> g_pool;  // recycled events: creating one per launch would cost more than recording it

hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void vitmi_prof_record_begin(int family, hipStream_t s, double work) {
    if (!g_on || (g_only >= 0 && family != g_only)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Family& f = g_fam[family];
    if (g_seen[family]++ % g_stride != 0) return;
    hipEvent_t a = take_event();
    if (!a) return;
    (void)hipEventRecord(a, s);
    f.pending = a;
    f.pending_work = work;
}

void vitmi_prof_record_end(int family, hipStream_t s) {
    if (!g_on || (g_only >= 0 && family != g_only)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Family& f = g_fam[family];
    if (!f.pending) return;
    hipEvent_t b = take_event();
    if (!b) return;
    (void)hipEventRecord(b, s);
    f.pairs.push_back({f.pending, b, f.pending_work});
    f.pending = nullptr;
}

extern "C" int vitmi_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return VITMI_OK;
}

extern "C" int vitmi_prof_select(int family) {
    VITMI_CHECK_ARG(family >= -1 && family < VITMI_FAM_COUNT, "vitmi_prof_select: family %d out of range", family);
    std::lock_guard<std::mutex> lk(g_mu);
    g_only = family;
    return VITMI_OK;
}

extern "C" int vitmi_prof_stride(int every) {
    VITMI_CHECK_ARG(every >= 1, "vitmi_prof_stride: stride %d < 1", every);
    std::lock_guard<std::mutex> lk(g_mu);
    g_stride = every;
    return VITMI_OK;
}

extern "C" int vitmi_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& f : g_fam) {
        for (auto& p : f.pairs) {
            g_pool.push_back(p.a);
            g_pool.push_back(p.b);
        }
        f.pairs.clear();
        if (f.pending) g_pool.push_back(f.pending);
        f.pending = nullptr;
    }
    for (auto& n : g_seen) n = 0;
    return VITMI_OK;
}

extern "C" int vitmi_prof_read(int family, double* total_ms, double* total_work, long long* launches) {
    VITMI_CHECK_ARG(family >= 0 && family < VITMI_FAM_COUNT, "vitmi_prof_read: family %d out of range", family);
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0, work = 0;
    long long n = 0;
    for (auto& p : g_fam[family].pairs) {
        hipError_t e = hipEventSynchronize(p.b);
        if (e != hipSuccess) return vitmi_hip_fail(e, "vitmi_prof_read: hipEventSynchronize");
        float t = 0;
        e = hipEventElapsedTime(&t, p.a, p.b);
        if (e != hipSuccess) return vitmi_hip_fail(e, "vitmi_prof_read: hipEventElapsedTime");
        ms += t;
        work += p.work;
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    if (launches) *launches = n;
    return VITMI_OK;
}

extern "C" const char* vitmi_prof_name(int family) {
    static const char* fixed[] = {"gemm(other)", "attn_fwd", "attn_bwd", "layernorm", "tokens", "other"};
    static const char* layout[] = {"A[m][k] B[n][k]", "A[m][k] B[k][n]", "A[k][m] B[n][k]", "A[k][m] B[k][n]"};
    static const char* epi[] = {"plain->bf16", "plain->f32", "gelu->bf16", "dgelu->bf16"};
    static thread_local char buf[128];
    if (family >= 0 && family < VITMI_FAM_GEMM_VARIANT0) return fixed[family];
    if (family >= VITMI_FAM_GEMM_VARIANT0 && family < VITMI_FAM_GEMM_NT_FAST0) {
        const int v = family - VITMI_FAM_GEMM_VARIANT0;
        snprintf(buf, sizeof(buf), "gemm_bf16_kernel[%s, %s]", layout[v >> 2], epi[v & 3]);
        return buf;
    }
    if (family >= VITMI_FAM_GEMM_NT_FAST0 && family < VITMI_FAM_GEMM_TN_FAST) {
        snprintf(buf, sizeof(buf), "gemm_nt_phased_kernel|gemm_nt_fast_kernel[%s]", epi[family - VITMI_FAM_GEMM_NT_FAST0]);
        return buf;
    }
    if (family == VITMI_FAM_GEMM_TN_FAST) return "gemm_tn_batch_kernel|gemm_tn_phased_kernel|gemm_tn_fast_kernel[+reduce]";
    return "?";
}
