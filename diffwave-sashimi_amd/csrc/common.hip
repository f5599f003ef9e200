// Error plumbing, ABI identification and the per-kernel event profiler of libdws.so.
#include "dws_common.h"

#include <mutex>

namespace dws {

static thread_local char g_err[1024] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- profiler -------------------------------------------------------------
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::string g_prof_substr;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;

bool profile_active() { return g_prof_on; }

ProfileScope::ProfileScope(const char* name, hipStream_t s) : on(false), stream(s), e0(nullptr), e1(nullptr) {
    if (!g_prof_on) return;
    if (strstr(name, g_prof_substr.c_str()) == nullptr) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
    on = true;
    hipEventRecord(e0, stream);
}

ProfileScope::~ProfileScope() {
    if (!on) return;
    hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_events.emplace_back(e0, e1);
}

}  // namespace dws

extern "C" {

const char* dws_last_error(void) { return dws::g_err; }
int dws_abi_version(void) { return 1; }
const char* dws_arch(void) { return "gfx950"; }

int dws_profile_enable(const char* substr) {
    std::lock_guard<std::mutex> lk(dws::g_prof_mu);
    for (auto& p : dws::g_prof_events) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    dws::g_prof_events.clear();
    dws::g_prof_substr = substr ? substr : "";
    dws::g_prof_on = true;
    return DWS_OK;
}

int dws_profile_query(int64_t* launches, double* total_ms) {
    std::lock_guard<std::mutex> lk(dws::g_prof_mu);
    double tot = 0.0;
    int64_t n = 0;
    for (auto& p : dws::g_prof_events) {
        if (hipEventSynchronize(p.second) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            tot += ms;
            ++n;
        }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    return DWS_OK;
}

int dws_profile_disable(void) {
    std::lock_guard<std::mutex> lk(dws::g_prof_mu);
    dws::g_prof_on = false;
    for (auto& p : dws::g_prof_events) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    dws::g_prof_events.clear();
    return DWS_OK;
}

}  // extern "C"
