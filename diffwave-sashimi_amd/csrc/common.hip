// Error plumbing, ABI identification and the per-kernel event profiler of libdws.so.
#include "dws_common.h"
#include "model.h"

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

// One launch for a table of device-to-device copies (model.h CopyBatch, dws_model_get_grads / update_params): block
// (j, part) copies a slice of job j.
__global__ void multi_copy_kernel(const CopyJob* __restrict__ jobs, int parts) {
    const CopyJob j = jobs[blockIdx.x];
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < j.n; i += (int64_t)parts * blockDim.x) j.dst[i] = j.src[i];
}

int launch_multi_copy(const CopyJob* table_dev, int njobs, hipStream_t s) {
    const int parts = 8;
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)njobs, parts), dim3(256), 0, s, table_dev, parts);
    DWS_HIP(hipGetLastError());
    return DWS_OK;
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

int dws_profile_query_each(double* ms, int64_t capacity, int64_t* launches) {
    std::lock_guard<std::mutex> lk(dws::g_prof_mu);
    int64_t n = 0;
    for (auto& p : dws::g_prof_events) {
        if (hipEventSynchronize(p.second) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.first, p.second) != hipSuccess) continue;
        if (ms && n < capacity) ms[n] = t;
        ++n;
    }
    if (launches) *launches = n;
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
