// Internal helpers shared by the libdws.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dws.h"

namespace dws {

// ---- error plumbing -------------------------------------------------------
int set_error(int code, const char* fmt, ...);

#define DWS_CHECK(cond, code, ...)                         \
    do {                                                   \
        if (!(cond)) return ::dws::set_error((code), __VA_ARGS__); \
    } while (0)

#define DWS_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return ::dws::set_error(DWS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,        \
                                    hipGetErrorString(_e), __FILE__, __LINE__);         \
    } while (0)

#define DWS_TRY(expr)              \
    do {                           \
        int _s = (expr);           \
        if (_s != DWS_OK) return _s; \
    } while (0)

// ---- per-kernel event profiling (bench.py roofline leg) --------------------
// When enabled for a name substring, launches wrapped in ProfileScope record a
// hipEvent pair on the launch stream.  Never active during graph capture.
struct ProfileScope {
    ProfileScope(const char* name, hipStream_t s);
    ~ProfileScope();
    bool on;
    hipStream_t stream;
    hipEvent_t e0, e1;
};
bool profile_active();

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Step-table mode of the sampler: element offset of row *idx (the device-resident step counter) of a [T][stride] table;
// 0 when the kernel runs on per-clip rows (idx == nullptr).  idx is a kernel argument: the load is scalar.
__device__ __forceinline__ size_t step_row_off(const int* __restrict__ idx, int stride) {
    return idx ? (size_t)__builtin_amdgcn_readfirstlane(*idx) * (size_t)stride : (size_t)0;
}

// ---------------------------------------------------------------------------
// Fast transcendental forms on v_exp_f32 / v_rcp_f32 (1 ulp each).
// GELU(erf) via Abramowitz-Stegun 7.1.26 for erfc (|eps| <= 1.5e-7):
//   E = poly(t) exp(-x^2/2), t = 1/(1 + p |x|/sqrt2);  Phi(x) = x >= 0 ? 1 - E/2 : E/2
// Measured over [-12, 12]: max |gelu_fast - gelu_fp64| = 4.2e-7 (torch's own fp32 GELU: 1.2e-6).  ~15 VALU
// instructions instead of ~45 for erff -- the GELU epilogues were VALU-bound, not the MFMAs.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dws_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// ReLU that propagates NaN like torch.relu (`wavenet.py:149,204`): fmaxf(NaN, 0) would return 0 and hide a range violation
__device__ __forceinline__ float dws_relu(float x) { return x < 0.f ? 0.f : x; }
__device__ __forceinline__ float dws_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + dws_exp(-x)); }
// Phi(x) (standard normal CDF) and ez = exp(-x^2/2)
__device__ __forceinline__ float dws_norm_cdf(float x, float& ez) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    ez = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float half = 0.5f * (p * t) * ez;
    return x >= 0.f ? 1.f - half : half;
}
// GELU itself in 13 instructions: constants folded (1/sqrt2 into the rational's argument, 1/2 into the polynomial,
// log2(e)/2 into the exponent) and the sign select replaced by  gelu(x) = max(x, 0) - |x| E/2  (both branches of Phi).
__device__ __forceinline__ float dws_gelu(float x) {
    const float w = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164188f, w, 1.f));          // 0.3275911 / sqrt2
    float p = fmaf(t, 0.5307027145f, -0.7265760135f);                          // A&S 7.1.26 coefficients / 2
    p = fmaf(t, p, 0.7107068705f);
    p = fmaf(t, p, -0.142248368f);
    p = fmaf(t, p, 0.127414796f);
    const float ez = __builtin_amdgcn_exp2f(-(w * w) * 0.72134752044448170368f);   // exp(-x^2 / 2)
    return fmaf(-w, (p * t) * ez, fmaxf(x, 0.f));
}
__device__ __forceinline__ float dws_gelu_grad(float x) {   // Phi(x) + x phi(x)
    float ez;
    const float cdf = dws_norm_cdf(x, ez);
    return fmaf(x * 0.39894228040143267794f, ez, cdf);
}

// Per-DEVICE launch state: hipFuncSetAttribute, the CU count and occupancy are properties of a device, not of the process -- a
// launcher's "done once" flag / cached count is an array indexed by the current device's ordinal (one process per GPU is
// the product layout; a process that drives several GPUs still gets every device set up).
constexpr int DWS_MAX_DEVICES = 16;
inline int current_device_slot() {
    int d = 0;
    (void)hipGetDevice(&d);
    return ((unsigned)d < (unsigned)DWS_MAX_DEVICES) ? d : 0;
}

// XCD-aware bijective remap of a linear block id (guide T1): hardware places
// block b on XCD b % 8; give every XCD a contiguous chunk of the tile space so
// neighbouring tiles (which share halo rows) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int xcd = bid % NX, idx = bid / NX;
    int q = nwg / NX, r = nwg % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace dws
