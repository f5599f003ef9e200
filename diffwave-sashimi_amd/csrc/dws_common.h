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
