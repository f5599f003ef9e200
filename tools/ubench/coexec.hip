// Micro-benchmark: do MFMA (wave A) and VALU / transcendental (wave B) instructions of two waves resident on the
// same SIMD execute concurrently on gfx950?  Build: hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int mode, int iters) {
    const int wave = threadIdx.x >> 6;
    float r = threadIdx.x * 1e-3f;
    if (wave < 4) {            // MFMA waves (one per SIMD)
        if (!(mode & 1)) return;
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 1.f, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 2.f, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 3.f, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 4.f, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {                   // VALU waves (second wave of each SIMD)
        if (!(mode & 2)) return;
        float x0 = r, x1 = r + 1, x2 = r + 2, x3 = r + 3, x4 = r + 4, x5 = r + 5, x6 = r + 6, x7 = r + 7;
        if (mode & 4) {        // transcendental mix
            for (int i = 0; i < iters; ++i) {
                x0 = __builtin_amdgcn_exp2f(x0) * 0.5f; x1 = __builtin_amdgcn_rcpf(x1 + 2.f);
                x2 = fmaf(x2, 0.99f, 0.1f); x3 = fmaf(x3, 0.98f, 0.2f);
                x4 = fmaf(x4, 0.97f, 0.3f); x5 = fmaf(x5, 0.96f, 0.4f);
                x6 = fmaf(x6, 0.95f, 0.5f); x7 = fmaf(x7, 0.94f, 0.6f);
            }
        } else {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    x0 = fmaf(x0, 0.99f, 0.1f); x1 = fmaf(x1, 0.98f, 0.2f); x2 = fmaf(x2, 0.97f, 0.3f); x3 = fmaf(x3, 0.96f, 0.4f);
                    x4 = fmaf(x4, 0.95f, 0.5f); x5 = fmaf(x5, 0.94f, 0.6f); x6 = fmaf(x6, 0.93f, 0.7f); x7 = fmaf(x7, 0.92f, 0.8f);
                }
            }
        }
        r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[] = {"", "MFMA only (4/iter)", "VALU only (16 fma/iter)", "MFMA + VALU", "", "", "TRANS only (2 trans + 6 fma/iter)", "MFMA + TRANS"};
    for (int mode : {1, 2, 3, 6, 7}) {
        k<<<256, 512>>>(d, mode, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<256, 512>>>(d, mode, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.3f ms  (%.1f cycles/iter at 2.4 GHz)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
