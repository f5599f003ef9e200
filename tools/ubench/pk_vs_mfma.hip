// Micro-test (gfx950): are packed-FP32 VALU results (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) reliable in a kernel that runs
// on one stream while ANOTHER kernel on another stream executes MFMA on the same CUs?
// Found with the SaShiMi training step's side streams: the Cauchy adjoint (packed math, no MFMA) beside the bf16x6 GEMMs of the
// main stream returned single wrong accumulator halves, a few per launch; compiled without packed-fp32 ops it was bit-exact.
//   hipcc --offload-arch=gfx950 -O3 pk_vs_mfma.hip -o pk_vs_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// VALU worker: a deterministic recurrence on 8 packed (or 16 scalar) accumulators; every lane writes its result
template <bool PACKED>
__global__ __launch_bounds__(64) void valu_kernel(float* out, int iters) {
    const float t = (threadIdx.x + 1) * 1e-3f + blockIdx.x * 1e-6f;
    if (PACKED) {
        v2f a[8];
        for (int k = 0; k < 8; ++k) a[k] = v2f{t + k, t - k};
        const v2f c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = __builtin_elementwise_fma(a[k], c1, c2);
                a[k] = a[k] * c1 + a[(k + 1) & 7] * c2;
            }
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += a[k].x + a[k].y;
        out[blockIdx.x * 64 + threadIdx.x] = s;
    } else {
        float ax[8], ay[8];
        for (int k = 0; k < 8; ++k) { ax[k] = t + k; ay[k] = t - k; }
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ax[k] = fmaf(ax[k], 0.999f, 1e-3f); ay[k] = fmaf(ay[k], 1.001f, -1e-3f);
                ax[k] = ax[k] * 0.999f + ax[(k + 1) & 7] * 1e-3f; ay[k] = ay[k] * 1.001f + ay[(k + 1) & 7] * -1e-3f;
            }
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += ax[k] + ay[k];
        out[blockIdx.x * 64 + threadIdx.x] = s;
    }
}
// MFMA hog: MODE 0 = v_mfma_f32_32x32x2_f32, 1 = v_mfma_f32_32x32x16_bf16
template <int MODE>
__global__ __launch_bounds__(256) void mfma_kernel(float* sink, int iters) {
    f32x16 acc[4] = {};
    const float r = threadIdx.x * 1e-3f;
    bf16x8 hb;
    for (int e = 0; e < 8; ++e) hb[e] = (__bf16)(r + e);
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 0) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 1.f + k, acc[k], 0, 0, 0);
            else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb, hb, acc[k], 0, 0, 0);
        }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345f) sink[0] = 1.f;
}
__global__ void differ(const unsigned* a, const unsigned* b, int n, int* cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(cnt, 1);
}

template <bool PACKED>
static void run(const char* what, int hog, int reps) {
    const int NB = 4096, n = NB * 64;
    float *ref, *out, *sink; int* cnt;
    hipMalloc(&ref, n * 4); hipMalloc(&out, n * 4); hipMalloc(&sink, 4); hipMalloc(&cnt, 4);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipLaunchKernelGGL(valu_kernel<PACKED>, dim3(NB), dim3(64), 0, sa, ref, 2000);
    hipDeviceSynchronize();
    int bad_runs = 0, words = 0;
    for (int it = 0; it < reps; ++it) {
        hipMemsetAsync(cnt, 0, 4, sa);
        if (hog == 1) hipLaunchKernelGGL(mfma_kernel<0>, dim3(1024), dim3(256), 0, sb, sink, 20000);
        if (hog == 2) hipLaunchKernelGGL(mfma_kernel<1>, dim3(1024), dim3(256), 0, sb, sink, 20000);
        hipLaunchKernelGGL(valu_kernel<PACKED>, dim3(NB), dim3(64), 0, sa, out, 2000);
        hipLaunchKernelGGL(differ, dim3(n / 256), dim3(256), 0, sa, (const unsigned*)out, (const unsigned*)ref, n, cnt);
        int c = 0;
        hipMemcpyAsync(&c, cnt, 4, hipMemcpyDeviceToHost, sa);
        hipStreamSynchronize(sa);
        if (c) { ++bad_runs; words += c; }
        hipDeviceSynchronize();
    }
    printf("%-22s beside %-26s: %d of %d launches differ from the launch that ran alone (%d of %d words each on average)\n", what,
           hog == 0 ? "nothing" : hog == 1 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_32x32x16_bf16", bad_runs, reps,
           bad_runs ? words / bad_runs : 0, n);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    for (int hog = 0; hog < 3; ++hog) { run<true>("packed fp32 (v_pk_*)", hog, reps); run<false>("scalar fp32", hog, reps); }
    return 0;
}
