// VALU issue-rate probe for gfx950: cycles per wave64 instruction and SIMD for scalar and packed fp32 ops, from s_memtime
// around an unrolled stream of independent instructions, at 1, 2, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048, UNROLL = 16;

template <int KIND>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, float seed) {
    extern __shared__ float pad[];                       // sized so that exactly `blocks per CU` workgroups fit
    float a[UNROLL];
    v2f p[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{seed + i, seed - i + threadIdx.x}; }
    const float m = seed * 0.5f, c = seed * 0.25f;
    const v2f pm = {m, m}, pc = {c, c}, pd = {c, m};
    if (seed == 123.f) pad[threadIdx.x] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 9) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(p[i]) : "v"(pm));
            if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(pd), "v"(p[(i + 5) % UNROLL]));
            if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 3) % UNROLL]), "v"(p[(i + 7) % UNROLL]));
            if (KIND == 12) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 3) % UNROLL]), "v"(a[(i + 7) % UNROLL]));
            if (KIND == 14) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 15) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 18) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
            if (KIND == 19) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 20) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 21) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
            if (KIND == 13) {   // 7 packed + 1 rcp, the Cauchy backward's mix
                if (i % 8 == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 3) % UNROLL]), "v"(p[(i + 7) % UNROLL]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0;
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1;
    }
}

// one workgroup of 64 * 4 * wps threads per CU (wps <= 4), or two of 1024 (wps = 8): the dynamic LDS size leaves room for
// exactly that many, so every SIMD holds wps waves; the span of a workgroup (first wave's start to last wave's end, in
// s_memtime ticks = shader cycles) over its instruction count is the SIMD's issue rate whatever the arbitration order
template <int KIND>
void run(const char* name) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 512 * 1024 * 4);
    hipMalloc(&cyc, 512 * 16 * 2 * 8);
    hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    printf("%-30s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int per_cu = wps == 8 ? 2 : 1, threads = wps == 8 ? 1024 : 256 * wps, blocks = 256 * per_cu;
        const size_t lds = per_cu == 2 ? 70 * 1024 : 120 * 1024;
        hipMemset(cyc, 0, 512 * 16 * 2 * 8);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), lds, 0, out, cyc, 1.0f);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), lds, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(512 * 16 * 2);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        const int nw = threads / 64;
        for (int b = 0; b < blocks; ++b) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < nw; ++w) { lo = std::min(lo, h[(b * 16 + w) * 2]); hi = std::max(hi, h[(b * 16 + w) * 2 + 1]); }
            mean += (double)(hi - lo);
        }
        mean /= blocks;
        // with two workgroups per CU the SIMD ran 8 waves' worth while one workgroup (4 per SIMD) was resident
        printf("  wps=%d: %5.2f", wps, mean / ((double)ITERS * UNROLL * (wps == 8 ? 8 : wps)));
    }
    printf("   cycles per wave-instruction and SIMD\n");
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32");
    run<12>("v_fma_f32 3 distinct srcs");
    run<7>("v_fmac_f32");
    run<1>("v_pk_fma_f32");
    run<11>("v_pk_fma_f32 3 distinct srcs");
    run<10>("v_pk_fma_f32 op_sel_hi");
    run<2>("v_add_f32");
    run<3>("v_pk_add_f32");
    run<9>("v_pk_add_f32 op_sel/neg_hi");
    run<4>("v_mul_f32");
    run<5>("v_pk_mul_f32");
    run<6>("v_rcp_f32");
    run<8>("v_mov_b32");
    run<13>("7 v_pk_fma_f32 + 1 v_rcp_f32");
    run<14>("v_cvt_pk_bf16_f32");
    run<15>("v_cvt_pk_f16_f32");
    run<16>("v_perm_b32");
    run<17>("v_and_b32");
    run<18>("v_cvt_f32_f16");
    run<19>("v_cvt_f16_f32");
    run<20>("v_sub_f32");
    run<21>("v_lshlrev_b32");
    return 0;
}
