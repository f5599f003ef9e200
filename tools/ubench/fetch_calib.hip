// Calibration of the FETCH_SIZE / WRITE_SIZE counters on gfx950 against KNOWN byte counts, per access pattern
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of a wide coalesced read stream; the guide's factor was only
// checked for 16-byte-per-lane streams, the register-chained SaShiMi tails read dwords in 128-byte half-wave segments).
//   fetch_calib <MiB>     reads a buffer of that size once per kernel, each byte exactly once:
//     read16   : 16 bytes per lane, a wave covers 1 KiB contiguous                    (LDS-DMA / dwordx4 streams)
//     read4    : 4 bytes per lane, a wave covers 256 contiguous bytes                 (ln_tile, generic kernels)
//     read4seg : 4 bytes per lane, the two halves of a wave read 128-byte segments of two rows 4 rows apart
//                (the accumulator-layout tile I/O of s4_tail_chain*_kernel)
//     write16 / write4seg : the same patterns as stores
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/r06_fetch_calib.sh): counter KiB per dispatch / known KiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read16(const f4* __restrict__ p, size_t n4, float* sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}
__global__ __launch_bounds__(256) void read4(const float* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678f) *sink = acc;
}
// rows of L floats; a wave owns 32 positions x 8 rows per step: lanes 0..31 row r, lanes 32..63 row r + 4, r = 0..3 (two instructions
// apart), i.e. every instruction touches two 128-byte segments L*16 bytes apart
__global__ __launch_bounds__(256) void read4seg(const float* __restrict__ p, int rows, int L, float* sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const int ntl = L / 32;
    float acc = 0.f;
    for (long t = wave; t < (long)(rows / 8) * ntl; t += nw) {
        const int rg = (int)(t / ntl), l0 = (int)(t % ntl) * 32;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc += p[(size_t)(rg * 8 + r + 4 * (lane >> 5)) * L + l0 + (lane & 31)];
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void write16(f4* __restrict__ p, size_t n4) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ __launch_bounds__(256) void write4seg(float* __restrict__ p, int rows, int L) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const int ntl = L / 32;
    for (long t = wave; t < (long)(rows / 8) * ntl; t += nw) {
        const int rg = (int)(t / ntl), l0 = (int)(t % ntl) * 32;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[(size_t)(rg * 8 + r + 4 * (lane >> 5)) * L + l0 + (lane & 31)] = (float)r;
    }
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1024;
    const size_t bytes = mib << 20;
    float *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    const int L = 16000, rows = (int)(bytes / 4 / L / 8) * 8;     // the SaShiMi top stage's row length
    const int grid = 256 * 8;
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const f4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(read4seg, dim3(grid), dim3(256), 0, 0, (const float*)buf, rows, L, sink);
        hipLaunchKernelGGL(write16, dim3(grid), dim3(256), 0, 0, (f4*)buf, bytes / 16);
        hipLaunchKernelGGL(write4seg, dim3(grid), dim3(256), 0, 0, buf, rows, L);
    }
    hipDeviceSynchronize();
    printf("known_kib read16 %zu read4 %zu read4seg %zu write16 %zu write4seg %zu\n", bytes >> 10, bytes >> 10,
           ((size_t)rows * L * 4) >> 10, bytes >> 10, ((size_t)rows * L * 4) >> 10);
    return 0;
}
