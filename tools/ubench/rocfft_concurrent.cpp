// Micro-test: does a rocFFT real transform return the same bits when OTHER kernels run on the device at the same time?
// (tools/ubench/rocfft_concurrent: the S4 kernel-generation adjoint on a side stream gave perturbed gradients whenever a rocFFT
// execution overlapped other work.)   hipcc --offload-arch=gfx950 -O2 rocfft_concurrent.cpp -o rocfft_concurrent -lrocfft
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void lds_trasher(float* sink, int iters) {       // fills 60 KB of LDS with garbage, over and over
    extern __shared__ float l[];
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < 15360; i += blockDim.x) l[i] = __int_as_float(0x7fc00000 + i + it);
        __syncthreads();
    }
    if (l[threadIdx.x] == 1.2345f) sink[0] = 1.f;
}
__global__ void streamer(float* a, size_t n, int iters) {   // memory traffic
    for (int it = 0; it < iters; ++it)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] * 1.0001f + 1.f;
}
__global__ void differ(const unsigned* a, const unsigned* b, size_t n, int* cnt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != b[i]) atomicAdd(cnt, 1);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2048, BATCH = argc > 2 ? atoi(argv[2]) : 256, dir = argc > 3 ? atoi(argv[3]) : 0;
    rocfft_setup();
    const size_t nreal = (size_t)N * BATCH, ncplx = (size_t)(N / 2 + 1) * BATCH;
    const size_t nin = dir == 0 ? nreal : 2 * ncplx, nout = dir == 0 ? 2 * ncplx : nreal;
    float *in, *in0, *out, *ref, *junk, *sink; int* cnt;
    hipMalloc(&in, nin * 4); hipMalloc(&in0, nin * 4); hipMalloc(&out, nout * 4); hipMalloc(&ref, nout * 4);
    hipMalloc(&junk, (size_t)64 << 20); hipMalloc(&sink, 4); hipMalloc(&cnt, 4);
    std::vector<float> h(nin);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in0, h.data(), nin * 4, hipMemcpyHostToDevice);
    hipMemset(junk, 0, (size_t)64 << 20);
    rocfft_plan plan; size_t len[1] = {(size_t)N};
    rocfft_plan_create(&plan, rocfft_placement_notinplace, dir == 0 ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                       rocfft_precision_single, 1, len, BATCH, nullptr);
    size_t wb = 0; rocfft_plan_get_work_buffer_size(plan, &wb);
    rocfft_execution_info info; rocfft_execution_info_create(&info);
    void* work = nullptr;
    if (wb) { hipMalloc(&work, wb); rocfft_execution_info_set_work_buffer(info, work, wb); }
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    rocfft_execution_info_set_stream(info, sa);
    auto exec = [&](float* dst) {
        hipMemcpyAsync(in, in0, nin * 4, hipMemcpyDeviceToDevice, sa);      // (the inverse may use its input as scratch)
        void* ib[1] = {in}; void* ob[1] = {dst};
        rocfft_execute(plan, ib, ob, info);
    };
    exec(ref);
    hipDeviceSynchronize();
    printf("real %s n=%d batch=%d work=%zu bytes\n", dir == 0 ? "forward" : "inverse", N, BATCH, wb);
    const char* names[] = {"alone", "beside an LDS-filling kernel", "beside a memory-streaming kernel", "beside both"};
    for (int mode = 0; mode < 4; ++mode) {
        int bad_runs = 0, worst = 0;
        for (int it = 0; it < 200; ++it) {
            hipMemsetAsync(cnt, 0, 4, sa);
            if (mode & 1) hipLaunchKernelGGL(lds_trasher, dim3(1024), dim3(256), 61440, sb, sink, 40);
            if (mode & 2) hipLaunchKernelGGL(streamer, dim3(2048), dim3(256), 0, sb, junk, (size_t)16 << 20, 2);
            exec(out);
            hipLaunchKernelGGL(differ, dim3(256), dim3(256), 0, sa, (const unsigned*)out, (const unsigned*)ref, nout, cnt);
            int c = 0;
            hipMemcpyAsync(&c, cnt, 4, hipMemcpyDeviceToHost, sa);
            hipStreamSynchronize(sa);
            if (c) { ++bad_runs; if (c > worst) worst = c; }
            if ((it & 15) == 15) hipDeviceSynchronize();
        }
        hipDeviceSynchronize();
        printf("  %-36s: %d of 200 executions differ from the reference bits (worst: %d of %zu words)\n", names[mode], bad_runs, worst, nout);
    }
    return 0;
}
