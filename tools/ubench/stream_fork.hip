// Micro-test: are cross-stream event waits honoured when the "main" stream is the legacy NULL stream and the side stream is
// hipStreamNonBlocking, with deep queues and no host synchronisation?  (The fork / join pattern of the training step's side
// streams: sashimi_model.hip kernel_backward.)   Build: hipcc --offload-arch=gfx950 -O3 stream_fork.hip -o stream_fork
//   main:  A(buf := iter)  [slow]   record(fork)                 wait(joined of iter-1 .. before the next A)
//   side:                            wait(fork)  B(check buf == iter; early = saw an older value, late = a newer one)  record(joined)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void fill_slow(int* buf, int n, int iter, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (i < n) buf[i] = iter;
}
__global__ void check(const int* buf, int n, int iter, int* early, int* late) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = buf[i];
    if (v < iter) atomicAdd(early, 1);
    if (v > iter) atomicAdd(late, 1);
}
__global__ void busy(float* x, int n, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (i < n) x[i] += 1.f;
}

static void run(const char* name, bool null_main, unsigned evflags, int iters, bool per_iter_events) {
    const int n = 1 << 20;
    int *buf, *cnt;
    float* x;
    hipMalloc(&buf, n * 4); hipMalloc(&cnt, 8); hipMalloc(&x, n * 4);
    hipMemset(buf, 0xff, n * 4); hipMemset(cnt, 0, 8); hipMemset(x, 0, n * 4);
    hipStream_t main_s = nullptr, side;
    if (!null_main) hipStreamCreate(&main_s);
    hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    const int NE = per_iter_events ? iters : 1;
    hipEvent_t* fork = new hipEvent_t[NE];
    hipEvent_t* joined = new hipEvent_t[NE];
    for (int i = 0; i < NE; ++i) { hipEventCreateWithFlags(&fork[i], evflags); hipEventCreateWithFlags(&joined[i], evflags); }
    hipDeviceSynchronize();
    for (int it = 0; it < iters; ++it) {
        const int e = per_iter_events ? it : 0;
        if (it > 0) hipStreamWaitEvent(main_s, joined[per_iter_events ? it - 1 : 0], 0);
        hipLaunchKernelGGL(fill_slow, dim3(n / 256), dim3(256), 0, main_s, buf, n, it, 20000 + (it % 7) * 30000);
        hipEventRecord(fork[e], main_s);
        hipStreamWaitEvent(side, fork[e], 0);
        hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, side, buf, n, it, cnt, cnt + 1);
        hipEventRecord(joined[e], side);
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(busy, dim3(n / 256), dim3(256), 0, main_s, x, n, 5000);   // main runs on
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(busy, dim3(64), dim3(64), 0, side, x + n / 2, 64 * 64, 2000);   // a chain
    }
    hipDeviceSynchronize();
    int h[2];
    hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
    printf("%-60s iters %d: side saw OLDER data (fork ignored) %d, NEWER data (join ignored) %d\n", name, iters, h[0], h[1]);
    hipFree(buf); hipFree(cnt); hipFree(x);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    run("null main, DisableTiming events, one event pair reused", true, hipEventDisableTiming, iters, false);
    run("null main, DisableTiming events, an event pair per iteration", true, hipEventDisableTiming, iters, true);
    run("null main, default events, one event pair reused", true, hipEventDefault, iters, false);
    run("created main, DisableTiming events, one event pair reused", false, hipEventDisableTiming, iters, false);
    run("created main, DisableTiming events, an event pair per iteration", false, hipEventDisableTiming, iters, true);
    return 0;
}
