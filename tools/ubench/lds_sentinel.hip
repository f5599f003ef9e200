// Debugging aid: does any kernel of the training step write outside its own LDS allocation?  A "sentinel" workgroup fills a
// small LDS array with a pattern, idles, and verifies it; run on a side stream beside the step, a corrupted pattern means a
// co-resident workgroup of ANOTHER kernel wrote into this one's LDS.  Also keeps a pattern in VGPRs and in a private global slab.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC lds_sentinel.hip -o liblds_sentinel.so
#include <hip/hip_runtime.h>

constexpr int WORDS = 2048;   // 8 KB of LDS per workgroup: fits beside a 131 KB workgroup on a 160 KB CU

__global__ __launch_bounds__(64) void sentinel_kernel(int* counters, unsigned* slab, int launch_id, int spin) {
    __shared__ unsigned pat[WORDS];
    const unsigned seed = (unsigned)launch_id * 2654435761u + blockIdx.x * 40503u;
    for (int i = threadIdx.x; i < WORDS; i += 64) pat[i] = seed ^ (unsigned)(i * 2246822519u);
    unsigned* mine = slab + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    for (int k = 0; k < 4; ++k) mine[k] = seed + k;
    unsigned r0 = seed + threadIdx.x, r1 = seed * 3u + threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    while (clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
    __syncthreads();
    int bad_lds = 0, bad_reg = 0, bad_mem = 0;
    for (int i = threadIdx.x; i < WORDS; i += 64) bad_lds += pat[i] != (seed ^ (unsigned)(i * 2246822519u));
    asm volatile("" : "+v"(r0), "+v"(r1));
    bad_reg += (r0 != seed + threadIdx.x) + (r1 != seed * 3u + threadIdx.x);
    for (int k = 0; k < 4; ++k) bad_mem += mine[k] != seed + k;
    if (bad_lds) atomicAdd(&counters[3 * launch_id + 0], bad_lds);
    if (bad_reg) atomicAdd(&counters[3 * launch_id + 1], bad_reg);
    if (bad_mem) atomicAdd(&counters[3 * launch_id + 2], bad_mem);
}

extern "C" int sentinel_launch(void* stream, int* counters, unsigned* slab, int launch_id, int nwg, int spin) {
    hipLaunchKernelGGL(sentinel_kernel, dim3(nwg), dim3(64), 0, (hipStream_t)stream, counters, slab, launch_id, spin);
    return (int)hipGetLastError();
}
