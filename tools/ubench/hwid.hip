// Which workgroups share a CU?  Prints (xcc, se, sh/sa, cu) from HW_ID / XCC_ID for the first workgroups of a launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    __builtin_amdgcn_s_sleep(100);
}
int main() {
    const int n = 1024;
    unsigned* d; hipMalloc(&d, n * 8);
    k<<<n, 256, 60000>>>(d);
    unsigned h[2 * n]; hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 40; ++i) {
        unsigned hw = h[2 * i];
        printf("wg %3d hw=%08x wave=%u simd=%u pipe=%u cu=%u sh=%u se=%u  xcc=%08x\n", i, hw, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3,
               (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, h[2 * i + 1]);
    }
    // count distinct (xcc, se, sh, cu)
    int cnt[65536] = {0}, distinct = 0, maxc = 0;
    for (int i = 0; i < n; ++i) {
        unsigned hw = h[2 * i], key = ((h[2 * i + 1] & 15) << 12) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        if (!cnt[key]++) ++distinct;
        if (cnt[key] > maxc) maxc = cnt[key];
    }
    printf("distinct CU keys among %d workgroups: %d (max per key %d)\n", n, distinct, maxc);
    return 0;
}
