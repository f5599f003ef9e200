// Channel-LayerNorm access-pattern probe: the (B, H, L) activations are normalised over H, so one position's H values sit
// L floats apart.  How wide must a wave's row request be before the pass runs at the HBM rate?
//   V = 1: a lane owns one position  (256 B per wave and row: the shape of `ln_tile_kernel` / `ln_bwd_reg_kernel`)
//   V = 2: a lane owns two positions (512 B)      V = 4: four positions (1 KB, one dwordx4 per lane)
// against a flat float4 copy of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ln_rows.hip -o /tmp/ln_rows && /tmp/ln_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };

template <int V> __device__ __forceinline__ void unpack(const typename Vec<V>::T& v, float* f);
template <> __device__ __forceinline__ void unpack<1>(const float& v, float* f) { f[0] = v; }
template <> __device__ __forceinline__ void unpack<2>(const float2& v, float* f) { f[0] = v.x; f[1] = v.y; }
template <> __device__ __forceinline__ void unpack<4>(const float4& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
template <int V> __device__ __forceinline__ typename Vec<V>::T pack(const float* f);
template <> __device__ __forceinline__ float pack<1>(const float* f) { return f[0]; }
template <> __device__ __forceinline__ float2 pack<2>(const float* f) { return make_float2(f[0], f[1]); }
template <> __device__ __forceinline__ float4 pack<4>(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }

// forward LN over H = RP * PARTS channels; block = 64 lanes x PARTS channel groups, 64 * V positions
template <int V, int RP, int PARTS>
__global__ __launch_bounds__(64 * PARTS) void ln_fwd(const float* __restrict__ x, float* __restrict__ out, int L, float m,
                                                      float s) {
    typedef typename Vec<V>::T T;
    constexpr int H = RP * PARTS;
    __shared__ float red[2][PARTS][64 * V];
    const int b = blockIdx.y, col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int l = (blockIdx.x * 64 + col) * V;
    const bool ok = l < L;
    const size_t off = (size_t)b * H * L + (ok ? l : 0);
    float v[RP][V];
#pragma unroll
    for (int r = 0; r < RP; ++r) unpack<V>(*(const T*)(x + off + (size_t)(part + PARTS * r) * L), v[r]);
    float sum[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        sum[j] = 0.f;
#pragma unroll
        for (int r = 0; r < RP; ++r) sum[j] += v[r][j];
        red[0][part][col * V + j] = sum[j];
    }
    __syncthreads();
    float mean[V], var[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        mean[j] = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) mean[j] += red[0][q][col * V + j];
        mean[j] *= 1.f / H;
        var[j] = 0.f;
#pragma unroll
        for (int r = 0; r < RP; ++r) { v[r][j] -= mean[j]; var[j] = fmaf(v[r][j], v[r][j], var[j]); }
        red[1][part][col * V + j] = var[j];
    }
    __syncthreads();
    float sc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) t += red[1][q][col * V + j];
        sc[j] = s / sqrtf(t * (1.f / H));
    }
    if (ok) {
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            float y[V];
#pragma unroll
            for (int j = 0; j < V; ++j) y[j] = sc[j] * v[r][j] + m;
            *(T*)(out + off + (size_t)(part + PARTS * r) * L) = pack<V>(y);
        }
    }
}

// the adjoint's traffic shape: reads x, dy, base; writes out (no GLU branch)
template <int V, int RP, int PARTS>
__global__ __launch_bounds__(64 * PARTS) void ln_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ base, float* __restrict__ out, int L,
                                                      float m, float s) {
    typedef typename Vec<V>::T T;
    constexpr int H = RP * PARTS;
    __shared__ float red[4][PARTS][64 * V];
    const int b = blockIdx.y, col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int l = (blockIdx.x * 64 + col) * V;
    const bool ok = l < L;
    const size_t off = (size_t)b * H * L + (ok ? l : 0);
    float xv[RP][V], dv[RP][V];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        unpack<V>(*(const T*)(x + off + (size_t)(part + PARTS * r) * L), xv[r]);
        unpack<V>(*(const T*)(dy + off + (size_t)(part + PARTS * r) * L), dv[r]);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        float sx = 0.f, sd = 0.f;
#pragma unroll
        for (int r = 0; r < RP; ++r) { sx += xv[r][j]; sd += dv[r][j]; }
        red[0][part][col * V + j] = sx;
        red[1][part][col * V + j] = sd;
    }
    __syncthreads();
    float mean[V], sdy[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        mean[j] = 0.f; sdy[j] = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) { mean[j] += red[0][q][col * V + j]; sdy[j] += red[1][q][col * V + j]; }
        mean[j] *= 1.f / H;
        float var = 0.f, sdx = 0.f;
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            xv[r][j] -= mean[j];
            var = fmaf(xv[r][j], xv[r][j], var);
            sdx = fmaf(dv[r][j], xv[r][j], sdx);
        }
        red[2][part][col * V + j] = var;
        red[3][part][col * V + j] = sdx;
    }
    __syncthreads();
    float rs[V], c2[V], scl[V], mdy[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        float var = 0.f, sdx = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) { var += red[2][q][col * V + j]; sdx += red[3][q][col * V + j]; }
        rs[j] = 1.f / sqrtf(var * (1.f / H));
        mdy[j] = sdy[j] * (1.f / H);
        c2[j] = sdx * rs[j] * (1.f / H) + m * rs[j] * mdy[j];
        scl[j] = s * rs[j];
    }
    if (ok) {
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const size_t ho = off + (size_t)(part + PARTS * r) * L;
            float bv[V], y[V];
            unpack<V>(*(const T*)(base + ho), bv);
#pragma unroll
            for (int j = 0; j < V; ++j) y[j] = scl[j] * (dv[r][j] - mdy[j] - (xv[r][j] * rs[j]) * c2[j]) + bv[j];
            *(T*)(out + ho) = pack<V>(y);
        }
    }
}

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = a[i];
}
__global__ void sum3(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                     float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 p = a[i], q = b[i], r = c[i];
        o[i] = make_float4(p.x + q.x + r.x, p.y + q.y + r.y, p.z + q.z + r.z, p.w + q.w + r.w);
    }
}

template <class F> static float time_us(F f, int reps = 30) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

template <int V, int RP, int PARTS> static void run(const char* tag, int B, int L, float* x, float* dy, float* base, float* out) {
    constexpr int H = RP * PARTS;
    const size_t n = (size_t)B * H * L;
    const dim3 grid((L + 64 * V - 1) / (64 * V), B), blk(64 * PARTS);
    const float tf = time_us([&] { hipLaunchKernelGGL((ln_fwd<V, RP, PARTS>), grid, blk, 0, 0, x, out, L, 0.1f, 1.1f); });
    const float tb = time_us([&] { hipLaunchKernelGGL((ln_bwd<V, RP, PARTS>), grid, blk, 0, 0, x, dy, base, out, L, 0.1f, 1.1f); });
    printf("%-28s H=%3d L=%5d  fwd %7.1f us %5.2f TB/s   bwd %7.1f us %5.2f TB/s\n", tag, H, L, tf, 2.0 * n * 4 / tf * 1e-6, tb,
           4.0 * n * 4 / tb * 1e-6);
}

int main() {
    const int B = 32;        // 262 MB per tensor: well past the 256 MB Infinity Cache, every pass is cold
    const size_t n = (size_t)B * 128 * 16000;
    float *x, *dy, *base, *out;
    (void)hipMalloc(&x, n * 4); (void)hipMalloc(&dy, n * 4); (void)hipMalloc(&base, n * 4); (void)hipMalloc(&out, n * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    (void)hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dy, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(base, h.data(), n * 4, hipMemcpyHostToDevice);
    const float tc = time_us([&] { hipLaunchKernelGGL(copy4, dim3(4096), dim3(256), 0, 0, (const float4*)x, (float4*)out, n / 4); });
    const float ts = time_us([&] { hipLaunchKernelGGL(sum3, dim3(4096), dim3(256), 0, 0, (const float4*)x, (const float4*)dy, (const float4*)base, (float4*)out, n / 4); });
    printf("flat copy  (1r 1w)  %7.1f us %5.2f TB/s\nflat sum3  (3r 1w)  %7.1f us %5.2f TB/s\n", tc, 2.0 * n * 4 / tc * 1e-6, ts,
           4.0 * n * 4 / ts * 1e-6);
    // H = 128, L = 16000 (top stage of the d_model 128 UNet at the training batch)
    run<1, 16, 8>("V1 RP16 PARTS8  (current)", B, 16000, x, dy, base, out);
    run<1, 32, 4>("V1 RP32 PARTS4", B, 16000, x, dy, base, out);
    run<2, 16, 8>("V2 RP16 PARTS8", B, 16000, x, dy, base, out);
    run<2, 8, 16>("V2 RP8  PARTS16", B, 16000, x, dy, base, out);
    run<4, 8, 16>("V4 RP8  PARTS16", B, 16000, x, dy, base, out);
    run<4, 16, 8>("V4 RP16 PARTS8", B, 16000, x, dy, base, out);
    // H = 256, L = 4000
    run<1, 16, 16>("V1 RP16 PARTS16 (current)", B, 4000, x, dy, base, out);
    run<2, 16, 16>("V2 RP16 PARTS16", B, 4000, x, dy, base, out);
    run<4, 16, 16>("V4 RP16 PARTS16", B, 4000, x, dy, base, out);
    // H = 512, L = 1000
    run<1, 32, 16>("V1 RP32 PARTS16 (current)", B, 1000, x, dy, base, out);
    run<2, 32, 16>("V2 RP32 PARTS16", B, 1000, x, dy, base, out);
    run<4, 32, 16>("V4 RP32 PARTS16", B, 1000, x, dy, base, out);
    // H = 64, L = 16000 (d_model 64)
    run<1, 16, 4>("V1 RP16 PARTS4 (current)", B, 16000, x, dy, base, out);
    run<4, 16, 4>("V4 RP16 PARTS4", B, 16000, x, dy, base, out);
    run<4, 8, 8>("V4 RP8 PARTS8", B, 16000, x, dy, base, out);
    return 0;
}
