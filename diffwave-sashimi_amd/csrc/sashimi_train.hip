// Adjoint (training) kernels of the SaShiMi block that are not GEMMs or FFTs
// (`models/sashimi.py:17-20,36-58,143-184`, `models/s4.py:704-807,1391-1437`).
// The GEMMs run on tapconv_mfma / wgrad_mfma (wavenet_backward_mfma.hip), the FFT
// convolution adjoints on fftconv_kernels.hip.
#include <cstdint>
#include <cstdlib>

#include "sashimi_train.h"

namespace dws {

__device__ __forceinline__ float sigm_t(float x) { return dws_sigmoid(x); }
__device__ __forceinline__ float2 cmul_t(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc_t(float2 a, float2 b) {  // conj(a) * b
    return make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float2 cdiv_t(float2 a, float2 b) {
    if (fabsf(b.x) >= fabsf(b.y)) {
        const float r = b.y / b.x, den = b.x + b.y * r;
        return make_float2((a.x + a.y * r) / den, (a.y - a.x * r) / den);
    }
    const float r = b.x / b.y, den = b.x * r + b.y;
    return make_float2((a.x * r + a.y) / den, (a.y * r - a.x) / den);
}
__device__ __forceinline__ float wave_sum_t(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// ---------------------------------------------------------------------------
// TransposedLN adjoint.  y_h = (s/sd)(x_h - mu + m) over the channel column of one (b, l):
//   dx_h = (s/sd) [dy_h - mean(dy) - xh_h (mean(dy xh) + (m/sd) mean(dy))],  xh = (x - mu)/sd
//   ds  += sum_h dy_h (xh_h + m/sd),   dm += (s/sd) sum_h dy_h
// out = (accumulate ? out : 0) + (base ? base : 0) + dx.  Block = 64 positions x 4 channel parts.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ m_p, const float* __restrict__ s_p,
                                                     const float* __restrict__ base, float* __restrict__ out,
                                                     int accumulate, float* __restrict__ partial, int H, int L) {
    __shared__ float red[4][4][64];
    __shared__ float cs[2][64];
    const int b = blockIdx.y, col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int l = blockIdx.x * 64 + col;
    const bool ok = l < L;
    const size_t off = (size_t)b * H * L + (ok ? l : 0);
    const float* __restrict__ xb = x + off;
    const float* __restrict__ db = dy + off;
    float sx = 0.f, sd = 0.f;
    for (int h = part; h < H; h += 4) {
        sx += xb[(size_t)h * L];
        sd += db[(size_t)h * L];
    }
    red[0][part][col] = sx;
    red[1][part][col] = sd;
    __syncthreads();
    const float invH = 1.f / (float)H;
    const float mean = (red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col]) * invH;
    const float sdy = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
    float var = 0.f, sdx = 0.f;
    for (int h = part; h < H; h += 4) {
        const float d = xb[(size_t)h * L] - mean;
        var = fmaf(d, d, var);
        sdx = fmaf(db[(size_t)h * L], d, sdx);
    }
    red[2][part][col] = var;
    red[3][part][col] = sdx;
    __syncthreads();
    var = red[2][0][col] + red[2][1][col] + red[2][2][col] + red[2][3][col];
    sdx = red[3][0][col] + red[3][1][col] + red[3][2][col] + red[3][3][col];
    const float sdv = sqrtf(var * invH), rs = 1.f / sdv;
    const float m = m_p[0], s = s_p[0];
    const float mdy = sdy * invH;
    const float c2 = sdx * rs * invH + m * rs * mdy;
    const float sc = s * rs;
    if (ok) {
        float* __restrict__ ob = out + off;
        const float* __restrict__ bb = base ? base + off : nullptr;
        for (int h = part; h < H; h += 4) {
            const float xh = (xb[(size_t)h * L] - mean) * rs;
            float v = sc * (db[(size_t)h * L] - mdy - xh * c2);
            if (bb) v += bb[(size_t)h * L];
            if (accumulate) v += ob[(size_t)h * L];
            ob[(size_t)h * L] = v;
        }
    }
    if (part == 0) {
        cs[0][col] = ok ? sc * sdy : 0.f;                         // dm
        cs[1][col] = ok ? sdx * rs + m * rs * sdy : 0.f;          // ds
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const float v = wave_sum_t(cs[threadIdx.x >> 6][threadIdx.x & 63]);
        // two planes: partial[0][blk] = dm, partial[1][blk] = ds
        if ((threadIdx.x & 63) == 0)
            partial[(size_t)(threadIdx.x >> 6) * gridDim.x * gridDim.y + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = v;
    }
}

// The same adjoint with the block's x and dy columns held in registers: ONE pass over HBM (the kernel above reads both
// tensors three times; its re-reads mostly miss: 64 positions x H channels x 2 tensors per block, 8 blocks per CU).
// Block = 64 positions x PARTS channel groups, a thread owns RP = H / PARTS channels (h = part + PARTS * r).
template <int RP, int PARTS>
__global__ __launch_bounds__(64 * PARTS) void ln_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ m_p, const float* __restrict__ s_p,
                                                              const float* __restrict__ base, float* __restrict__ out,
                                                              int accumulate, float* __restrict__ partial, int L,
                                                              const float* __restrict__ glu_o, float* __restrict__ glu_do) {
    constexpr int H = RP * PARTS;
    __shared__ float red[4][PARTS][64];
    __shared__ float cs[2][64];
    const int b = blockIdx.y, col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int l = blockIdx.x * 64 + col;
    const bool ok = l < L;
    const size_t off = (size_t)b * H * L + (ok ? l : 0);
    const float* __restrict__ xb = x + off;
    const float* __restrict__ db = dy + off;
    constexpr bool PRE = RP <= 16;        // 3 x 32 values per thread do not fit the 128 VGPRs of a 1024-thread block
    float xv[RP], dv[RP], bv[PRE ? RP : 1];
    const float* __restrict__ bb = base ? base + off : nullptr;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        xv[r] = xb[(size_t)(part + PARTS * r) * L];
        dv[r] = db[(size_t)(part + PARTS * r) * L];
    }
    // the residual gradient added at the end is fetched with the operands: all of the block's loads in flight at once
    if (PRE) {
#pragma unroll
        for (int r = 0; r < RP; ++r) bv[PRE ? r : 0] = bb ? bb[(size_t)(part + PARTS * r) * L] : 0.f;
    }
    float sx = 0.f, sd = 0.f;
#pragma unroll
    for (int r = 0; r < RP; ++r) { sx += xv[r]; sd += dv[r]; }
    red[0][part][col] = sx;
    red[1][part][col] = sd;
    __syncthreads();
    const float invH = 1.f / (float)H;
    float mean = 0.f, sdy = 0.f;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) { mean += red[0][q][col]; sdy += red[1][q][col]; }
    mean *= invH;
    float var = 0.f, sdx = 0.f;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        xv[r] -= mean;
        var = fmaf(xv[r], xv[r], var);
        sdx = fmaf(dv[r], xv[r], sdx);
    }
    red[2][part][col] = var;
    red[3][part][col] = sdx;
    __syncthreads();
    var = 0.f; sdx = 0.f;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) { var += red[2][q][col]; sdx += red[3][q][col]; }
    const float sdv = sqrtf(var * invH), rs = 1.f / sdv;
    const float m = m_p[0], s = s_p[0];
    const float mdy = sdy * invH;
    const float c2 = sdx * rs * invH + m * rs * mdy;
    const float sc = s * rs;
    if (ok) {
        float* __restrict__ ob = out + off;
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const size_t ho = (size_t)(part + PARTS * r) * L;
            float v = sc * (dv[r] - mdy - (xv[r] * rs) * c2) + (PRE ? bv[PRE ? r : 0] : (bb ? bb[ho] : 0.f));
            if (accumulate) v += ob[ho];
            ob[ho] = v;
            if (glu_o) {   // the GLU adjoint of `s4.py:1435` on the gradient just produced: d o = [v sg; v o_a sg (1 - sg)]
                const size_t ia = (size_t)b * 2 * H * L + (ok ? l : 0) + ho, ib = ia + (size_t)H * L;
                const float oa = glu_o[ia], sg = sigm_t(glu_o[ib]);
                glu_do[ia] = v * sg;
                glu_do[ib] = v * oa * sg * (1.f - sg);
            }
        }
    }
    if (part == 0) {
        cs[0][col] = ok ? sc * sdy : 0.f;                         // dm
        cs[1][col] = ok ? sdx * rs + m * rs * sdy : 0.f;          // ds
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const float v = wave_sum_t(cs[threadIdx.x >> 6][threadIdx.x & 63]);
        if ((threadIdx.x & 63) == 0)
            partial[(size_t)(threadIdx.x >> 6) * gridDim.x * gridDim.y + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = v;
    }
}

bool ln_bwd_fuses_glu(int H) {
    static const bool old_path = getenv("DWS_LN_BWD_OLD") != nullptr;
    return !old_path && (H == 32 || H == 64 || H == 128 || H == 256 || H == 512);
}

int launch_ln_bwd(const float* x, const float* dy, const float* m_p, const float* s_p, const float* base, float* out,
                  int accumulate, float* partial, int B, int H, int L, hipStream_t s, const float* glu_o, float* glu_do) {
    ProfileScope ps("ln_bwd", s);
    DWS_CHECK(glu_o == nullptr || ln_bwd_fuses_glu(H), DWS_ERR_INVALID, "ln_bwd: the fused GLU adjoint needs H in {32..512}");
    const dim3 grid(ceil_div(L, 64), B);
#define DWS_LN_BWD(RP, PARTS)                                                                                         \
    hipLaunchKernelGGL((ln_bwd_reg_kernel<RP, PARTS>), grid, dim3(64 * PARTS), 0, s, x, dy, m_p, s_p, base, out, accumulate, \
                       partial, L, glu_o, glu_do)
    static const bool old_path = getenv("DWS_LN_BWD_OLD") != nullptr;
    switch (old_path ? 0 : H) {
        case 32: DWS_LN_BWD(8, 4); return DWS_OK;
        case 64: DWS_LN_BWD(16, 4); return DWS_OK;
        case 128: DWS_LN_BWD(16, 8); return DWS_OK;
        case 256: DWS_LN_BWD(16, 16); return DWS_OK;
        case 512: DWS_LN_BWD(32, 16); return DWS_OK;   // (85 us against 225 us for the three-pass kernel below)
    }
#undef DWS_LN_BWD
    hipLaunchKernelGGL(ln_bwd_kernel, grid, dim3(256), 0, s, x, dy, m_p, s_p, base, out, accumulate, partial, H, L);
    return DWS_OK;
}

// out[i] = scale * sum_k partial[k][i]   (fixed order: deterministic)
__global__ void sum_leading_kernel(const float* __restrict__ partial, float* __restrict__ out, size_t n, int k,
                                   float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += partial[(size_t)j * n + i];
    out[i] = s * scale;
}

// few outputs, many partials: one block per output
__global__ __launch_bounds__(256) void sum_leading_wide_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int n, int k, float scale) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < k; j += 256) s += partial[(size_t)j * n + i];
    s = wave_sum_t(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

// two scalars in one launch: out0 = sum partial[0..k), out1 = sum partial[k..2k)  (the dm / ds partials of a LayerNorm
// adjoint, in the same order as sum_leading_wide_kernel sums them)
__global__ __launch_bounds__(256) void sum_pair_kernel(const float* __restrict__ partial, float* __restrict__ out0,
                                                       float* __restrict__ out1, int k) {
    __shared__ float red[4];
    const float* p = partial + (size_t)blockIdx.x * k;
    float s = 0.f;
    for (int j = threadIdx.x; j < k; j += 256) s += p[j];
    s = wave_sum_t(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) (blockIdx.x ? out1 : out0)[0] = red[0] + red[1] + red[2] + red[3];
}

int launch_sum_pair(const float* partial, float* out0, float* out1, int k, hipStream_t s) {
    hipLaunchKernelGGL(sum_pair_kernel, dim3(2), dim3(256), 0, s, partial, out0, out1, k);
    return DWS_OK;
}

// The same sums for a table of jobs in one launch (blockIdx.y = job): the (dm, ds) partials of every LayerNorm adjoint of a
// backward are reduced together at its end instead of by one 8 us launch each (61 per config-5 step).
__global__ __launch_bounds__(256) void sum_pair_multi_kernel(const SumPairJob* __restrict__ jobs) {
    __shared__ float red[4];
    const SumPairJob j = jobs[blockIdx.y];
    const float* p = j.partial + (size_t)blockIdx.x * j.k;
    float s = 0.f;
    for (int i = threadIdx.x; i < j.k; i += 256) s += p[i];
    s = wave_sum_t(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) (blockIdx.x ? j.out1 : j.out0)[0] = red[0] + red[1] + red[2] + red[3];
}

int launch_sum_pair_multi(const SumPairJob* table_dev, int njobs, hipStream_t s) {
    if (njobs <= 0) return DWS_OK;
    hipLaunchKernelGGL(sum_pair_multi_kernel, dim3(2, njobs), dim3(256), 0, s, table_dev);
    DWS_HIP(hipGetLastError());
    return DWS_OK;
}

int launch_sum_leading(const float* partial, float* out, size_t n, int k, float scale, hipStream_t s) {
    if (n <= 64 && k >= 256)
        hipLaunchKernelGGL(sum_leading_wide_kernel, dim3((unsigned)n), dim3(256), 0, s, partial, out, (int)n, k, scale);
    else
        hipLaunchKernelGGL(sum_leading_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, partial, out, n, k, scale);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// Plain-FMA 1x1 GEMM (GemmRowsArgs): thread = position, block = (64 positions, one output row, one batch element).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gemm_rows_generic_kernel(GemmRowsArgs a) {
    const int l = blockIdx.x * 64 + threadIdx.x, m = blockIdx.y, b = blockIdx.z;
    if (l >= a.L) return;
    const float* __restrict__ w = a.W + (size_t)m * a.K;
    const float* __restrict__ x = a.src + (size_t)b * a.K * a.L + l;
    float acc = 0.f;
    for (int k = 0; k < a.K; ++k) acc = fmaf(w[k], x[(size_t)k * a.L], acc);
    const size_t i = ((size_t)b * a.M + m) * a.L + l;
    switch (a.epi) {
        case 0: a.out[i] = acc + (a.addin ? a.addin[i] : 0.f); break;
        case 2: a.out[i] = acc + (a.bias ? a.bias[m] : 0.f); break;
        case 3: {
            const float v = acc + (a.bias ? a.bias[m] : 0.f);
            a.out[i] = v;
            a.out2[i] = dws_gelu(v);
            break;
        }
        case 4: a.out[i] = acc + a.bias[m] + a.res[i] + (a.addend ? a.addend[i] : 0.f); break;
        default: a.out[i] = acc * dws_gelu_grad(a.aux[i]); break;   // 5
    }
}

int launch_gemm_rows_generic(const GemmRowsArgs& a, hipStream_t s) {
    DWS_CHECK(a.epi == 0 || (a.epi >= 2 && a.epi <= 5), DWS_ERR_INVALID, "gemm_rows_generic: epilogue %d", a.epi);
    hipLaunchKernelGGL(gemm_rows_generic_kernel, dim3(ceil_div(a.L, 64), a.M, a.B), dim3(64), 0, s, a);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// GLU + residual (`s4.py:1435`, `sashimi.py:177`): x1 = x + o_a * sigmoid(o_b), o = [o_a; o_b] [B, 2H, L]
// ---------------------------------------------------------------------------
__global__ void glu_res_kernel(const float* __restrict__ o, const float* __restrict__ x, const float* __restrict__ mel,
                               float* __restrict__ x1, int H, int L, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t hl = (size_t)H * L, b = i / hl, r = i % hl;
    const float oa = o[b * 2 * hl + r], ob = o[b * 2 * hl + hl + r];
    x1[i] = x[i] + oa * sigm_t(ob) + (mel ? mel[i] : 0.f);      // + conditioner term (`sashimi.py:160-175`)
}

__global__ void glu_bwd_kernel(const float* __restrict__ dx1, const float* __restrict__ o, float* __restrict__ dout,
                               int H, int L, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t hl = (size_t)H * L, b = i / hl, r = i % hl;
    const float oa = o[b * 2 * hl + r], ob = o[b * 2 * hl + hl + r];
    const float sg = sigm_t(ob), d = dx1[i];
    dout[b * 2 * hl + r] = d * sg;
    dout[b * 2 * hl + hl + r] = d * oa * sg * (1.f - sg);
}

int launch_glu_res(const float* o, const float* x, const float* mel, float* x1, int B, int H, int L, hipStream_t s) {
    const size_t n = (size_t)B * H * L;
    hipLaunchKernelGGL(glu_res_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, o, x, mel, x1, H, L, n);
    return DWS_OK;
}

int launch_glu_bwd(const float* dx1, const float* o, float* dout, int B, int H, int L, hipStream_t s) {
    const size_t n = (size_t)B * H * L;
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, dx1, o, dout, H, L, n);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// Pooling rearrangements (`sashimi.py:37,57`), long = [B, H, Lp*p], wide = [B, H*p, Lp]:
//   dir 0: wide[b, h*p+s, l] = long[b, h, l*p+s]
//   dir 1: long[b, h, l*p+s] = wide[b, h*p+s, l] (+ addend) (+ previous content when accumulate)
// ---------------------------------------------------------------------------
__global__ void pool_rearrange_kernel(const float* __restrict__ in, float* __restrict__ out,
                                      const float* __restrict__ addend, int dir, int accumulate, int H, int p, int Lp,
                                      size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // index in the long layout
    if (i >= n) return;
    const size_t L = (size_t)Lp * p;
    const size_t bh = i / L;
    const int l = (int)(i % L), lp = l / p, sidx = l % p;
    const size_t b = bh / H, h = bh % H;
    const size_t w = ((b * H + h) * p + sidx) * Lp + lp;
    if (dir == 0) {
        out[w] = in[i];
    } else {
        float v = in[w];
        if (addend) v += addend[i];
        if (accumulate) v += out[i];
        out[i] = v;
    }
}

// The same maps with a thread owning the P phases of one pooled position: the long side is one 4P-byte access per lane (a wave
// covers 64 P consecutive floats), the wide side P dwords per lane that are consecutive across the lanes of a row -- both sides
// in full cache lines.  (The per-element kernel above touches the wide side in 64-byte runs: a wave's 64 long positions are 16
// pooled positions of 4 rows.)
template <int P>
__global__ __launch_bounds__(256) void pool_rearrange_vec_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 const float* __restrict__ addend, int dir, int accumulate,
                                                                 int Lp, size_t npool) {
    typedef float vecP __attribute__((ext_vector_type(P)));
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (b h, lp)
    if (t >= npool) return;
    const size_t bh = t / Lp;
    const int lp = (int)(t % Lp);
    const size_t i = t * P;                         // long index of phase 0: bh * (Lp P) + lp P
    const size_t w = bh * P * Lp + lp;              // wide index of phase 0; phase s is s * Lp further on
    if (dir == 0) {
        const vecP v = *reinterpret_cast<const vecP*>(in + i);
#pragma unroll
        for (int q = 0; q < P; ++q) out[w + (size_t)q * Lp] = v[q];
    } else {
        vecP v;
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = in[w + (size_t)q * Lp];
        if (addend) v += *reinterpret_cast<const vecP*>(addend + i);
        if (accumulate) v += *reinterpret_cast<const vecP*>(out + i);
        *reinterpret_cast<vecP*>(out + i) = v;
    }
}

int launch_pool_rearrange(const float* in, float* out, const float* addend, int dir, int accumulate, int B, int H,
                          int p, int Lp, hipStream_t s) {
    const size_t n = (size_t)B * H * p * Lp;
    static const bool per_element = getenv("DWS_POOL_REARRANGE_OLD") != nullptr;      // same-box A/B
    // the long-side tensors are accessed as P-float vectors: rows start at multiples of Lp P floats, so the base pointers decide
    const uintptr_t long_side = (uintptr_t)(dir == 0 ? (const void*)in : (const void*)out) | (uintptr_t)addend;
    if (!per_element && (p == 2 || p == 4) && long_side % (4 * p) == 0) {
        const size_t npool = n / p;
        if (p == 4)
            hipLaunchKernelGGL(pool_rearrange_vec_kernel<4>, dim3(ceil_div(npool, 256)), dim3(256), 0, s, in, out, addend, dir,
                               accumulate, Lp, npool);
        else
            hipLaunchKernelGGL(pool_rearrange_vec_kernel<2>, dim3(ceil_div(npool, 256)), dim3(256), 0, s, in, out, addend, dir,
                               accumulate, Lp, npool);
        return DWS_OK;
    }
    hipLaunchKernelGGL(pool_rearrange_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, out, addend, dir, accumulate,
                       H, p, Lp, n);
    return DWS_OK;
}

// out = (accumulate ? out : 0) + a
__global__ void add_into_kernel(const float* __restrict__ a, float* __restrict__ out, int accumulate, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = accumulate ? out[i] + a[i] : a[i];
}

int launch_add_into(const float* a, float* out, int accumulate, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(add_into_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, a, out, accumulate, n);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// S4 kernel generation adjoint (`s4.py:704-807`; torch convention: the gradient of a complex
// tensor is dL/dRe + i dL/dIm, so a holomorphic y = f(x) gives g_x = conj(f'(x)) g_y)
// ---------------------------------------------------------------------------

// time-domain kernel gradient from the gradient of the re-placed transform input (s4_twosided_pow2):
//   dkt[0][h][j] = dK[h][j] * sc,  dkt[1][h][j] = dK[h][Nf-1-j] * sc;   dD[h] = dK[h][0] * scD
__global__ void s4_twosided_pow2_bwd_kernel(const float* __restrict__ dK, float* __restrict__ dkt, float* __restrict__ dD,
                                            int H, int L, int Nf, float sc, float scD) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (j >= L) return;
    const float* r = dK + (size_t)h * Nf;
    dkt[(size_t)h * L + j] = r[j] * sc;
    dkt[((size_t)H + h) * L + j] = r[Nf - 1 - j] * sc;
    if (j == 0) dD[h] = r[0] * scD;
}

int launch_s4_twosided_pow2_bwd(const float* dK, float* dkt, float* dD, int H, int L, int Nf, float sc, float scD,
                                hipStream_t s) {
    hipLaunchKernelGGL(s4_twosided_pow2_bwd_kernel, dim3(ceil_div(L, 256), H), dim3(256), 0, s, dK, dkt, dD, H, L, Nf, sc,
                       scD);
    return DWS_OK;
}

// Adjoint of s4_woodbury_kernel.  dkf = R2C(dkt) (unnormalised); the C2R it is the adjoint of counts
// interior bins twice and ignores Im of DC / Nyquist.  Writes g_r = dt * g_R [6][H][Lh] and per-block
// partial sums of d(dt) = sum Re(conj(g_R) r).
__global__ __launch_bounds__(256) void s4_woodbury_bwd_kernel(const float2* __restrict__ r, const float2* __restrict__ omega,
                                                              const float* __restrict__ dt, const float2* __restrict__ dkf,
                                                              float2* __restrict__ gr, float* __restrict__ part_dt, int H,
                                                              int Lh, int n_even) {
    __shared__ float red[4];
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    float acc = 0.f;
    if (l < Lh) {
        const float d = dt[h];
        float2 rr[2][3], R[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rr[a][c] = r[((size_t)(a * 3 + c) * H + h) * Lh + l];
                R[a][c] = make_float2(rr[a][c].x * d, rr[a][c].y * d);
            }
        const bool edge = (l == 0) || (n_even && l == Lh - 1);
        const float cf = edge ? 1.f : 2.f;
        const float2 om = omega[l];
        const float2 sfac = cdiv_t(make_float2(2.f, 0.f), make_float2(1.f + om.x, om.y));
        const float2 Dn = make_float2(1.f + R[1][2].x, R[1][2].y);
        const float2 invD = cdiv_t(make_float2(1.f, 0.f), Dn);
        const float2 t02 = cmul_t(R[0][2], invD);              // R02 / D
        float2 gR[2][3];
        gR[0][2] = gR[1][2] = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float2 g = dkf[((size_t)c * H + h) * Lh + l];
            g = make_float2(g.x * cf, edge ? 0.f : g.y * cf);
            const float2 G = cmulc_t(sfac, g);                  // conj(s) * g
            gR[0][c] = G;
            const float2 m1 = cmulc_t(t02, G);                  // conj(R02/D) G
            gR[1][c] = make_float2(-m1.x, -m1.y);
            const float2 t1c = cmul_t(R[1][c], invD);           // R1c / D
            const float2 m2 = cmulc_t(t1c, G);
            gR[0][2] = make_float2(gR[0][2].x - m2.x, gR[0][2].y - m2.y);
            const float2 m3 = cmulc_t(cmul_t(t02, t1c), G);     // conj(R02 R1c / D^2) G
            gR[1][2] = make_float2(gR[1][2].x + m3.x, gR[1][2].y + m3.y);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gr[((size_t)(a * 3 + c) * H + h) * Lh + l] = make_float2(gR[a][c].x * d, gR[a][c].y * d);
                acc += gR[a][c].x * rr[a][c].x + gR[a][c].y * rr[a][c].y;
            }
    }
    acc = wave_sum_t(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part_dt[(size_t)h * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

int launch_s4_woodbury_bwd(const float* r, const float* omega, const float* dt, const float* dkf, float* gr,
                           float* part_dt, int H, int Lh, int n_even, hipStream_t s) {
    hipLaunchKernelGGL(s4_woodbury_bwd_kernel, dim3(ceil_div(Lh, 256), H), dim3(256), 0, s, (const float2*)r,
                       (const float2*)omega, dt, (const float2*)dkf, (float2*)gr, part_dt, H, Lh, n_even);
    return DWS_OK;
}

// Adjoint of s4_prep_kernel: gv [6][H][N], gw6 [6][H][N] (per-row dw of the broadcast Cauchy) ->
// gradients of C [2][H][N], B, P [H][N] (complex), inv_w_real, w_imag [H][N], log_dt [H].
__global__ __launch_bounds__(64) void s4_prep_bwd_kernel(const float2* __restrict__ C, const float2* __restrict__ Bp,
                                                         const float2* __restrict__ P, const float* __restrict__ iwr,
                                                         const float* __restrict__ wim, const float* __restrict__ log_dt,
                                                         const float2* __restrict__ gv, const float2* __restrict__ gw6,
                                                         const float* __restrict__ part_dt, int nparts,
                                                         float2* __restrict__ gC, float2* __restrict__ gB,
                                                         float2* __restrict__ gP, float* __restrict__ giwr,
                                                         float* __restrict__ gwim, float* __restrict__ glogdt, int H,
                                                         int N) {
    const int h = blockIdx.x;
    const float dt = expf(log_dt[h]);
    float gdt = 0.f;
    for (int n = threadIdx.x; n < N; n += 64) {
        const size_t i = (size_t)h * N + n;
        const float2 b = Bp[i], p = P[i];
        const float2 ct[3] = {C[i], C[(size_t)H * N + i], make_float2(p.x, -p.y)};
        const float2 bt[2] = {b, p};
        float2 g[2][3];
        float2 gw = make_float2(0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t k = (size_t)(a * 3 + c) * H * N + i;
                g[a][c] = gv[k];
                gw.x += gw6[k].x;
                gw.y += gw6[k].y;
            }
        float2 gb = make_float2(0.f, 0.f), gp = make_float2(0.f, 0.f), gq = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float2 t0 = cmulc_t(ct[c], g[0][c]), t1 = cmulc_t(ct[c], g[1][c]);
            gb.x += t0.x; gb.y += t0.y;
            gp.x += t1.x; gp.y += t1.y;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float2 t = cmulc_t(bt[a], g[a][2]);   // gradient w.r.t. Ct[2] = conj(P)
            gq.x += t.x; gq.y += t.y;
        }
        gB[i] = gb;
        gP[i] = make_float2(gp.x + gq.x, gp.y - gq.y);  // + conj(g_{conj P})
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float2 t0 = cmulc_t(bt[0], g[0][c]), t1 = cmulc_t(bt[1], g[1][c]);
            gC[(size_t)c * H * N + i] = make_float2(t0.x + t1.x, t0.y + t1.y);
        }
        const float ew = expf(iwr[i]);
        const float wr = -ew, wi = wim[i];
        giwr[i] = gw.x * dt * wr;        // d Re(w)/d inv_w_real = -exp(inv_w_real) = wr
        gwim[i] = gw.y * dt;
        gdt += gw.x * wr + gw.y * wi;    // Re(conj(g_wdt) w)
    }
    for (int j = threadIdx.x; j < nparts; j += 64) gdt += part_dt[(size_t)h * nparts + j];
    gdt = wave_sum_t(gdt);
    if (threadIdx.x == 0) glogdt[h] = gdt * dt;
}

int launch_s4_prep_bwd(const float* C, const float* Bp, const float* P, const float* iwr, const float* wim,
                       const float* log_dt, const float* gv, const float* gw6, const float* part_dt, int nparts, float* gC,
                       float* gB, float* gP, float* giwr, float* gwim, float* glogdt, int H, int N, hipStream_t s) {
    hipLaunchKernelGGL(s4_prep_bwd_kernel, dim3(H), dim3(64), 0, s, (const float2*)C, (const float2*)Bp, (const float2*)P,
                       iwr, wim, log_dt, (const float2*)gv, (const float2*)gw6, part_dt, nparts, (float2*)gC, (float2*)gB,
                       (float2*)gP, giwr, gwim, glogdt, H, N);
    return DWS_OK;
}

}  // namespace dws
