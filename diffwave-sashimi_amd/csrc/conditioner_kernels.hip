// Mel-spectrogram conditioner (`models/wavenet.py:98-111` == `models/sashimi.py:160-175`).
// Constant per utterance, so it is evaluated once in dws_model_set_condition()
// instead of in every block on every reverse step as the reference does.
#include <cstdlib>

#include "conditioner.h"

namespace dws {

// ConvTranspose2d(1,1,(3,2s), stride=(1,s), padding=(1,s/2)) + leaky_relu(slope):
//   out[m, x] = bias + sum_{ky<3} sum_{ix} in[m + 1 - ky, ix] * W[ky, x + s/2 - ix*s]
// with 0 <= x + s/2 - ix*s < 2s  (SURVEY.md appendix B).  Output width
// (Tin-1)*s - 2*(s/2) + 2s.
__global__ void mel_upsample_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                    const float* __restrict__ bias, float* __restrict__ out, int M, int Tin,
                                    int Tout, int s, float slope) {
    const int bm = blockIdx.z, m = blockIdx.y;
    const float* inb = in + (size_t)bm * M * Tin;
    const int pad = s / 2, kw = 2 * s;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < Tout; x += gridDim.x * blockDim.x) {
        float acc = bias[0];
        // ix*s <= x + pad  and  ix*s > x + pad - 2s
        const int hi = (x + pad) / s;
        for (int ix = hi; ix >= 0 && ix > hi - 3; --ix) {
            const int kx = x + pad - ix * s;
            if (kx < 0 || kx >= kw || ix >= Tin) continue;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = m + 1 - ky;
                if (iy < 0 || iy >= M) continue;
                acc = fmaf(inb[(size_t)iy * Tin + ix], W[ky * kw + kx], acc);
            }
        }
        out[((size_t)bm * M + m) * Tout + x] = acc > 0.f ? acc : acc * slope;
    }
}

// The same sum when the stride is a power of two (the usual hop = 16 x 16): output x receives exactly two input
// columns, ix0 = (x + s/2) >> log2 s with kernel column kx0 = (x + s/2) & (s - 1) and ix0 - 1 with kx0 + s (a third
// candidate always falls outside the 2s-wide kernel), so the column search, its bounds tests and the integer division
// of the general kernel go away; four outputs per thread.  Same order of additions (ix descending, ky ascending):
// bit-identical results.
__global__ __launch_bounds__(256) void mel_upsample_pow2_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int M, int Tin, int Tout, int log2s, float slope) {
    const int bm = blockIdx.z, m = blockIdx.y, s = 1 << log2s, pad = s >> 1, kw = 2 * s;
    const float* inb = in + (size_t)bm * M * Tin;
    const float b0 = bias[0];
    // rows m+1, m, m-1 (ky = 0, 1, 2); a row outside [0, M) contributes nothing
    const bool r0 = m + 1 < M, r2 = m >= 1;
    const float* i0 = inb + (size_t)(r0 ? m + 1 : m) * Tin;
    const float* i1 = inb + (size_t)m * Tin;
    const float* i2 = inb + (size_t)(r2 ? m - 1 : m) * Tin;
    float* ob = out + ((size_t)bm * M + m) * Tout;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int x = (blockIdx.x * 4 + u) * 256 + threadIdx.x;
        if (x >= Tout) break;
        const int xp = x + pad, ixa = xp >> log2s, kxa = xp & (s - 1);
        float acc = b0;
        if (ixa < Tin) {
            if (r0) acc = fmaf(i0[ixa], W[kxa], acc);
            acc = fmaf(i1[ixa], W[kw + kxa], acc);
            if (r2) acc = fmaf(i2[ixa], W[2 * kw + kxa], acc);
        }
        const int ixb = ixa - 1, kxb = kxa + s;
        if (ixb >= 0 && ixb < Tin) {
            if (r0) acc = fmaf(i0[ixb], W[kxb], acc);
            acc = fmaf(i1[ixb], W[kw + kxb], acc);
            if (r2) acc = fmaf(i2[ixb], W[2 * kw + kxb], acc);
        }
        ob[x] = acc > 0.f ? acc : acc * slope;
    }
}

int launch_mel_upsample(const float* in, const float* W, const float* bias, float* out, int Bm, int M, int Tin,
                        int Tout, int s, float slope, hipStream_t st) {
    static const bool generic = getenv("DWS_MEL_UPSAMPLE_GENERIC") != nullptr;
    if (!generic && s >= 2 && (s & (s - 1)) == 0) {
        int log2s = 0;
        while ((1 << log2s) < s) ++log2s;
        dim3 grid(ceil_div(Tout, 1024), M, Bm);
        hipLaunchKernelGGL(mel_upsample_pow2_kernel, grid, dim3(256), 0, st, in, W, bias, out, M, Tin, Tout, log2s, slope);
        return DWS_OK;
    }
    dim3 grid(min(ceil_div(Tout, 256), 1024), M, Bm);
    hipLaunchKernelGGL(mel_upsample_kernel, grid, dim3(256), 0, st, in, W, bias, out, M, Tin, Tout, s, slope);
    return DWS_OK;
}

// out[b, o, l] = bias[o] + sum_k W[o, k] * in[b, k, l]   for l < L (in rows have stride Lin >= L:
// the truncation `mel_spec[:, :, :L]` of `wavenet.py:106-108`).
// A workgroup owns OT output channels x 1024 positions: the weight tile sits in LDS as [k][OT] (one broadcast
// ds_read_b128 feeds four FMAs of every lane), a thread keeps OT accumulators for four positions 256 apart, so the
// input rows are read O / OT times instead of O times.  Summation order over k as in the definition (k ascending).
template <int OT>
__global__ __launch_bounds__(256) void conv1x1_trunc_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int K, int O, int Lin, int L) {
    constexpr int KC = 64, NP = 4;
    __shared__ __attribute__((aligned(16))) float Ws[KC * OT];
    const int b = blockIdx.z, o0 = blockIdx.y * OT, tid = threadIdx.x;
    const float* inb = in + (size_t)b * K * Lin;
    const int l0 = blockIdx.x * (256 * NP) + tid;
    float acc[NP][OT];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < OT; ++j) acc[p][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kn = min(KC, K - k0);
        __syncthreads();
        for (int i = tid; i < kn * OT; i += 256) {
            const int k = i / OT, j = i % OT;
            Ws[i] = (o0 + j < O) ? W[(size_t)(o0 + j) * K + k0 + k] : 0.f;
        }
        __syncthreads();
        for (int k = 0; k < kn; ++k) {
            float x[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int l = l0 + 256 * p;
                x[p] = l < L ? inb[(size_t)(k0 + k) * Lin + l] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < OT; ++j) {
                const float w = Ws[k * OT + j];
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[p][j] = fmaf(w, x[p], acc[p][j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < OT; ++j) {
        if (o0 + j >= O) break;
        const float bj = bias[o0 + j];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int l = l0 + 256 * p;
            if (l < L) out[((size_t)b * O + o0 + j) * L + l] = acc[p][j] + bj;
        }
    }
}

int launch_conv1x1_trunc(const float* in, const float* W, const float* bias, float* out, int Bm, int K, int O,
                         int Lin, int L, hipStream_t st) {
    constexpr int OT = 16;
    dim3 grid(ceil_div(L, 1024), ceil_div(O, OT), Bm);
    hipLaunchKernelGGL(conv1x1_trunc_kernel<OT>, grid, dim3(256), 0, st, in, W, bias, out, K, O, Lin, L);
    return DWS_OK;
}

}  // namespace dws

// ---------------------------------------------------------------------------
// Training: adjoint of the conditioner (`wavenet.py:98-111`, `sashimi.py:160-175`)
// ---------------------------------------------------------------------------
namespace dws {

// gradient entering an upsampler's pre-activation: dout masked to the columns the forward kept
// (`mel_spec[:, :, :L]`) times leaky_relu'(out) (out > 0 <=> pre-activation > 0, the slope is positive)
__device__ __forceinline__ float dpre_at(const float* __restrict__ dout, const float* __restrict__ out, int m, int x,
                                         int M, int Tout, int dstride, int dvalid, float slope) {
    if (m < 0 || m >= M || x < 0 || x >= Tout || x >= dvalid) return 0.f;
    const float d = dout[(size_t)m * dstride + x];
    return out[(size_t)m * Tout + x] > 0.f ? d : d * slope;
}

// din[b, m, t] = sum_{ky<3} sum_{kx<2s} dpre[b, m - 1 + ky, t*s - s/2 + kx] * W[ky, kx]
__global__ void mel_upsample_bwd_input_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                              const float* __restrict__ W, float* __restrict__ din, int M, int Tin,
                                              int Tout, int s, int dstride, int dbstride, int dvalid, float slope) {
    const int b = blockIdx.z, m = blockIdx.y;
    const float* db = dout + (size_t)b * dbstride;
    const float* ob = out + (size_t)b * M * Tout;
    const int kw = 2 * s, pad = s / 2;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tin; t += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < kw; ++kx)
                acc = fmaf(dpre_at(db, ob, m - 1 + ky, t * s - pad + kx, M, Tout, dstride, dvalid, slope), W[ky * kw + kx], acc);
        din[((size_t)b * M + m) * Tin + t] = acc;
    }
}

// dW[ky, kx] = sum_{b, m, t} in[b, m, t] * dpre[b, m - 1 + ky, t*s - s/2 + kx];  block 3*2s (the last one: dbias = sum dpre)
__global__ __launch_bounds__(256) void mel_upsample_bwd_weight_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                                      const float* __restrict__ out, float* __restrict__ dW,
                                                                      float* __restrict__ dbias, int B, int M, int Tin,
                                                                      int Tout, int s, int dstride, int dbstride,
                                                                      int dvalid, float slope) {
    __shared__ float red[4];
    const int kw = 2 * s, pad = s / 2;
    const int w = blockIdx.x;
    float acc = 0.f;
    if (w < 3 * kw) {
        const int ky = w / kw, kx = w % kw;
        const int n = B * M * Tin;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int t = i % Tin, m = (i / Tin) % M, b = i / (Tin * M);
            acc = fmaf(in[i], dpre_at(dout + (size_t)b * dbstride, out + (size_t)b * M * Tout, m - 1 + ky, t * s - pad + kx, M,
                                      Tout, dstride, dvalid, slope), acc);
        }
    } else {
        const int n = B * M * Tout;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int x = i % Tout, m = (i / Tout) % M, b = i / (Tout * M);
            acc += dpre_at(dout + (size_t)b * dbstride, out + (size_t)b * M * Tout, m, x, M, Tout, dstride, dvalid, slope);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = red[0] + red[1] + red[2] + red[3];
        if (w < 3 * kw) dW[w] = v;
        else dbias[0] = v;
    }
}

int launch_mel_upsample_bwd(const float* in, const float* out, const float* dout, const float* W, float* din, float* dW,
                            float* dbias, int B, int M, int Tin, int Tout, int s, int dstride, int dbstride, int dvalid,
                            float slope, hipStream_t st) {
    if (din) {
        dim3 grid(min(ceil_div(Tin, 256), 1024), M, B);
        hipLaunchKernelGGL(mel_upsample_bwd_input_kernel, grid, dim3(256), 0, st, dout, out, W, din, M, Tin, Tout, s, dstride,
                           dbstride, dvalid, slope);
    }
    hipLaunchKernelGGL(mel_upsample_bwd_weight_kernel, dim3(3 * 2 * s + 1), dim3(256), 0, st, in, dout, out, dW, dbias, B, M,
                       Tin, Tout, s, dstride, dbstride, dvalid, slope);
    return DWS_OK;
}

}  // namespace dws
