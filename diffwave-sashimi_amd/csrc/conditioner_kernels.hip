// Mel-spectrogram conditioner (`models/wavenet.py:98-111` == `models/sashimi.py:160-175`).
// Constant per utterance, so it is evaluated once in dws_model_set_condition()
// instead of in every block on every reverse step as the reference does.
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

int launch_mel_upsample(const float* in, const float* W, const float* bias, float* out, int Bm, int M, int Tin,
                        int Tout, int s, float slope, hipStream_t st) {
    dim3 grid(min(ceil_div(Tout, 256), 1024), M, Bm);
    hipLaunchKernelGGL(mel_upsample_kernel, grid, dim3(256), 0, st, in, W, bias, out, M, Tin, Tout, s, slope);
    return DWS_OK;
}

// out[b, o, l] = bias[o] + sum_k W[o, k] * in[b, k, l]   for l < L (in rows have stride Lin >= L:
// the truncation `mel_spec[:, :, :L]` of `wavenet.py:106-108`).
__global__ void conv1x1_trunc_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                     const float* __restrict__ bias, float* __restrict__ out, int K, int O, int Lin,
                                     int L) {
    const int b = blockIdx.z, o = blockIdx.y;
    const float* inb = in + (size_t)b * K * Lin;
    const float* w = W + (size_t)o * K;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(w[k], inb[(size_t)k * Lin + l], acc);
        out[((size_t)b * O + o) * L + l] = acc + bias[o];
    }
}

int launch_conv1x1_trunc(const float* in, const float* W, const float* bias, float* out, int Bm, int K, int O,
                         int Lin, int L, hipStream_t st) {
    dim3 grid(min(ceil_div(L, 256), 256), O, Bm);
    hipLaunchKernelGGL(conv1x1_trunc_kernel, grid, dim3(256), 0, st, in, W, bias, out, K, O, Lin, L);
    return DWS_OK;
}

}  // namespace dws
