// Launch interface of conditioner_kernels.hip (internal to libdws.so).
#pragma once
#include "dws_common.h"
#include "model.h"

namespace dws {
int launch_mel_upsample(const float* in, const float* W, const float* bias, float* out, int Bm, int M, int Tin,
                        int Tout, int s, float slope, hipStream_t st);
int launch_conv1x1_trunc(const float* in, const float* W, const float* bias, float* out, int Bm, int K, int O,
                         int Lin, int L, hipStream_t st);
// Adjoint of one upsampler (training).  `out` = the layer's activation [B][M][Tout]; `dout` = gradient w.r.t. it with
// row stride `dstride` and batch stride `dbstride` floats, only columns < dvalid are non-zero.  Produces din [B][M][Tin]
// (skipped when null), dW [3][2s] (folded weight) and dbias [1].
int launch_mel_upsample_bwd(const float* in, const float* out, const float* dout, const float* W, float* din, float* dW,
                            float* dbias, int B, int M, int Tin, int Tout, int s, int dstride, int dbstride, int dvalid,
                            float slope, hipStream_t st);
// Adjoint of one layer's whole conditioner given d melc [B][O][L] (conditioner_train.hip): gradients of the FOLDED
// upsampler weights [3][2s] / biases [1] and of the folded 1x1 weight [O][MB].
struct CondTrainWs {
    DevBuf u0, u1, du0, du1, tmp, AT, wpart;
};
int conditioner_backward(CondTrainWs& ws, const float* mel, int B, int MB, int Tmel, int s0, int s1, const float* W0f,
                         const float* b0, const float* W1f, const float* b1, const float* Wcf, int O, int L,
                         const float* dmelc, float* gW0f, float* gb0, float* gW1f, float* gb1, float* gWcf, hipStream_t s);
// (Tin-1)*s - 2*(s/2) + 2s : width after one ConvTranspose2d upsampler of stride s.
inline int mel_upsampled_len(int Tin, int s) { return (Tin - 1) * s - 2 * (s / 2) + 2 * s; }
}  // namespace dws
