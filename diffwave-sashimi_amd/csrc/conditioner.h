// Launch interface of conditioner_kernels.hip (internal to libdws.so).
#pragma once
#include "dws_common.h"

namespace dws {
int launch_mel_upsample(const float* in, const float* W, const float* bias, float* out, int Bm, int M, int Tin,
                        int Tout, int s, float slope, hipStream_t st);
int launch_conv1x1_trunc(const float* in, const float* W, const float* bias, float* out, int Bm, int K, int O,
                         int Lin, int L, hipStream_t st);
// (Tin-1)*s - 2*(s/2) + 2s : width after one ConvTranspose2d upsampler of stride s.
inline int mel_upsampled_len(int Tin, int s) { return (Tin - 1) * s - 2 * (s / 2) + 2 * s; }
}  // namespace dws
