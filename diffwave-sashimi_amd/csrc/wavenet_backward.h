// Launch interface of wavenet_backward.hip (internal to libdws.so).
#pragma once
#include <algorithm>

#include "dws_common.h"

namespace dws {
int launch_rowsum(const float* dY, float* db, int B, int O, int L, float scale, int accumulate, hipStream_t s);
int launch_wgrad(const float* dY, const float* X, const float* addc, int addc_bstride, float* dW, int B, int O, int C,
                 int L, int taps, int d, float scale, hipStream_t s);
int launch_conv_t(const float* dY, const float* W, float* dX, int B, int O, int C, int L, int taps, int d, float scale,
                  int accumulate, hipStream_t s);
int launch_gate_bwd(const float* dg, const float* H, float* dH, float* g, int B, int C, int L, hipStream_t s);
int launch_final_dy(const float* dout, const float* Wz, const float* y, float* dy, int B, int S, int Cout, int L,
                    hipStream_t s);
int launch_dx_combine(float* dh, const float* dx_out, size_t n, hipStream_t s);
int launch_scale(const float* in, float* out, float a, size_t n, hipStream_t s);
int launch_rowsum_bc(const float* dh, float* out, int out_bstride, int B, int C, int L, hipStream_t s);
int launch_relu_bwd(float* dx, const float* y, size_t n, hipStream_t s);
int launch_weight_norm_bwd(const float* dW, const float* v, const float* g, float* dv, float* dg, int O, int inner,
                           hipStream_t s);
int launch_lin_bwd_w(const float* dy, const float* x, float* dW, float* db, int B, int K, int O, hipStream_t s);
int launch_lin_bwd_x(const float* dy, const float* W, const float* pre, float* dx, int B, int K, int O, hipStream_t s);
}  // namespace dws
