// Launch interface of wavenet_backward.hip (internal to libdws.so).
#pragma once
#include <algorithm>

#include "dws_common.h"
#include "model.h"

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
int launch_lin_bwd_x(const float* dy, const float* W, const float* pre, float* dx, int B, int K, int O, DevBuf& part,
                     hipStream_t s);

// ---- MFMA adjoints (wavenet_backward_mfma.hip)
struct TapConvArgs {
    const float* src0; int K0;   // [B][K0][L]   first K0 contraction channels
    const float* src1; int K1;   // [B][K1][L]   next K1 (may be 0)
    const float* A;              // A fragments of the [M][nkg_total*8] transposed weight (pack_a_frag order)
    int nkg_total;               // k-groups per fragment row in A (the launch may use a prefix of them)
    int M, T, dil, sign;         // tap t reads position l + sign * (t - T/2) * dil
    int epi;                     // see the kernel: 0 add, 1 gate adjoint, 2 bias, 3 bias+gelu (pre and act), 4 bias+res, 5 gelu',
                                 // 6 bias + GLU + residual (out = o, out2 = x1 = res + o_a sigmoid(o_b) (+ aux))
    float* out;
    const float* addin; float addscale;
    const float* H; float* dH; float* g;
    const float* bias; const float* res; const float* addend; const float* aux; float* out2;
    int B, L;
    int split;                   // 1: precision = bf16x6 (3-term bf16 split, six products) where an instance exists (T = 1, L % 4 == 0)
    // TransposedLayerNorm of the tile's output columns fused into the epilogue (`sashimi.py:17-20`: population std down the
    // channel column, no eps): epi 4 normalises `out` [B,M,L], epi 6 normalises `out2` = x1 [B,M/2,L];
    //   ln_out = (ln_s / std) (v - mean + ln_m) (+ ln_pt[b * ln_pt_bstride + row])
    // Needs the whole channel column in ONE workgroup (tapconv_ln_supported) and the float4 epilogue.
    float* ln_out; const float* ln_m; const float* ln_s; const float* ln_pt; int ln_pt_bstride;
};
bool tapconv_ln_supported(int epi, int M, int L);   // the LayerNorm epilogue exists for this GEMM shape
bool tapconv_mfma_supported(int M, int K0, int K1, int T);
bool tapconv_glu_supported(int M, int K, int L);   // epilogue 6 (GLU + residual fused into the 2H x H GEMM)
int launch_tapconv_mfma(const TapConvArgs& a, hipStream_t s);
int launch_tapconv_pack_transposed(const float* W, float* out, int O, int C, int T, int ldo, int coff, float scale,
                                   hipStream_t s);
// Winograd F(2,3) form of the T = 3, sign = -1 data gradient (wavenet_backward_wino.hip): A = fragments of the [M][4 K]
// transformed weights (launch_tapwino_pack_transposed + pack_a_frag), nkg_total = K / 2; power-of-two dilations
bool tapwino_mfma_supported(int M, int K, int dil);
int launch_tapwino_mfma(const TapConvArgs& a, hipStream_t s);
int launch_tapwino_pack_transposed(const float* W, float* out, int O, int C, hipStream_t s);

struct WgradArgs {
    const float* dY;             // [B][O][L]
    const float* X;              // [B][C][L]
    const float* addc; int addc_bstride;   // optional per-(b, c) constant added to in-range X
    float* partial;              // [nsplit][O][C][T] scratch
    int B, O, C, L, dil, nsplit;
    int xact;                    // 1: the X operand is gelu(X) (recomputed from the saved pre-activation)
    float* bias_part;            // optional [nsplit][O] scratch: also produce dbias[o] = bias_scale * sum_{b,l} dY[b,o,l]
    float* dbias; float bias_scale;
    int xL;                      // row stride of X in floats (0 = L)
    int split;                   // 1: precision = bf16x6 in the 16-byte single-tap kernel (wgrad_dma4)
    // optional (T = 1, C <= WGRAD_WN_MAX_INNER): the weight is weight-normed (W = g v / ||v|| per row) -- the split-K reduce
    // applies the weight-norm adjoint to the row it has just summed and writes dv / dg instead of dW (no folded-gradient
    // buffer, no separate weight_norm_bwd launch)
    const float* wn_v; const float* wn_g; float* wn_dv; float* wn_dg;
};
constexpr int WGRAD_WN_MAX_INNER = 4096;
int wgrad_mfma_nsplit(int B, int O, int C, int L, int T);
int launch_wgrad_mfma(const WgradArgs& a, int T, float scale, float* dW, hipStream_t s);
// The T = 3 weight gradient in Winograd F(2,3) pairing (wavenet_backward_wino.hip): partial is [nsplit][O][C][4]
bool wgrad_wino_supported(const WgradArgs& a);
int wgrad_wino_nsplit(int B, int O, int C, int L, int dil);
int launch_wgrad_wino(const WgradArgs& a, float scale, float* dW, hipStream_t s);
}  // namespace dws
