// Launch interface of sashimi_kernels.hip (internal to libdws.so).
#pragma once
#include "dws_common.h"

namespace dws {
int launch_s4_prep(const float* C, const float* Bp, const float* P, const float* iwr, const float* wim,
                   const float* log_dt, float* v, float* wdt, float* dt, int H, int N, hipStream_t s);
int launch_s4_woodbury(const float* r, const float* omega, const float* dt, float* kf, int H, int Lh, int n_even,
                       hipStream_t s);
int launch_s4_twosided(const float* k, float* K, int H, int L, int Lk, int Lt, hipStream_t s);
// step_idx != null (sampler's step-table mode): part_t is row 0 of a [T][pt_tstride] table, the kernel adds row *step_idx
int launch_ln(const float* x, const float* m_p, const float* s_p, const float* part_t, int pt_bstride, float* out,
              int B, int H, int L, size_t ostride, hipStream_t s, const int* step_idx = nullptr, int pt_tstride = 0);
// init_conv + the first block's LN1 + step embedding in one pass (x = relu(conv(audio)), y = LN1(x) + part_t)
bool init_conv_ln_supported(int Cin);
int launch_init_conv_ln(const float* audio, const float* W, const float* bias, const float* m_p, const float* s_p,
                        const float* part_t, int pt_bstride, const int* step_idx, int pt_tstride, float* x, float* y, int B,
                        int Cin, int D, int L, hipStream_t s);
int launch_spec_mul(float* uf, const float* kf, int B, int H, int Lf, hipStream_t s);
int launch_s4_post(const float* yc, const float* u, const float* D, float* g, int B, int H, int L, hipStream_t s);
int launch_pw_conv(const float* in, const float* W, const float* bias, float* out, int B, int K, int O, int L, int act,
                   hipStream_t s);
int launch_pw_glu_res(const float* in, const float* W, const float* bias, const float* res, const float* mel,
                      int mel_bstride, float* out, int B, int H, int L, hipStream_t s);
int launch_pw_res(const float* in, const float* W, const float* bias, const float* res, const float* addend,
                  float* out, int B, int K, int O, int L, hipStream_t s);
int launch_pw_downpool(const float* x, const float* W, const float* bias, float* out, int B, int Hin, int p, int O,
                       int Lout, hipStream_t s);
int launch_pw_uppool(const float* x, const float* W, const float* bias, const float* addend, float* out, int B,
                     int Hin, int p, int Hout, int Lin, hipStream_t s);
int launch_cauchy_sym_fwd_bcast(const float* v, const float* z, const float* w, float* out, int64_t B, int64_t N,
                                int64_t L, int wmod, hipStream_t s);
int launch_cauchy_sym_bwd_bcast(const float* v, const float* z, const float* w, const float* dout, float* dv, float* dw,
                                int64_t B, int64_t N, int64_t L, int wmod, hipStream_t s);
}  // namespace dws
