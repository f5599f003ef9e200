// Launch interface of sashimi_train.hip (internal to libdws.so).
#pragma once
#include "dws_common.h"

namespace dws {
// glu_o / glu_do (optional, H in {32, 64, 128, 256, 512}: ln_bwd_fuses_glu): also apply the GLU adjoint to the gradient
// just produced:  glu_do[b, :2H, l] = [out sg(o_b); out o_a sg(o_b)(1 - sg(o_b))]  with o = glu_o [B, 2H, L]
bool ln_bwd_fuses_glu(int H);
int launch_ln_bwd(const float* x, const float* dy, const float* m_p, const float* s_p, const float* base, float* out,
                  int accumulate, float* partial, int B, int H, int L, hipStream_t s, const float* glu_o = nullptr,
                  float* glu_do = nullptr);
int launch_sum_leading(const float* partial, float* out, size_t n, int k, float scale, hipStream_t s);
int launch_sum_pair(const float* partial, float* out0, float* out1, int k, hipStream_t s);
struct SumPairJob { const float* partial; float* out0; float* out1; int k; int pad; };
int launch_sum_pair_multi(const SumPairJob* table_dev, int njobs, hipStream_t s);   // blockIdx.y = job; same sums as launch_sum_pair
int launch_glu_res(const float* o, const float* x, const float* mel, float* x1, int B, int H, int L, hipStream_t s);
int launch_glu_bwd(const float* dx1, const float* o, float* dout, int B, int H, int L, hipStream_t s);
int launch_pool_rearrange(const float* in, float* out, const float* addend, int dir, int accumulate, int B, int H,
                          int p, int Lp, hipStream_t s);
int launch_add_into(const float* a, float* out, int accumulate, size_t n, hipStream_t s);
// Plain-FMA 1x1 GEMM with the epilogues 0, 2, 3, 4, 5 of tapconv_mfma_kernel, for channel counts its 32-row tiles do not
// cover (test-sized models):  out[b, m, l] = epi(sum_k W[m, k] src[b, k, l]),  W row-major [M][K].
struct GemmRowsArgs {
    const float* W; const float* src; float* out;
    const float* bias; const float* res; const float* addend; const float* aux; const float* addin; float* out2;
    int B, M, K, L, epi;
};
int launch_gemm_rows_generic(const GemmRowsArgs& a, hipStream_t s);
int launch_s4_twosided_pow2_bwd(const float* dK, float* dkt, float* dD, int H, int L, int Nf, float sc, float scD,
                                hipStream_t s);
int launch_s4_woodbury_bwd(const float* r, const float* omega, const float* dt, const float* dkf, float* gr,
                           float* part_dt, int H, int Lh, int n_even, hipStream_t s);
int launch_s4_prep_bwd(const float* C, const float* Bp, const float* P, const float* iwr, const float* wim,
                       const float* log_dt, const float* gv, const float* gw6, const float* part_dt, int nparts, float* gC,
                       float* gB, float* gP, float* giwr, float* gwim, float* glogdt, int H, int N, hipStream_t s);
}  // namespace dws
