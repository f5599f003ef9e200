// Launch interface of fftconv_kernels.hip (internal to libdws.so).
#pragma once
#include <cmath>

#include "dws_common.h"
#include "fft_core.h"

namespace dws {

struct FftConvArgs {
    const float* u;     // [B,H,L]  S4 input (LN1(x) + fc_t(e))
    float* g;           // [B,H,L]  GELU(conv + D u)
    const float* D;     // [H]
    const c2* tw;   // exp(-2 pi i k / M), k < M/2
    const c2* twp;  // exp(-2 pi i brev(2q) / 2M), q < M/2
    const c2* kfa;  // [H][M/2] K_f at k = brev(2q)
    const c2* kfb;  // [H][M/2] K_f at M - k
    const c2* kfs;  // [H][3]   K_f at 0, M, M/2
    int B, H, L;
    // training variants (0 / null = the sampling path)
    float* pre;         // also store the pre-activation conv + D u
    int conj_k;         // multiply by conj(K_f): the adjoint (correlation) of the convolution
    int no_act;         // g = conv + D u without the GELU
    float* rowsum;      // also rowsum[b * rowsum_bstride + h] = sum_l g[b,h,l] (the adjoint pass: d fc_t(e)[b,h] = sum_l du[b,h,l],
    int rowsum_bstride; //   `sashimi.py:151`); even plans only (fftconv_rowsum_supported): a workgroup owns the whole row
    unsigned long long* trace;   // DWS_FFT_TRACE=1 only: s_memtime stamps [workgroup][wave][row][slot] (fftconv_kernels.hip)
};

// dK_f partials of the convolution's kernel gradient: part[bs][h][k] = sum_{b in chunk bs} conj(U_b[k]) * dA_b[k],
// k = 0..M (natural order), U / dA the Nf-point real spectra of the zero-padded rows.
struct FftCorrArgs {
    const float* u;     // [B,H,L]
    const float* da;    // [B,H,L]
    c2* part;       // [nbs][H][M+1]
    const c2* tw;
    const c2* twp;
    int B, H, L, bchunk;
};
int launch_fftcorr(int log2m, const FftCorrArgs& a, hipStream_t s);
bool fftconv_rowsum_supported(int log2m);   // FftConvArgs::rowsum is honoured for this transform size

// Rows longer than the largest in-LDS transform (L > 16384: vocoding lengths, `generate.py:156`): segments of
// S = 16384 samples; output segment j = first half of IFFT(A_j K_f + A_{j-1} Kc' + A_{j+1} Ka') where A_i is the spectrum
// of input segment i and Kc' / Ka' the spectra of the causal / anti-causal kernel half alone, times (-1)^k (a shift by S).
// Needs at most S kernel taps per direction (`s4.py:1387`: min(L, l_max) taps).
struct FftConvSegArgs {
    const float* u;        // [B,H,L]
    float* g;              // [B,H,L]  GELU(conv + D u)
    const float* D;        // [H]
    const c2* tw;
    const c2* twp;
    const c2* kfa[3];  // pair-ordered spectra [H][M/2]: full, causal', anti-causal'
    const c2* kfb[3];
    const c2* kfs[3];  // [H][3]
    int B, H, L;
};
constexpr int FFTCONV_SEG_LOG2M = 14;
bool fftconv_seg_supported(int L, int taps);
int launch_fftconv_seg(const FftConvSegArgs& a, hipStream_t s);
// which = 0: both halves (== launch_s4_twosided_pow2), 1: causal taps only, 2: anti-causal taps only
int launch_s4_twosided_pow2_part(const float* k, float* K, int H, int Lt, int Nf, int Lk, int which, hipStream_t s);
// sign_alt: multiply bin k by (-1)^k
int launch_kf_permute_signed(const float* kf, float* kfa, float* kfb, float* kfs, int H, int log2m, int sign_alt, hipStream_t s);

bool fftconv_supported(int L, int* log2m);
int launch_fftconv(int log2m, const FftConvArgs& a, hipStream_t s);
int launch_rfft_rows(int log2m, const float* in, float* out, const float* tw, const float* twn, int H, hipStream_t s);
int launch_s4_twosided_pow2(const float* k, float* K, int H, int Lt, int Nf, int Lk, hipStream_t s);
int launch_kf_permute(const float* kf, float* kfa, float* kfb, float* kfs, int H, int log2m, hipStream_t s);
void build_fft_tables(int log2m, std::vector<float>& tw, std::vector<float>& twn, std::vector<float>& twp);

}  // namespace dws
