// Launch interface of fftconv_kernels.hip (internal to libdws.so).
#pragma once
#include <cmath>

#include "dws_common.h"

namespace dws {

struct FftConvArgs {
    const float* u;     // [B,H,L]  S4 input (LN1(x) + fc_t(e))
    float* g;           // [B,H,L]  GELU(conv + D u)
    const float* D;     // [H]
    const float2* tw;   // exp(-2 pi i k / M), k < M/2
    const float2* twp;  // exp(-2 pi i brev(2q) / 2M), q < M/2
    const float2* kfa;  // [H][M/2] K_f at k = brev(2q)
    const float2* kfb;  // [H][M/2] K_f at M - k
    const float2* kfs;  // [H][3]   K_f at 0, M, M/2
    int B, H, L;
    // training variants (0 / null = the sampling path)
    float* pre;         // also store the pre-activation conv + D u
    int conj_k;         // multiply by conj(K_f): the adjoint (correlation) of the convolution
    int no_act;         // g = conv + D u without the GELU
};

// dK_f partials of the convolution's kernel gradient: part[bs][h][k] = sum_{b in chunk bs} conj(U_b[k]) * dA_b[k],
// k = 0..M (natural order), U / dA the Nf-point real spectra of the zero-padded rows.
struct FftCorrArgs {
    const float* u;     // [B,H,L]
    const float* da;    // [B,H,L]
    float2* part;       // [nbs][H][M+1]
    const float2* tw;
    const float2* twp;
    int B, H, L, bchunk;
};
int launch_fftcorr(int log2m, const FftCorrArgs& a, hipStream_t s);

bool fftconv_supported(int L, int* log2m);
int launch_fftconv(int log2m, const FftConvArgs& a, hipStream_t s);
int launch_rfft_rows(int log2m, const float* in, float* out, const float* tw, const float* twn, int H, hipStream_t s);
int launch_s4_twosided_pow2(const float* k, float* K, int H, int Lt, int Nf, int Lk, hipStream_t s);
int launch_kf_permute(const float* kf, float* kfa, float* kfb, float* kfs, int H, int log2m, hipStream_t s);
void build_fft_tables(int log2m, std::vector<float>& tw, std::vector<float>& twn, std::vector<float>& twp);

}  // namespace dws
