// Host emulation of a workgroup running the FFT core of the S4 convolution kernels (csrc/fft_core.h), thread by thread
// and pass by pass (a loop over the thread index stands in for the barrier between passes).  Test infrastructure:
// built by tests/test_fft_core_cpu.py with g++, compared against numpy there.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../diffwave-sashimi_amd/csrc/fft_core.h"

using namespace dws;

template <int LOG2M, int P0, bool TAIL = true>
static void fwd_from(float2* X, const float2* tw) {
    using P = FftPlan<LOG2M>;
    constexpr int T = (1 << LOG2M) / 16;
    if constexpr (P0 < P::N16) {
        for (int tid = 0; tid < T; ++tid)
            pass16_lds<LOG2M, P::b0(P0), false>(X, FftTw<LOG2M>::template phi<P::b0(P0)>(tw, tid), tid);
        fwd_from<LOG2M, P0 + 1, TAIL>(X, tw);
    } else if constexpr (P::TAIL4 && TAIL) {
        for (int tid = 0; tid < T; ++tid) pass4_lds<false>(X, FftTw<LOG2M>::tail_twiddle(tw, tid), tid);
    }
}

template <int LOG2M, int P0, int PCUR>
static void inv_passes(float2* X, const float2* tw) {
    constexpr int T = (1 << LOG2M) / 16;
    if constexpr (PCUR > P0) {
        for (int tid = 0; tid < T; ++tid) {
            FftTw<LOG2M> W;
            W.load(tw, tid);
            pass16_lds<LOG2M, FftPlan<LOG2M>::b0(PCUR - 1), true>(X, W.theta[PCUR - 1][0], tid);
        }
        inv_passes<LOG2M, P0, PCUR - 1>(X, tw);
    }
}

template <int LOG2M, int P0, bool TAIL = true>
static void inv_to(float2* X, const float2* tw) {
    using P = FftPlan<LOG2M>;
    constexpr int T = (1 << LOG2M) / 16;
    if constexpr (P::TAIL4 && TAIL)
        for (int tid = 0; tid < T; ++tid) pass4_lds<true>(X, make_float2(1.f, 0.f), tid);
    inv_passes<LOG2M, P0, P::N16>(X, tw);
}

template <int LOG2M>
static void forward_lds(float2* X, const float2* tw) {
    constexpr int M = 1 << LOG2M;
    if constexpr (FftPlan<LOG2M>::ODD)
        for (int t = 0; t < M / 2; ++t) pass2_top<LOG2M, false>(X, tw, t);
    fwd_from<LOG2M, 0>(X, tw);
}

template <int LOG2M>
static void inverse_lds(float2* X, const float2* tw) {
    constexpr int M = 1 << LOG2M;
    inv_to<LOG2M, 0>(X, tw);
    if constexpr (FftPlan<LOG2M>::ODD)
        for (int t = 0; t < M / 2; ++t) pass2_top<LOG2M, true>(X, tw, t);
}

// complex transform, natural order in, BIT-REVERSED order out (forward) / the reverse (inverse, unnormalised)
template <int LOG2M>
static void transform(float* data, const float* tw, int inverse) {
    constexpr int M = 1 << LOG2M;
    std::vector<float2> X(M + M / 16);
    float2* d = reinterpret_cast<float2*>(data);
    for (int i = 0; i < M; ++i) X[pidx(i)] = d[i];
    if (inverse) inverse_lds<LOG2M>(X.data(), reinterpret_cast<const float2*>(tw));
    else forward_lds<LOG2M>(X.data(), reinterpret_cast<const float2*>(tw));
    for (int i = 0; i < M; ++i) d[i] = X[pidx(i)];
}

// One row of fftconv_kernel: out[0..L) = conv part only (scaled by 1/M), as the kernel sequences it -- for even sizes the
// top radix-16 pass runs on registers straight from the (zero padded) input and straight to the output.
// FUSED (even plans that end in a radix-4 tail): forward tail + pair stage + inverse tail as the one pass
// `pass_tail_pointwise` (a thread per block of 16 positions and its mirror block).
template <int LOG2M, bool FUSED = false, bool X4 = false>
static void conv_row(const float* u, int L, const float* tw_, const float* twp_, const float* kfa_, const float* kfb_,
                     const float* kfs_, float csign, float* out) {
    static_assert(!FUSED || (FftPlan<LOG2M>::TAIL4 && !FftPlan<LOG2M>::ODD), "fused tail");
    using P = FftPlan<LOG2M>;
    constexpr int M = 1 << LOG2M, T = M / 16;
    const int Lc = L / 2;
    const float2* u2 = reinterpret_cast<const float2*>(u);
    const float2 *tw = reinterpret_cast<const float2*>(tw_), *twp = reinterpret_cast<const float2*>(twp_);
    const float2 *kfa = reinterpret_cast<const float2*>(kfa_), *kfb = reinterpret_cast<const float2*>(kfb_);
    const float2* kfs = reinterpret_cast<const float2*>(kfs_);
    float2* o2 = reinterpret_cast<float2*>(out);
    std::vector<float2> X(M + M / 16);
    if constexpr (!P::ODD) {
        for (int tid = 0; tid < T; ++tid) {
            FftTw<LOG2M> W;
            W.load(tw, tid);
            float2 x[16];
            for (int r = 0; r < 8; ++r) {
                const int i = tid + T * r;
                x[r] = (i < Lc) ? u2[i] : make_float2(0.f, 0.f);
            }
            for (int r = 8; r < 16; ++r) x[r] = make_float2(0.f, 0.f);
            fft16<false, false, true>(x, make_float2(1.f, 0.f));
            for (int r = 0; r < 16; ++r) X[pidx(tid + T * r)] = x[r];
        }
        fwd_from<LOG2M, 1, !FUSED>(X.data(), tw);
    } else {
        for (int j = 0; j < M; ++j) X[pidx(j)] = (j < Lc) ? u2[j] : make_float2(0.f, 0.f);
        forward_lds<LOG2M>(X.data(), tw);
    }
    if constexpr (FUSED) {
        // every thread reads and writes its own 16 points only: any order of the threads gives the same row
        for (int tid = T - 1; tid >= 0; --tid) pass_tail_pointwise<LOG2M, X4>(X.data(), tw, twp, kfa, kfb, kfs, tid, csign);
    } else if constexpr (X4) {
        // the kernel's pair stage at sizes without a fused tail: the same block / mirror-block order, no butterflies
        for (int tid = T - 1; tid >= 0; --tid)
            pass_tail_pointwise<LOG2M, true, false>(X.data(), tw, twp, kfa, kfb, kfs, tid, csign);
    } else {
        for (int q = 0; q < M / 2; ++q) {
            if (q == 0) {
                if (X4) pointwise_self4(X[pidx(0)], X[pidx(1)], kfs[0], kfs[1], kfs[2], csign);
                else pointwise_self(X[pidx(0)], X[pidx(1)], kfs[0], kfs[1], kfs[2], csign);
                continue;
            }
            const int p = 2 * q;
            const int pm = brev_bits(M - brev_bits(p, LOG2M), LOG2M);
            if (X4) pointwise_pair4(X[pidx(p)], X[pidx(pm)], twp[q], kfa[q], kfb[q], csign);
            else pointwise_pair(X[pidx(p)], X[pidx(pm)], twp[q], kfa[q], kfb[q], csign);
        }
    }
    const float scale = (X4 ? 0.25f : 1.f) / (float)M;
    if constexpr (!P::ODD) {
        inv_to<LOG2M, 1, !FUSED>(X.data(), tw);
        for (int tid = 0; tid < T; ++tid) {
            FftTw<LOG2M> W;
            W.load(tw, tid);
            float2 x[16];
            for (int r = 0; r < 16; ++r) x[r] = X[pidx(tid + T * r)];
            fft16<true, true, false, true>(x, W.theta[0][0]);
            for (int r = 0; r < 8; ++r) {
                const int i = tid + T * r;
                if (i < Lc) o2[i] = make_float2(x[r].x * scale, x[r].y * scale);
            }
        }
    } else {
        inverse_lds<LOG2M>(X.data(), tw);
        for (int j = 0; j < Lc; ++j) o2[j] = make_float2(X[pidx(j)].x * scale, X[pidx(j)].y * scale);
    }
}

// One row of fftconv_seg_kernel: inputs longer than the transform (L > M): output segment j (S = M samples... the real
// transform has Nf = 2M points, a segment S = Nf/2 = M samples) = first S points of
//   IFFT( A_j K_f + A_{j-1} Kc' + A_{j+1} Ka' ),   Kc' / Ka' = (-1)^k x spectrum of the causal / anti-causal half alone.
// Tables per spectrum in pair order: (kfa, kfb, kfs) x {full, causal, anti}.
template <int LOG2M>
static void conv_long_row(const float* u, int L, const float* tw_, const float* twp_, const float* const* kfa_,
                          const float* const* kfb_, const float* const* kfs_, float* out) {
    constexpr int M = 1 << LOG2M, S = M;                 // S real samples = M/2 packed complex points
    const float2 *tw = reinterpret_cast<const float2*>(tw_), *twp = reinterpret_cast<const float2*>(twp_);
    const int nseg = (L + S - 1) / S;
    std::vector<float2> X(M + M / 16), ya(M / 2), yb(M / 2);
    for (int j = 0; j < nseg; ++j) {
        float y0 = 0.f, yM = 0.f;
        float2 yh = make_float2(0.f, 0.f);
        for (int q = 0; q < M / 2; ++q) ya[q] = yb[q] = make_float2(0.f, 0.f);
        for (int t = 0; t < 3; ++t) {                    // source segments j, j-1, j+1 with spectra full, causal', anti'
            const int sj = j + (t == 0 ? 0 : t == 1 ? -1 : 1);
            if (sj < 0 || sj >= nseg) continue;
            const float2 *kfa = reinterpret_cast<const float2*>(kfa_[t]), *kfb = reinterpret_cast<const float2*>(kfb_[t]);
            const float2* kfs = reinterpret_cast<const float2*>(kfs_[t]);
            for (int i = 0; i < M; ++i) {                // packed complex point i = samples (2i, 2i+1) of the segment
                const int s0 = sj * S + 2 * i;
                const float re = (i < S / 2 && s0 < L) ? u[s0] : 0.f, im = (i < S / 2 && s0 + 1 < L) ? u[s0 + 1] : 0.f;
                X[pidx(i)] = make_float2(re, im);
            }
            forward_lds<LOG2M>(X.data(), tw);
            for (int q = 0; q < M / 2; ++q) {
                if (q == 0) {
                    const float2 z0 = X[pidx(0)];
                    y0 += (z0.x + z0.y) * kfs[0].x;
                    yM += (z0.x - z0.y) * kfs[1].x;
                    yh = cadd(yh, cmul_(cconj(X[pidx(1)]), kfs[2]));
                    continue;
                }
                const int p = 2 * q, pm = brev_bits(M - brev_bits(p, LOG2M), LOG2M);
                float2 ak, am;
                pair_bins(X[pidx(p)], X[pidx(pm)], twp[q], ak, am);
                ya[q] = cadd(ya[q], cmul_(ak, kfa[q]));
                yb[q] = cadd(yb[q], cmul_(am, kfb[q]));
            }
        }
        X[pidx(0)] = make_float2(0.5f * (y0 + yM), 0.5f * (y0 - yM));
        X[pidx(1)] = cconj(yh);
        for (int q = 1; q < M / 2; ++q) {
            const int p = 2 * q, pm = brev_bits(M - brev_bits(p, LOG2M), LOG2M);
            pair_repack(ya[q], yb[q], twp[q], X[pidx(p)], X[pidx(pm)]);
        }
        inverse_lds<LOG2M>(X.data(), tw);
        const float scale = 1.f / (float)M;
        for (int i = 0; i < S / 2; ++i) {
            const int s0 = j * S + 2 * i;
            if (s0 < L) out[s0] = X[pidx(i)].x * scale;
            if (s0 + 1 < L) out[s0 + 1] = X[pidx(i)].y * scale;
        }
    }
}

#define DISPATCH(FN, ...)                      \
    switch (log2m) {                           \
        case 6: FN<6>(__VA_ARGS__); return 0;  \
        case 7: FN<7>(__VA_ARGS__); return 0;  \
        case 8: FN<8>(__VA_ARGS__); return 0;  \
        case 10: FN<10>(__VA_ARGS__); return 0; \
        case 11: FN<11>(__VA_ARGS__); return 0; \
        case 12: FN<12>(__VA_ARGS__); return 0; \
        case 13: FN<13>(__VA_ARGS__); return 0; \
        case 14: FN<14>(__VA_ARGS__); return 0; \
    }                                          \
    return -1

extern "C" int dws_host_mirror_block(int t) { return mirror_block(t); }

extern "C" int dws_host_fft(int log2m, float* data, const float* tw, int inverse) { DISPATCH(transform, data, tw, inverse); }

extern "C" int dws_host_conv_row(int log2m, const float* u, int L, const float* tw, const float* twp, const float* kfa,
                                 const float* kfb, const float* kfs, float csign, float* out) {
    DISPATCH(conv_row, u, L, tw, twp, kfa, kfb, kfs, csign, out);
}

// the same row with the fused tail pass (sizes whose plan is even and ends in a radix-4 tail: 2^6, 2^10, 2^14)
extern "C" int dws_host_conv_row_fused(int log2m, const float* u, int L, const float* tw, const float* twp, const float* kfa,
                                       const float* kfb, const float* kfs, float csign, float* out) {
    switch (log2m) {
        case 6: conv_row<6, true>(u, L, tw, twp, kfa, kfb, kfs, csign, out); return 0;
        case 10: conv_row<10, true>(u, L, tw, twp, kfa, kfb, kfs, csign, out); return 0;
        case 14: conv_row<14, true>(u, L, tw, twp, kfa, kfb, kfs, csign, out); return 0;
    }
    return 1;
}

// the separate pair stage with the packed pair arithmetic (`pointwise_pair4`; the kernel's form at every size)
template <int LOG2M>
static void conv_row_x4(const float* u, int L, const float* tw, const float* twp, const float* kfa, const float* kfb,
                        const float* kfs, float csign, float* out) {
    conv_row<LOG2M, false, true>(u, L, tw, twp, kfa, kfb, kfs, csign, out);
}
extern "C" int dws_host_conv_row_x4(int log2m, const float* u, int L, const float* tw, const float* twp, const float* kfa,
                                    const float* kfb, const float* kfs, float csign, float* out) {
    DISPATCH(conv_row_x4, u, L, tw, twp, kfa, kfb, kfs, csign, out);
}

// ... and the fused tail pass with it (sizes 2^6, 2^10, 2^14 in the kernel)
extern "C" int dws_host_conv_row_fused4(int log2m, const float* u, int L, const float* tw, const float* twp, const float* kfa,
                                        const float* kfb, const float* kfs, float csign, float* out) {
    switch (log2m) {
        case 6: conv_row<6, true, true>(u, L, tw, twp, kfa, kfb, kfs, csign, out); return 0;
        case 10: conv_row<10, true, true>(u, L, tw, twp, kfa, kfb, kfs, csign, out); return 0;
        case 14: conv_row<14, true, true>(u, L, tw, twp, kfa, kfb, kfs, csign, out); return 0;
    }
    return 1;
}

extern "C" int dws_host_conv_long_row(int log2m, const float* u, int L, const float* tw, const float* twp,
                                      const float* const* kfa, const float* const* kfb, const float* const* kfs, float* out) {
    DISPATCH(conv_long_row, u, L, tw, twp, kfa, kfb, kfs, out);
}
