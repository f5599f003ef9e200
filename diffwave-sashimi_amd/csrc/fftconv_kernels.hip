// S4 FFT long convolution (`models/s4.py:1391-1430`) as ONE kernel per block:
// a workgroup owns one (b, h) row, keeps it in LDS for the whole
//   real FFT -> multiply by the cached kernel spectrum -> inverse real FFT -> + D*u -> GELU
// chain, so the row is read once and written once (8*L bytes) instead of the
// ~72*L bytes of the rocFFT r2c / multiply / c2r / post sequence.
//
// Transform size: the reference uses n = 2L (2^k * 5^3) and relies on circular wrap at
// exactly 2L (anti-causal taps at indices L..2L-1, s4.py:1393-1394).  Here n = Nf is the next
// power of two >= 2L with the anti-causal half re-placed at Nf-L..Nf-1 and u zero-padded to Nf;
// the first L outputs are identical up to rounding (alias-free iff n >= 2L).
//
// Real FFT of size Nf via a complex FFT of size M = Nf/2 on z[j] = u[2j] + i u[2j+1] (the row
// reinterpreted as float2).  In-place radix-2 decimation-in-frequency forward (fused in pairs
// = radix-4 passes through LDS, the lowest four bits as a 16-point transform in registers)
// leaves the spectrum in BIT-REVERSED order; the pointwise stage works in that order and a
// mirrored decimation-in-time inverse brings natural order back -- no reordering pass at all.
// LDS rows are padded by one complex per 16 so both the strided and the 16-contiguous access
// patterns are bank-conflict free for ds_read_b64 / ds_write_b64.
#include "fftconv.h"

namespace dws {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul_(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 mul_pos_i(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)
__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }
__device__ __forceinline__ float gelu_f(float x) { return dws_gelu(x); }

// W_16^k = exp(-2 pi i k / 16), k = 0..7
__device__ __forceinline__ float2 w16(int k) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r = 0.70710678118654752440f;
    switch (k & 7) {
        case 0: return make_float2(1.f, 0.f);
        case 1: return make_float2(c1, -s1);
        case 2: return make_float2(r, -r);
        case 3: return make_float2(s1, -c1);
        case 4: return make_float2(0.f, -1.f);
        case 5: return make_float2(-s1, -c1);
        case 6: return make_float2(-r, -r);
        default: return make_float2(-c1, -s1);
    }
}

// 16-point transforms on registers (the lowest four index bits).
template <bool INV>
__device__ __forceinline__ void fft16_regs(float2 (&x)[16]) {
    if (!INV) {  // DIF: spans 8,4,2,1 ; x[i+h] = (u - v) * W_{2h}^{i mod h}
#pragma unroll
        for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((i & h) == 0) {
                    const float2 u = x[i], v = x[i + h];
                    x[i] = cadd(u, v);
                    x[i + h] = cmul_(csub(u, v), w16((i & (h - 1)) * (8 / h)));
                }
            }
        }
    } else {  // DIT: spans 1,2,4,8 ; v = x[i+h] * conj(W_{2h}^{i mod h})
#pragma unroll
        for (int h = 1; h <= 8; h <<= 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((i & h) == 0) {
                    const float2 u = x[i], v = cmulc(x[i + h], w16((i & (h - 1)) * (8 / h)));
                    x[i] = cadd(u, v);
                    x[i + h] = csub(u, v);
                }
            }
        }
    }
}

// One fused radix-4 pass over index bits (log2(s)+1, log2(s)).  Butterfly t uses W_{4s}^j, j = t mod s;
// the caller supplies it from registers (w[i] for this thread's i-th butterfly): the twiddles of ALL
// passes are fetched once at kernel start (radix4_twiddles) instead of one L2 round trip per pass.
template <int LOG2M, int THREADS, bool INV, bool TOPPASS>
__device__ __forceinline__ void radix4_pass(float2* __restrict__ X, const float2* __restrict__ tw, float2 wreg,
                                            int log2s, int tid) {
    constexpr int M = 1 << LOG2M;
    const int s = 1 << log2s;
    const int twstep = M >> (log2s + 2);  // W_{4s}^j = tw[j * M/(4s)]
#pragma unroll 1
    for (int t = tid; t < M / 4; t += THREADS) {
        const int j = t & (s - 1);
        const int base = ((t >> log2s) << (log2s + 2)) + j;
        const float2 w1 = TOPPASS ? tw[j * twstep] : wreg;   // narrower passes: j = tid mod s, fetched at kernel start
        const float2 w2 = cmul_(w1, w1);
        float2 x0 = X[pidx(base)], x1 = X[pidx(base + s)], x2 = X[pidx(base + 2 * s)], x3 = X[pidx(base + 3 * s)];
        if (!INV) {
            // stage h = 2s: pairs (0,2) tw w1, (1,3) tw w1 * (-i); stage h = s: pairs (0,1), (2,3) tw w2
            const float2 a0 = cadd(x0, x2), a2 = cmul_(csub(x0, x2), w1);
            const float2 a1 = cadd(x1, x3), a3 = cmul_(mul_neg_i(csub(x1, x3)), w1);
            x0 = cadd(a0, a1);
            x1 = cmul_(csub(a0, a1), w2);
            x2 = cadd(a2, a3);
            x3 = cmul_(csub(a2, a3), w2);
        } else {
            // stage h = s first (conj w2), then h = 2s (conj w1, and +i for the odd pair)
            const float2 v1 = cmulc(x1, w2), v3 = cmulc(x3, w2);
            const float2 a0 = cadd(x0, v1), a1 = csub(x0, v1), a2 = cadd(x2, v3), a3 = csub(x2, v3);
            const float2 b2 = cmulc(a2, w1), b3 = mul_pos_i(cmulc(a3, w1));
            x0 = cadd(a0, b2);
            x2 = csub(a0, b2);
            x1 = cadd(a1, b3);
            x3 = csub(a1, b3);
        }
        X[pidx(base)] = x0; X[pidx(base + s)] = x1; X[pidx(base + 2 * s)] = x2; X[pidx(base + 3 * s)] = x3;
    }
}

// Register-resident twiddles of the narrower radix-4 passes.  With THREADS = M/16 a thread owns
// butterflies t = tid + i*THREADS (i < 4).  In the widest pass (s = M/4 or M/8) j = t mod s differs per
// i and is read from the table in the pass; in every narrower pass s <= THREADS, so j = tid mod s for
// all i -> ONE value per pass, fetched once at kernel start and reused by forward and inverse.
template <int LOG2M, int THREADS>
struct FftTw {
    static constexpr int M = 1 << LOG2M;
    static constexpr bool ODD = ((LOG2M - 4) & 1) != 0;
    static constexpr int TOP = LOG2M - (ODD ? 3 : 2);         // log2(s) of the widest radix-4 pass
    static constexpr int NLOW = (TOP - 4) / 2;                // narrower passes: log2s = TOP-2, ..., 4
    static constexpr int NBF = M / 4 / THREADS;
    float2 wlow[NLOW > 0 ? NLOW : 1];
    __device__ __forceinline__ void load(const float2* __restrict__ tw, int tid) {
#pragma unroll
        for (int p = 0; p < NLOW; ++p) {
            const int log2s = TOP - 2 * (p + 1);
            wlow[p] = tw[(tid & ((1 << log2s) - 1)) * (M >> (log2s + 2))];
        }
    }
};

template <int LOG2M, int THREADS, bool INV>
__device__ __forceinline__ void radix2_top_pass(float2* __restrict__ X, const float2* __restrict__ tw, int tid) {
    constexpr int M = 1 << LOG2M, h = M / 2;
    for (int t = tid; t < h; t += THREADS) {
        const float2 u = X[pidx(t)], v = X[pidx(t + h)];
        if (!INV) {
            X[pidx(t)] = cadd(u, v);
            X[pidx(t + h)] = cmul_(csub(u, v), tw[t]);
        } else {
            const float2 vv = cmulc(v, tw[t]);
            X[pidx(t)] = cadd(u, vv);
            X[pidx(t + h)] = csub(u, vv);
        }
    }
}

template <int LOG2M, int THREADS, bool INV>
__device__ __forceinline__ void low16_pass(float2* __restrict__ X, int tid) {
    constexpr int M = 1 << LOG2M;
    for (int t = tid; t < M / 16; t += THREADS) {
        float2 x[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) x[d] = X[17 * t + d];  // pidx(16 t + d) = 17 t + d
        fft16_regs<INV>(x);
#pragma unroll
        for (int d = 0; d < 16; ++d) X[17 * t + d] = x[d];
    }
}

template <int LOG2M, int THREADS>
__device__ __forceinline__ void fft_forward(float2* X, const float2* tw, const FftTw<LOG2M, THREADS>& W, int tid) {
    using F = FftTw<LOG2M, THREADS>;
    static_assert(F::NBF * THREADS * 4 == (1 << LOG2M), "one 16-point group per thread");
    if (F::ODD) {
        radix2_top_pass<LOG2M, THREADS, false>(X, tw, tid);
        __syncthreads();
    }
    radix4_pass<LOG2M, THREADS, false, true>(X, tw, make_float2(0.f, 0.f), F::TOP, tid);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < F::NLOW; ++p) {
        radix4_pass<LOG2M, THREADS, false, false>(X, tw, W.wlow[p], F::TOP - 2 * (p + 1), tid);
        __syncthreads();
    }
    low16_pass<LOG2M, THREADS, false>(X, tid);
    __syncthreads();
}

template <int LOG2M, int THREADS>
__device__ __forceinline__ void fft_inverse(float2* X, const float2* tw, const FftTw<LOG2M, THREADS>& W, int tid) {
    using F = FftTw<LOG2M, THREADS>;
    low16_pass<LOG2M, THREADS, true>(X, tid);
    __syncthreads();
#pragma unroll
    for (int p = F::NLOW - 1; p >= 0; --p) {
        radix4_pass<LOG2M, THREADS, true, false>(X, tw, W.wlow[p], F::TOP - 2 * (p + 1), tid);
        __syncthreads();
    }
    radix4_pass<LOG2M, THREADS, true, true>(X, tw, make_float2(0.f, 0.f), F::TOP, tid);
    __syncthreads();
    if (F::ODD) {
        radix2_top_pass<LOG2M, THREADS, true>(X, tw, tid);
        __syncthreads();
    }
}

__device__ __forceinline__ int brev(int k, int bits) { return (int)(__brev((unsigned)k) >> (32 - bits)); }

// Pointwise stage in bit-reversed order.  Pair q <-> positions p = 2q (k = brev(p) < M/2) and the
// position of M - k.  With N = 2M, Wk = exp(-2 pi i k / N):
//   Xe = (Zk + conj Zm)/2, Xo = -(i/2)(Zk - conj Zm), t = Wk Xo
//   A[k] = Xe + t, A[M-k] = conj(Xe - t);  Y = A * Kf
//   Ye = (Yk + conj Ym)/2, Yo = (Yk - conj Ym)/2 * conj(Wk);  Zy[k] = Ye + i Yo, Zy[M-k] = conj(Ye - i Yo)
template <int LOG2M, int THREADS>
__device__ __forceinline__ void pointwise_pairs(float2* __restrict__ X, const float2* __restrict__ twp,
                                                const float2* __restrict__ kfa, const float2* __restrict__ kfb,
                                                const float2* __restrict__ kfs, int tid, float csign) {
    constexpr int M = 1 << LOG2M;
#pragma unroll 1
    for (int it = 0; it < M / 2 / THREADS; ++it) {
        const int q = tid + it * THREADS;
        const int p = 2 * q;
        if (q == 0) {
            // k = 0 (self-paired, carries DC and Nyquist) and k = M/2 (position 1, self-paired)
            const float2 z0 = X[pidx(0)];
            const float y0 = (z0.x + z0.y) * kfs[0].x;   // A[0] = Re+Im, real; irfft ignores Im of DC / Nyquist
            const float ym = (z0.x - z0.y) * kfs[1].x;   // A[M] = Re-Im
            X[pidx(0)] = make_float2(0.5f * (y0 + ym), 0.5f * (y0 - ym));
            X[pidx(1)] = cmulc(X[pidx(1)], make_float2(kfs[2].x, csign * kfs[2].y));  // Zy[M/2] = Z[M/2] * conj(Kf[M/2])
            continue;
        }
        const int k = brev(p, LOG2M);
        const int pm = brev(M - k, LOG2M);
        const float2 zk = X[pidx(p)], zm = X[pidx(pm)];
        const float2 wk = twp[q];
        const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 d = make_float2(zk.x - zm.x, zk.y + zm.y);           // Zk - conj Zm
        const float2 xo = make_float2(0.5f * d.y, -0.5f * d.x);           // -(i/2) d
        const float2 t = cmul_(wk, xo);
        const float2 ak = cadd(xe, t), am = cconj(csub(xe, t));
        const float2 ka = kfa[q], kb = kfb[q];   // csign = -1: conj(K_f), the adjoint of the convolution
        const float2 yk = cmul_(ak, make_float2(ka.x, csign * ka.y)), ym = cmul_(am, make_float2(kb.x, csign * kb.y));
        const float2 ye = make_float2(0.5f * (yk.x + ym.x), 0.5f * (yk.y - ym.y));
        const float2 e = make_float2(0.5f * (yk.x - ym.x), 0.5f * (yk.y + ym.y));  // (Yk - conj Ym)/2
        const float2 yo = cmulc(e, wk);
        const float2 iyo = mul_pos_i(yo);
        X[pidx(p)] = cadd(ye, iyo);
        X[pidx(pm)] = cconj(csub(ye, iyo));
    }
}

// g[b,h,:] = GELU(conv(u, K_h)[:L] + D[h] * u)
template <int LOG2M, int THREADS>
__global__ __launch_bounds__(THREADS) void fftconv_kernel(FftConvArgs a) {
    constexpr int M = 1 << LOG2M;
    extern __shared__ __attribute__((aligned(16))) float2 X[];  // M + M/16 complex
    const int tid = threadIdx.x;
    // rows of one channel h are adjacent in the launch order (the kernel spectrum is shared by the batch)
    const int row = blockIdx.x;
    const int h = row / a.B, b = row % a.B;
    const int L = a.L, Lc = L / 2;  // L even
    const float2* __restrict__ u2 = reinterpret_cast<const float2*>(a.u + ((size_t)b * a.H + h) * L);
    FftTw<LOG2M, THREADS> W;
    W.load(a.tw, tid);
    for (int j = tid; j < M; j += THREADS) X[pidx(j)] = (j < Lc) ? u2[j] : make_float2(0.f, 0.f);
    __syncthreads();
    fft_forward<LOG2M, THREADS>(X, a.tw, W, tid);
    pointwise_pairs<LOG2M, THREADS>(X, a.twp, a.kfa + (size_t)h * (M / 2), a.kfb + (size_t)h * (M / 2),
                                    a.kfs + (size_t)h * 3, tid, a.conj_k ? -1.f : 1.f);
    __syncthreads();
    fft_inverse<LOG2M, THREADS>(X, a.tw, W, tid);
    const float scale = 1.f / (float)M, Dh = a.D[h];
    float2* __restrict__ g2 = reinterpret_cast<float2*>(a.g + ((size_t)b * a.H + h) * L);
    float2* __restrict__ p2 = a.pre ? reinterpret_cast<float2*>(a.pre + ((size_t)b * a.H + h) * L) : nullptr;
    for (int j = tid; j < Lc; j += THREADS) {
        const float2 y = X[pidx(j)], uu = u2[j];
        const float2 v = make_float2(fmaf(y.x, scale, Dh * uu.x), fmaf(y.y, scale, Dh * uu.y));
        if (p2) p2[j] = v;
        g2[j] = a.no_act ? v : make_float2(gelu_f(v.x), gelu_f(v.y));
    }
}

// Kernel-gradient partials (FftCorrArgs): the block owns channel h and a chunk of the batch; per sample it
// transforms u then dA through the same LDS FFT and accumulates conj(U) * dA for the bins its threads own.
template <int LOG2M, int THREADS>
__global__ __launch_bounds__(THREADS) void fftcorr_kernel(FftCorrArgs a) {
    constexpr int M = 1 << LOG2M, NP = M / 2 / THREADS;
    extern __shared__ __attribute__((aligned(16))) float2 X[];
    const int tid = threadIdx.x, h = blockIdx.x, bs = blockIdx.y;
    const int L = a.L, Lc = L / 2;
    FftTw<LOG2M, THREADS> W;
    W.load(a.tw, tid);
    float2 ua[NP], ub[NP], pa[NP], pb[NP];
    float u0 = 0.f, uM = 0.f, p0 = 0.f, pM = 0.f;
    float2 uh = make_float2(0.f, 0.f), ph = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NP; ++i) pa[i] = pb[i] = make_float2(0.f, 0.f);
    const int b_end = min(a.B, (bs + 1) * a.bchunk);
    // real-FFT bins of pair q from the bit-reversed half-size spectrum (see pointwise_pairs)
    auto bins = [&](int q, float2& ak, float2& am) {
        const int p = 2 * q;
        const int k = brev(p, LOG2M);
        const int pm = brev(M - k, LOG2M);
        const float2 zk = X[pidx(p)], zm = X[pidx(pm)];
        const float2 wk = a.twp[q];
        const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 d = make_float2(zk.x - zm.x, zk.y + zm.y);
        const float2 t = cmul_(wk, make_float2(0.5f * d.y, -0.5f * d.x));
        ak = cadd(xe, t);
        am = cconj(csub(xe, t));
    };
#pragma unroll 1
    for (int b = bs * a.bchunk; b < b_end; ++b) {
        const float2* __restrict__ u2 = reinterpret_cast<const float2*>(a.u + ((size_t)b * a.H + h) * L);
        const float2* __restrict__ d2 = reinterpret_cast<const float2*>(a.da + ((size_t)b * a.H + h) * L);
        for (int j = tid; j < M; j += THREADS) X[pidx(j)] = (j < Lc) ? u2[j] : make_float2(0.f, 0.f);
        __syncthreads();
        fft_forward<LOG2M, THREADS>(X, a.tw, W, tid);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * THREADS;
            if (q == 0) {
                const float2 z0 = X[pidx(0)];
                u0 = z0.x + z0.y; uM = z0.x - z0.y; uh = cconj(X[pidx(1)]);   // A[0], A[M], A[M/2]
            } else {
                bins(q, ua[i], ub[i]);
            }
        }
        __syncthreads();
        for (int j = tid; j < M; j += THREADS) X[pidx(j)] = (j < Lc) ? d2[j] : make_float2(0.f, 0.f);
        __syncthreads();
        fft_forward<LOG2M, THREADS>(X, a.tw, W, tid);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * THREADS;
            if (q == 0) {
                const float2 z0 = X[pidx(0)];
                p0 = fmaf(u0, z0.x + z0.y, p0);
                pM = fmaf(uM, z0.x - z0.y, pM);
                ph = cadd(ph, cmulc(cconj(X[pidx(1)]), uh));
            } else {
                float2 dk, dm;
                bins(q, dk, dm);
                pa[i] = cadd(pa[i], cmulc(dk, ua[i]));   // dA * conj(U)
                pb[i] = cadd(pb[i], cmulc(dm, ub[i]));
            }
        }
        __syncthreads();
    }
    float2* __restrict__ o = a.part + ((size_t)bs * a.H + h) * (M + 1);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = tid + i * THREADS;
        if (q == 0) {
            o[0] = make_float2(p0, 0.f);
            o[M] = make_float2(pM, 0.f);
            o[M / 2] = ph;
        } else {
            const int k = brev(2 * q, LOG2M);
            o[k] = pa[i];
            o[M - k] = pb[i];
        }
    }
}

// Forward real FFT only (weight time): spectrum of the re-placed two-sided kernel, natural order
// out[h][k], k = 0..M (Nf/2+1 bins).  Used to build K_f with the SAME transform the convolution uses.
template <int LOG2M, int THREADS>
__global__ __launch_bounds__(THREADS) void rfft_rows_kernel(const float* __restrict__ in, float2* __restrict__ out,
                                                            const float2* __restrict__ tw,
                                                            const float2* __restrict__ twn) {
    constexpr int M = 1 << LOG2M;
    extern __shared__ __attribute__((aligned(16))) float2 X[];
    const int tid = threadIdx.x, h = blockIdx.x;
    const float2* __restrict__ r2 = reinterpret_cast<const float2*>(in + (size_t)h * 2 * M);
    FftTw<LOG2M, THREADS> W;
    W.load(tw, tid);
    for (int j = tid; j < M; j += THREADS) X[pidx(j)] = r2[j];
    __syncthreads();
    fft_forward<LOG2M, THREADS>(X, tw, W, tid);
    float2* __restrict__ o = out + (size_t)h * (M + 1);
    for (int k = tid; k <= M / 2; k += THREADS) {
        if (k == 0) {
            const float2 z0 = X[pidx(0)];
            o[0] = make_float2(z0.x + z0.y, 0.f);
            o[M] = make_float2(z0.x - z0.y, 0.f);
            continue;
        }
        const float2 zk = X[pidx(brev(k, LOG2M))], zm = X[pidx(brev(M - k, LOG2M))];
        const float2 wk = twn[k];  // exp(-2 pi i k / (2M))
        const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 d = make_float2(zk.x - zm.x, zk.y + zm.y);
        const float2 t = cmul_(wk, make_float2(0.5f * d.y, -0.5f * d.x));
        o[k] = cadd(xe, t);
        o[M - k] = cconj(csub(xe, t));
    }
}

// Two-sided kernel re-placed for transform size Nf >= 2L (s4.py:1391-1394 generalised):
//   K[j] = k0[j]/L (j < L);  K[Nf - m] = k1[m-1]/L (m = 1..L);  0 elsewhere.
// k rows have length Lk (the kernel's own length); the first Lt = min(run length, Lk) taps of each direction are used
// (`L_kernel`, s4.py:1387,805); 1/Lk is the irfft normalisation the unnormalised C2R left out.
__global__ void s4_twosided_pow2_kernel(const float* __restrict__ k, float* __restrict__ K, int H, int Lt, int Nf, int Lk) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (j >= Nf) return;
    const float inv = 1.f / (float)Lk;
    float v = 0.f;
    if (j < Lt) v = k[(size_t)h * Lk + j] * inv;
    else if (j >= Nf - Lt) v = k[((size_t)H + h) * Lk + (Nf - j - 1)] * inv;
    K[(size_t)h * Nf + j] = v;
}

// Pair-ordered copies of the spectrum for the pointwise stage: q -> k = brev(2q):
//   kfa[h][q] = Kf[h][k], kfb[h][q] = Kf[h][M-k];  kfs[h] = {Kf[0], Kf[M], Kf[M/2]}
__global__ void kf_permute_kernel(const float2* __restrict__ kf, float2* __restrict__ kfa, float2* __restrict__ kfb,
                                  float2* __restrict__ kfs, int log2m) {
    const int M = 1 << log2m;
    const int q = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y;
    if (q >= M / 2) return;
    const float2* r = kf + (size_t)h * (M + 1);
    if (q == 0) {
        kfs[h * 3 + 0] = r[0]; kfs[h * 3 + 1] = r[M]; kfs[h * 3 + 2] = r[M / 2];
        kfa[(size_t)h * (M / 2)] = r[0]; kfb[(size_t)h * (M / 2)] = r[M];
        return;
    }
    const int k = brev(2 * q, log2m);
    kfa[(size_t)h * (M / 2) + q] = r[k];
    kfb[(size_t)h * (M / 2) + q] = r[M - k];
}

template <int LOG2M>
struct FcCfg {
    static constexpr int M = 1 << LOG2M;
    static constexpr int THREADS = M / 16;   // one 16-point group / four radix-4 butterflies per thread
    static constexpr size_t LDS = (size_t)(M + M / 16) * 8;
};

bool fftconv_supported(int L, int* log2m) {
    if (L < 16 || (L & 1)) return false;
    int lg = 4;
    while ((1 << lg) < L) ++lg;  // M = Nf/2 >= L  <=>  Nf >= 2L
    if (lg < 10) lg = 10;               // smaller rows still use M = 1024 (one wave); the padding is zeros
    if (lg > 14) return false;
    if (log2m) *log2m = lg;
    return true;
}

template <int LOG2M>
static int launch_fc(const FftConvArgs& a, hipStream_t s) {
    using C = FcCfg<LOG2M>;
    auto kern = fftconv_kernel<LOG2M, C::THREADS>;
    static bool attr = false;
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.B * a.H), dim3(C::THREADS), C::LDS, s, a);
    return DWS_OK;
}

template <int LOG2M>
static int launch_fcorr(const FftCorrArgs& a, hipStream_t s) {
    using C = FcCfg<LOG2M>;
    auto kern = fftcorr_kernel<LOG2M, C::THREADS>;
    static bool attr = false;
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.H, ceil_div(a.B, a.bchunk)), dim3(C::THREADS), C::LDS, s, a);
    return DWS_OK;
}

template <int LOG2M>
static int launch_rf(const float* in, float* out, const float* tw, const float* twn, int H, hipStream_t s) {
    using C = FcCfg<LOG2M>;
    auto kern = rfft_rows_kernel<LOG2M, C::THREADS>;
    static bool attr = false;
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(H), dim3(C::THREADS), C::LDS, s, in, (float2*)out, (const float2*)tw,
                       (const float2*)twn);
    return DWS_OK;
}

#define DWS_FC_DISPATCH(FN, ...)                                                        \
    switch (log2m) {                                                                    \
        case 10: return FN<10>(__VA_ARGS__);                                            \
        case 11: return FN<11>(__VA_ARGS__);                                            \
        case 12: return FN<12>(__VA_ARGS__);                                            \
        case 13: return FN<13>(__VA_ARGS__);                                            \
        case 14: return FN<14>(__VA_ARGS__);                                            \
    }                                                                                   \
    return set_error(DWS_ERR_UNSUPPORTED, "fftconv: log2(M)=%d not instantiated", log2m)

int launch_fftconv(int log2m, const FftConvArgs& a, hipStream_t s) {
    ProfileScope ps("fftconv", s);
    DWS_FC_DISPATCH(launch_fc, a, s);
}

int launch_fftcorr(int log2m, const FftCorrArgs& a, hipStream_t s) {
    ProfileScope ps("fftcorr", s);
    DWS_FC_DISPATCH(launch_fcorr, a, s);
}

int launch_rfft_rows(int log2m, const float* in, float* out, const float* tw, const float* twn, int H,
                     hipStream_t s) {
    DWS_FC_DISPATCH(launch_rf, in, out, tw, twn, H, s);
}

int launch_s4_twosided_pow2(const float* k, float* K, int H, int Lt, int Nf, int Lk, hipStream_t s) {
    hipLaunchKernelGGL(s4_twosided_pow2_kernel, dim3(ceil_div(Nf, 256), H), dim3(256), 0, s, k, K, H, Lt, Nf, Lk);
    return DWS_OK;
}

int launch_kf_permute(const float* kf, float* kfa, float* kfb, float* kfs, int H, int log2m, hipStream_t s) {
    const int M = 1 << log2m;
    hipLaunchKernelGGL(kf_permute_kernel, dim3(ceil_div(M / 2, 256), H), dim3(256), 0, s, (const float2*)kf,
                       (float2*)kfa, (float2*)kfb, (float2*)kfs, log2m);
    return DWS_OK;
}

// Host-built twiddle tables (double precision, rounded once):
//   tw[k]  = exp(-2 pi i k / M),  k < M/2        (FFT passes)
//   twn[k] = exp(-2 pi i k / 2M), k <= M/2       (real-FFT split, natural order)
//   twp[q] = twn[brev(2q)],       q < M/2        (real-FFT split, pair order of the pointwise stage)
void build_fft_tables(int log2m, std::vector<float>& tw, std::vector<float>& twn, std::vector<float>& twp) {
    const int M = 1 << log2m;
    const double PI = 3.14159265358979323846;
    tw.resize((size_t)M); twn.resize((size_t)M + 2); twp.resize((size_t)M);
    for (int k = 0; k < M / 2; ++k) {
        tw[2 * k] = (float)std::cos(-2.0 * PI * k / M);
        tw[2 * k + 1] = (float)std::sin(-2.0 * PI * k / M);
    }
    for (int k = 0; k <= M / 2; ++k) {
        twn[2 * k] = (float)std::cos(-PI * k / M);
        twn[2 * k + 1] = (float)std::sin(-PI * k / M);
    }
    for (int q = 0; q < M / 2; ++q) {
        unsigned p = 2u * (unsigned)q, k = 0;
        for (int bit = 0; bit < log2m; ++bit) k |= ((p >> bit) & 1u) << (log2m - 1 - bit);
        twp[2 * q] = (float)std::cos(-PI * (double)k / M);
        twp[2 * q + 1] = (float)std::sin(-PI * (double)k / M);
    }
}

}  // namespace dws
