// In-LDS complex FFT of size M = 2^LOG2M used by the S4 convolution kernels (fftconv_kernels.hip).
//
// In-place radix-2 decimation-in-frequency forward whose stages are fused four at a time into radix-16 passes held in
// registers (a thread owns 16 points; a pass = one LDS round trip), mirrored decimation-in-time inverse.  The forward
// leaves the spectrum in BIT-REVERSED order, the inverse consumes that order: no reordering pass.  Pass plan for
// E = LOG2M (even; odd sizes first take one radix-2 pass over the top bit):
//      index bits [E-4, E), [E-8, E-4), ... as radix-16 passes, and a final radix-4 pass over bits [0, 2) if E % 4 == 2
//      (E = 14: 4+4+4+2, E = 12: 4+4+4, E = 10: 4+4+2).
// A radix-16 pass over bits [b, b+4) (s = 2^b, a thread's points are base + r*s) is two radix-4 sub-steps; with
// theta = W_{16 s}^j, j = index mod s:  sub-step 1 butterflies (r0, r0+4, r0+8, r0+12) use w1 = theta * W_16^{r0},
// sub-step 2 butterflies (4g .. 4g+3) use w1 = theta^4.  theta comes from the table once per thread and pass.
//
// LDS rows are padded by one complex per 16 (pidx) so the strided and the 16-contiguous access patterns of every pass
// are bank-conflict free for ds_read_b64 / ds_write_b64.
//
// The file compiles for the host too (tests/fft_core_host.cpp emulates a workgroup thread by thread against numpy).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DWS_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define DWS_HD inline
struct float2 {
    float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
#endif

namespace dws {

// (Complex arithmetic on float2 ext-vectors -- one v_pk_add_f32 per complex add, v_pk_mul + v_pk_fma per multiply -- was
// tried for the device side: hipcc does not fold the re/im swaps and single-lane negations into op_sel / neg modifiers
// (1023 packed instructions + 219 v_mov + 132 v_xor + spills against 2276 scalar ones: no fewer issue cycles at the
// measured rates, tools/valu_rate.hip), so the helpers stay componentwise.)
DWS_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
DWS_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
DWS_HD float2 cmul_(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
DWS_HD float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
DWS_HD float2 csqr(float2 a) { return make_float2(a.x * a.x - a.y * a.y, 2.f * a.x * a.y); }
DWS_HD float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
DWS_HD float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
DWS_HD float2 mul_pos_i(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)
DWS_HD int pidx(int i) { return i + (i >> 4); }

// W_16^k = exp(-2 pi i k / 16), k = 0..3 and W_8^k, as compile-time selected constants
template <int K>
DWS_HD float2 w16c() {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r = 0.70710678118654752440f;
    return K == 0 ? make_float2(1.f, 0.f) : K == 1 ? make_float2(c1, -s1) : K == 2 ? make_float2(r, -r) : make_float2(s1, -c1);
}

// Radix-4 butterfly of the in-place DIF (forward) / DIT (inverse) on four points spaced s apart:
// w1 = W_{4s}^j, w2 = w1^2 (forward math as in two fused radix-2 stages: spans 2s then s).
template <bool INV>
DWS_HD void bf4(float2& x0, float2& x1, float2& x2, float2& x3, float2 w1, float2 w2) {
    if (!INV) {
        const float2 a0 = cadd(x0, x2), a2 = cmul_(csub(x0, x2), w1);
        const float2 a1 = cadd(x1, x3), a3 = cmul_(mul_neg_i(csub(x1, x3)), w1);
        x0 = cadd(a0, a1);
        x1 = cmul_(csub(a0, a1), w2);
        x2 = cadd(a2, a3);
        x3 = cmul_(csub(a2, a3), w2);
    } else {
        const float2 v1 = cmulc(x1, w2), v3 = cmulc(x3, w2);
        const float2 a0 = cadd(x0, v1), a1 = csub(x0, v1), a2 = cadd(x2, v3), a3 = csub(x2, v3);
        const float2 b2 = cmulc(a2, w1), b3 = mul_pos_i(cmulc(a3, w1));
        x0 = cadd(a0, b2);
        x2 = csub(a0, b2);
        x1 = cadd(a1, b3);
        x3 = csub(a1, b3);
    }
}

// Forward butterfly whose upper two inputs are zero (the zero padding of the convolution input: x2 = x3 = 0).
DWS_HD void bf4_fwd_zero_hi(float2& x0, float2& x1, float2& x2, float2& x3, float2 w1, float2 w2) {
    const float2 a2 = cmul_(x0, w1), a3 = cmul_(mul_neg_i(x1), w1);
    const float2 a0 = x0, a1 = x1;
    x0 = cadd(a0, a1);
    x1 = cmul_(csub(a0, a1), w2);
    x2 = cadd(a2, a3);
    x3 = cmul_(csub(a2, a3), w2);
}

// Inverse butterfly when only the lower two outputs are needed (the convolution keeps the first half): x2, x3 are left
// undefined.
DWS_HD void bf4_inv_lo_only(float2& x0, float2& x1, float2 x2, float2 x3, float2 w1, float2 w2) {
    const float2 v1 = cmulc(x1, w2), v3 = cmulc(x3, w2);
    const float2 a0 = cadd(x0, v1), a1 = csub(x0, v1), a2 = cadd(x2, v3), a3 = csub(x2, v3);
    x0 = cadd(a0, cmulc(a2, w1));
    x1 = cadd(a1, mul_pos_i(cmulc(a3, w1)));
}

// Unit-twiddle radix-4 butterfly (index bits [0, 2): w1 = w2 = 1).
template <bool INV>
DWS_HD void bf4_unit(float2& x0, float2& x1, float2& x2, float2& x3) {
    if (!INV) {
        const float2 a0 = cadd(x0, x2), a2 = csub(x0, x2), a1 = cadd(x1, x3), a3 = mul_neg_i(csub(x1, x3));
        x0 = cadd(a0, a1);
        x1 = csub(a0, a1);
        x2 = cadd(a2, a3);
        x3 = csub(a2, a3);
    } else {
        const float2 a0 = cadd(x0, x1), a1 = csub(x0, x1), a2 = cadd(x2, x3), a3 = mul_pos_i(csub(x2, x3));
        x0 = cadd(a0, a2);
        x2 = csub(a0, a2);
        x1 = cadd(a1, a3);
        x3 = csub(a1, a3);
    }
}

// Keeps a value opaque to the optimiser (device code): the twiddles derived from a pass's theta are loop invariant over
// the rows a persistent workgroup walks, and hoisting them out of that loop costs ~20 VGPRs per pass for the kernel's
// whole lifetime (spills at 1024 threads); recomputing them per row is ~10 % of a pass's arithmetic.
DWS_HD float2 opaque(float2 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v.x), "+v"(v.y));
#endif
    return v;
}

DWS_HD int opaque(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// Twiddle w1 of sub-step-1 butterfly r0: theta * W_16^{r0} (TW = false: theta = 1).
template <bool TW, int R0>
DWS_HD float2 tw16(float2 theta) {
    if (!TW) return w16c<R0>();
    return R0 == 0 ? theta : cmul_(theta, w16c<R0>());
}

// The four fused stages of a radix-16 pass on registers: x[r] is the point base + r*s, theta = W_{16 s}^j.
// Sub-step 1: butterflies (r0, r0+4, r0+8, r0+12) with w1 = theta W_16^{r0}, w2 = w1^2; sub-step 2: butterflies
// (4g .. 4g+3) with w1 = theta^4, w2 = theta^8.  Twiddles are derived just before their butterfly (few live registers).
// ZERO_HI (forward): x[8..15] are zero on entry.  LO_ONLY (inverse): only x[0..7] are needed on exit.
template <bool INV, bool TW, bool ZERO_HI = false, bool LO_ONLY = false>
DWS_HD void fft16(float2 (&x)[16], float2 theta_in) {
    const float2 theta = TW ? opaque(theta_in) : theta_in;
    const float2 t2 = csqr(theta), t4 = csqr(t2), t8 = csqr(t4);   // dead code when !TW
    // w2 of butterfly r0 is w1^2 = theta^2 W_8^{r0}: {1, (1-i)/sqrt2, -i, -(1+i)/sqrt2} are cheaper than a squaring
    constexpr float RH = 0.70710678118654752440f;
    const float2 t2w[4] = {t2, make_float2(RH * (t2.x + t2.y), RH * (t2.y - t2.x)), mul_neg_i(t2),
                           make_float2(RH * (t2.y - t2.x), -RH * (t2.x + t2.y))};
#define DWS_W2(R0) t2w[R0]
#define DWS_STEP1(R0)                                                                                   \
    {                                                                                                   \
        const float2 w1 = tw16<TW, R0>(theta);                                                          \
        const float2 w2 = TW ? DWS_W2(R0) : (R0 == 0 ? make_float2(1.f, 0.f) : R0 == 1 ? w16c<2>()         \
                                          : R0 == 2 ? make_float2(0.f, -1.f) : mul_neg_i(w16c<2>()));   \
        if (!INV) {                                                                                     \
            if (ZERO_HI) bf4_fwd_zero_hi(x[R0], x[R0 + 4], x[R0 + 8], x[R0 + 12], w1, w2);              \
            else bf4<false>(x[R0], x[R0 + 4], x[R0 + 8], x[R0 + 12], w1, w2);                           \
        } else {                                                                                        \
            if (LO_ONLY) bf4_inv_lo_only(x[R0], x[R0 + 4], x[R0 + 8], x[R0 + 12], w1, w2);              \
            else bf4<true>(x[R0], x[R0 + 4], x[R0 + 8], x[R0 + 12], w1, w2);                            \
        }                                                                                               \
    }
#define DWS_STEP2(G)                                                                                    \
    {                                                                                                   \
        if (TW) bf4<INV>(x[4 * G], x[4 * G + 1], x[4 * G + 2], x[4 * G + 3], t4, t8);                   \
        else bf4_unit<INV>(x[4 * G], x[4 * G + 1], x[4 * G + 2], x[4 * G + 3]);                         \
    }
    if (!INV) {
        DWS_STEP1(0) DWS_STEP1(1) DWS_STEP1(2) DWS_STEP1(3)
        DWS_STEP2(0) DWS_STEP2(1) DWS_STEP2(2) DWS_STEP2(3)
    } else {
        DWS_STEP2(0) DWS_STEP2(1) DWS_STEP2(2) DWS_STEP2(3)
        DWS_STEP1(0) DWS_STEP1(1) DWS_STEP1(2) DWS_STEP1(3)
    }
#undef DWS_STEP1
#undef DWS_STEP2
#undef DWS_W2
}

// Pass plan of size 2^LOG2M with THREADS = M/16 (one 16-point group per thread and pass).
template <int LOG2M>
struct FftPlan {
    static constexpr int M = 1 << LOG2M;
    static constexpr bool ODD = (LOG2M & 1) != 0;            // one radix-2 pass over the top bit first
    static constexpr int E = LOG2M - (ODD ? 1 : 0);          // bits handled by the radix-16 / radix-4 passes
    static constexpr int N16 = E / 4;                        // radix-16 passes, pass p over bits [E-4(p+1), E-4p)
    static constexpr bool TAIL4 = (E % 4) == 2;              // final radix-4 pass over bits [0, 2)
    static constexpr int b0(int p) { return E - 4 * (p + 1); }
};

// theta of every radix-16 pass for the 16-point groups g = tid + i*THREADS (i < NG) this thread owns: j = g mod s.
template <int LOG2M, int NG = 1>
struct FftTw {
    using P = FftPlan<LOG2M>;
    float2 theta[P::N16 > 0 ? P::N16 : 1][NG];
    DWS_HD void load(const float2* __restrict__ tw, int tid) {
        constexpr int THREADS = (P::M / 16) / NG;
#pragma unroll
        for (int p = 0; p < P::N16; ++p) {
            const int b = P::b0(p);
#pragma unroll
            for (int i = 0; i < NG; ++i)
                theta[p][i] = (b == 0) ? make_float2(1.f, 0.f)
                                       : tw[((tid + i * THREADS) & ((1 << b) - 1)) * (P::M >> (b + 4))];
        }
    }
};

// Point r of the 16-point group `g` of the pass over bits [B0, B0+4): padded LDS index.
template <int B0>
DWS_HD int group_base(int g) {
    return ((g >> B0) << (B0 + 4)) + (g & ((1 << B0) - 1));
}

// One radix-16 pass LDS -> LDS (group index g = tid when THREADS = M/16).
template <int LOG2M, int B0, bool INV>
DWS_HD void pass16_lds(float2* __restrict__ X, float2 theta, int g) {
    constexpr int S = 1 << B0;
    const int base = group_base<B0>(g);
    float2 x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = X[pidx(base + r * S)];
    fft16<INV, (B0 != 0)>(x, theta);
#pragma unroll
    for (int r = 0; r < 16; ++r) X[pidx(base + r * S)] = x[r];
}

// Final (forward) / first (inverse) radix-4 pass over bits [0, 2): thread g owns points 16g .. 16g+15.
template <bool INV>
DWS_HD void pass4_lds(float2* __restrict__ X, int g) {
    float2 x[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) x[d] = X[17 * g + d];  // pidx(16 g + d) = 17 g + d
#pragma unroll
    for (int q = 0; q < 4; ++q) bf4_unit<INV>(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
#pragma unroll
    for (int d = 0; d < 16; ++d) X[17 * g + d] = x[d];
}

// Radix-2 pass over the top bit (odd LOG2M only): `it` runs over M/2 butterflies.
template <int LOG2M, bool INV>
DWS_HD void pass2_top(float2* __restrict__ X, const float2* __restrict__ tw, int t) {
    constexpr int h = (1 << LOG2M) / 2;
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

DWS_HD int brev_bits(int k, int bits) {
    unsigned v = (unsigned)k, r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return (int)r;
}

// Pointwise stage of the real-input convolution in bit-reversed order for pair q (0 < q < M/2): positions p = 2q
// (k = brev(p) < M/2) and pm = position of M - k.  With N = 2M, Wk = exp(-2 pi i k / N):
//   Xe = (Zk + conj Zm)/2, Xo = -(i/2)(Zk - conj Zm), t = Wk Xo;  A[k] = Xe + t, A[M-k] = conj(Xe - t);  Y = A * Kf
//   Ye = (Yk + conj Ym)/2, Yo = (Yk - conj Ym)/2 * conj(Wk);  Zy[k] = Ye + i Yo, Zy[M-k] = conj(Ye - i Yo)
// csign = -1 multiplies by conj(K_f): the adjoint (correlation) of the convolution.
DWS_HD void pointwise_pair(float2& zk_io, float2& zm_io, float2 wk, float2 ka, float2 kb, float csign) {
    const float2 zk = zk_io, zm = zm_io;
    const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const float2 d = make_float2(zk.x - zm.x, zk.y + zm.y);           // Zk - conj Zm
    const float2 xo = make_float2(0.5f * d.y, -0.5f * d.x);           // -(i/2) d
    const float2 t = cmul_(wk, xo);
    const float2 ak = cadd(xe, t), am = cconj(csub(xe, t));
    const float2 yk = cmul_(ak, make_float2(ka.x, csign * ka.y)), ym = cmul_(am, make_float2(kb.x, csign * kb.y));
    const float2 ye = make_float2(0.5f * (yk.x + ym.x), 0.5f * (yk.y - ym.y));
    const float2 e = make_float2(0.5f * (yk.x - ym.x), 0.5f * (yk.y + ym.y));  // (Yk - conj Ym)/2
    const float2 iyo = mul_pos_i(cmulc(e, wk));
    zk_io = cadd(ye, iyo);
    zm_io = cconj(csub(ye, iyo));
}

// The two halves of pointwise_pair on their own (kernels that accumulate several spectra between them):
// bins A[k], A[M-k] of the real row from the packed spectrum, and the packed form of a real-row spectrum Y[k], Y[M-k].
DWS_HD void pair_bins(float2 zk, float2 zm, float2 wk, float2& ak, float2& am) {
    const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const float2 d = make_float2(zk.x - zm.x, zk.y + zm.y);
    const float2 t = cmul_(wk, make_float2(0.5f * d.y, -0.5f * d.x));
    ak = cadd(xe, t);
    am = cconj(csub(xe, t));
}
DWS_HD void pair_repack(float2 yk, float2 ym, float2 wk, float2& zk, float2& zm) {
    const float2 ye = make_float2(0.5f * (yk.x + ym.x), 0.5f * (yk.y - ym.y));
    const float2 e = make_float2(0.5f * (yk.x - ym.x), 0.5f * (yk.y + ym.y));
    const float2 iyo = mul_pos_i(cmulc(e, wk));
    zk = cadd(ye, iyo);
    zm = cconj(csub(ye, iyo));
}

// q = 0: k = 0 (self-paired, carries DC and Nyquist, both real) and k = M/2 (position 1, self-paired).
DWS_HD void pointwise_self(float2& z0, float2& z1, float2 kf0, float2 kfM, float2 kfh, float csign) {
    const float y0 = (z0.x + z0.y) * kf0.x;   // A[0] = Re + Im; irfft ignores Im of DC / Nyquist
    const float ym = (z0.x - z0.y) * kfM.x;   // A[M] = Re - Im
    z0 = make_float2(0.5f * (y0 + ym), 0.5f * (y0 - ym));
    z1 = cmulc(z1, make_float2(kfh.x, csign * kfh.y));  // Zy[M/2] = Z[M/2] * conj(Kf[M/2])
}

}  // namespace dws
