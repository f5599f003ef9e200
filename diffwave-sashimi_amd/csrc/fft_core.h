// In-LDS complex FFT of size M = 2^LOG2M used by the S4 convolution kernels (fftconv_kernels.hip).
//
// Both directions are "twiddle first" radix-2 butterflies  x0' = a + w b,  x1' = a - w b = 2 a - x0'  fused four stages at
// a time into radix-16 passes held in registers (a thread owns 16 points; a pass = one LDS round trip):
//   forward  natural order in -> BIT-REVERSED order out.  Stage t = 1..LOG2M has span M / 2^t and the twiddle
//            W_{2^t}^{rev_{t-1}(position >> (LOG2M - t + 1))}: it depends on the bits ABOVE the stage's bit, so the first
//            stages (where the zero padding of the convolution input is pruned) have unit twiddles;
//   inverse  bit-reversed in -> natural out, the textbook decimation-in-time order: spans 1, 2, 4 ..., twiddle
//            conj(W_{2 span}^{position mod span}): it depends on the bits BELOW the stage's bit.
// The spectrum between them is the DFT in bit-reversed order: no reordering pass.  (Rounds 1-3 ran the forward as
// decimation-in-frequency, twiddle AFTER the butterfly: (a + b, (a - b) w) costs 8 scalar / 4 packed instructions, the
// twiddle-first form 3 packed ones.)
//
// Arithmetic: complex values live in 64-bit register pairs and every butterfly is THREE v_pk_fma_f32
//      t  = a + w.x * b            (op_sel broadcasts w.x to both lanes)
//      x0 = t + w.y * (-b.y, b.x)  (op_sel swaps b's halves, neg_lo negates one)
//      x1 = 2 a - x0
// written as inline assembly: hipcc does not fold the half swaps and single-lane negations of complex arithmetic into the
// op_sel / neg modifiers of packed instructions (it emits v_mov / v_xor for them), and a packed fp32 instruction costs
// 4.3 cycles against 2.35-3.9 for a scalar one (tools/valu_rate.hip): 12.9 cycles per butterfly instead of ~21.
// Multiplications by -i / +i (W_4) ride in the same modifiers.
//
// Pass plan for E = LOG2M (even; odd sizes first take one radix-2 pass over the top bit):
//      index bits [E-4, E), [E-8, E-4), ... as radix-16 passes, and a final radix-4 pass over bits [0, 2) if E % 4 == 2
//      (E = 14: 4+4+4+2, E = 12: 4+4+4, E = 10: 4+4+2).
// A radix-16 pass over bits [b, b+4) (s = 2^b, a thread's points x[r] are base + r*s, r = r3 r2 r1 r0) with tau the pass's
// base twiddle, in units of W_16 (e >= 4: the twiddle of e - 4 times -i, folded into the operand modifiers):
//   forward: tau = W_M^{rev(position >> (b+4)) << b}; stages r3, r2, r1, r0 with twiddles tau^8, tau^4 W16^{4 r3},
//            tau^2 W16^{2 (r3 + 2 r2)}, tau W16^{r3 + 2 r2 + 4 r1};
//   inverse: tau = W_{16 s}^{position mod s}; stages r0, r1, r2, r3 with the conjugates of tau^8, tau^4 W16^{4 r0},
//            tau^2 W16^{2 (r & 3)}, tau W16^{r & 7}.
// The radix-4 pass over bits [0, 2) is the last two (forward) / first two (inverse) stages of the same 16-point
// transform on a thread's 16 CONTIGUOUS points.
//
// LDS rows are padded by one complex per 16 (pidx) so the strided and the 16-contiguous access patterns of every pass
// are bank-conflict free for ds_read_b64 / ds_write_b64.
//
// The file compiles for the host too (tests/native/fft_core_host.cpp emulates a workgroup thread by thread against numpy).
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

// A complex number in one 64-bit register pair (device) / a plain struct (host emulation).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float c2 __attribute__((ext_vector_type(2)));
DWS_HD c2 mk(float x, float y) { return c2{x, y}; }
#else
typedef float2 c2;
DWS_HD c2 mk(float x, float y) { return make_float2(x, y); }
#endif

// ---- packed complex primitives -------------------------------------------------------------------------------------
// Device: one or two VOP3P instructions each, operand halves routed by op_sel / op_sel_hi, signs by neg_lo / neg_hi.
// (asm without `volatile`: pure functions of their inputs, so the compiler may schedule, combine and drop them.)
#if defined(__HIP_DEVICE_COMPILE__)
#define DWS_PK3(OP, MODS, D, A, B, C) asm(OP " %0, %1, %2, %3 " MODS : "=v"(D) : "v"(A), "v"(B), "v"(C))
#define DWS_PK2(OP, MODS, D, A, B) asm(OP " %0, %1, %2 " MODS : "=v"(D) : "v"(A), "v"(B))
// first source in a scalar register pair (compile-time constant twiddles: no vector registers, no v_mov)
#define DWS_PK3S(OP, MODS, D, A, B, C) asm(OP " %0, %1, %2, %3 " MODS : "=v"(D) : "s"(A), "v"(B), "v"(C))
#define DWS_PK2S(OP, MODS, D, A, B) asm(OP " %0, %1, %2 " MODS : "=v"(D) : "s"(A), "v"(B))
DWS_HD c2 cadd(c2 a, c2 b) { c2 r; DWS_PK2("v_pk_add_f32", "", r, a, b); return r; }
DWS_HD c2 csub(c2 a, c2 b) { c2 r; DWS_PK2("v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]", r, a, b); return r; }
// a - i b = (a.x + b.y, a.y - b.x);  a + i b = (a.x - b.y, a.y + b.x)
DWS_HD c2 cadd_mi(c2 a, c2 b) { c2 r; DWS_PK2("v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]", r, a, b); return r; }
DWS_HD c2 cadd_pi(c2 a, c2 b) { c2 r; DWS_PK2("v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", r, a, b); return r; }
// a + W b with W = w (-i)^ROT (forward) or conj(w) i^ROT (inverse): two dependent v_pk_fma_f32
// WS: w is a compile-time constant, read from a scalar register pair
template <bool INV, int ROT, bool WS = false>
DWS_HD c2 cfma(c2 a, c2 w, c2 b) {
    c2 t, r;
#define DWS_CFMA(M1, M2)                                   \
    if constexpr (WS) {                                    \
        DWS_PK3S("v_pk_fma_f32", M1, t, w, b, a);          \
        DWS_PK3S("v_pk_fma_f32", M2, r, w, b, t);          \
    } else {                                               \
        DWS_PK3("v_pk_fma_f32", M1, t, w, b, a);           \
        DWS_PK3("v_pk_fma_f32", M2, r, w, b, t);           \
    }
    if (!INV && ROT == 0) {
        DWS_CFMA("op_sel_hi:[0,1,1]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]")
    } else if (!INV && ROT == 1) {     // w (-i b), -i b = (b.y, -b.x)
        DWS_CFMA("op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]")
    } else if (INV && ROT == 0) {      // conj(w) b
        DWS_CFMA("op_sel_hi:[0,1,1]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]")
    } else {                           // conj(w) (i b), i b = (-b.y, b.x)
        DWS_CFMA("op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]")
    }
#undef DWS_CFMA
    return r;
}
// 2 a - x0  (the second output of a butterfly whose first output is x0 = a + W b); the inline constant sits in the low
// half of its 64-bit operand, op_sel_hi = 0 hands it to both lanes
DWS_HD c2 cmirror(c2 a, c2 x0) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(x0));
    return r;
}
// w a and conj(w) a: v_pk_mul_f32 + v_pk_fma_f32
DWS_HD c2 cmul_(c2 a, c2 w) {
    c2 t, r;
    DWS_PK2("v_pk_mul_f32", "op_sel_hi:[0,1]", t, w, a);
    DWS_PK3("v_pk_fma_f32", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]", r, w, a, t);
    return r;
}
DWS_HD c2 cmulc(c2 a, c2 w) {  // a * conj(w)
    c2 t, r;
    DWS_PK2("v_pk_mul_f32", "op_sel_hi:[0,1]", t, w, a);
    DWS_PK3("v_pk_fma_f32", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]", r, w, a, t);
    return r;
}
// k a for a compile-time constant k (scalar register pair)
DWS_HD c2 cmulk(c2 a, c2 k) {
    c2 t, r;
    DWS_PK2S("v_pk_mul_f32", "op_sel_hi:[0,1]", t, k, a);
    DWS_PK3S("v_pk_fma_f32", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]", r, k, a, t);
    return r;
}
// s * a for a real scalar s held in the LOW half of `s2`
DWS_HD c2 cscale(c2 a, c2 s2) { c2 r; DWS_PK2("v_pk_mul_f32", "op_sel_hi:[0,1]", r, s2, a); return r; }
// a + conj(b), a - conj(b)
DWS_HD c2 cadd_conj(c2 a, c2 b) { c2 r; DWS_PK2("v_pk_add_f32", "neg_hi:[0,1]", r, a, b); return r; }
DWS_HD c2 csub_conj(c2 a, c2 b) { c2 r; DWS_PK2("v_pk_add_f32", "neg_lo:[0,1]", r, a, b); return r; }
// conj(2 a - x0)
DWS_HD c2 cmirror_conj(c2 a, c2 x0) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(x0));
    return r;
}
#undef DWS_PK3
#undef DWS_PK2
#undef DWS_PK3S
#undef DWS_PK2S
#else
DWS_HD c2 cadd(c2 a, c2 b) { return mk(a.x + b.x, a.y + b.y); }
DWS_HD c2 csub(c2 a, c2 b) { return mk(a.x - b.x, a.y - b.y); }
DWS_HD c2 cadd_mi(c2 a, c2 b) { return mk(a.x + b.y, a.y - b.x); }
DWS_HD c2 cadd_pi(c2 a, c2 b) { return mk(a.x - b.y, a.y + b.x); }
template <bool INV, int ROT, bool WS = false>
DWS_HD c2 cfma(c2 a, c2 w, c2 b) {
    // the same two fused multiply-adds per lane as the device form
    if (!INV && ROT == 0) return mk(fmaf(w.y, -b.y, fmaf(w.x, b.x, a.x)), fmaf(w.y, b.x, fmaf(w.x, b.y, a.y)));
    if (!INV && ROT == 1) return mk(fmaf(w.y, b.x, fmaf(w.x, b.y, a.x)), fmaf(w.y, b.y, fmaf(w.x, -b.x, a.y)));
    if (INV && ROT == 0) return mk(fmaf(w.y, b.y, fmaf(w.x, b.x, a.x)), fmaf(w.y, -b.x, fmaf(w.x, b.y, a.y)));
    return mk(fmaf(w.y, b.x, fmaf(w.x, -b.y, a.x)), fmaf(w.y, b.y, fmaf(w.x, b.x, a.y)));
}
DWS_HD c2 cmirror(c2 a, c2 x0) { return mk(fmaf(a.x, 2.f, -x0.x), fmaf(a.y, 2.f, -x0.y)); }
DWS_HD c2 cmul_(c2 a, c2 w) { return mk(fmaf(w.y, -a.y, w.x * a.x), fmaf(w.y, a.x, w.x * a.y)); }
DWS_HD c2 cmulc(c2 a, c2 w) { return mk(fmaf(w.y, a.y, w.x * a.x), fmaf(w.y, -a.x, w.x * a.y)); }
DWS_HD c2 cmulk(c2 a, c2 k) { return cmul_(a, k); }
DWS_HD c2 cscale(c2 a, c2 s2) { return mk(s2.x * a.x, s2.x * a.y); }
DWS_HD c2 cadd_conj(c2 a, c2 b) { return mk(a.x + b.x, a.y - b.y); }
DWS_HD c2 csub_conj(c2 a, c2 b) { return mk(a.x - b.x, a.y + b.y); }
DWS_HD c2 cmirror_conj(c2 a, c2 x0) { return mk(fmaf(a.x, 2.f, -x0.x), fmaf(-a.y, 2.f, x0.y)); }
#endif

DWS_HD c2 csqr(c2 a) { return cmul_(a, a); }
DWS_HD c2 cconj(c2 a) { return mk(a.x, -a.y); }
DWS_HD c2 mul_neg_i(c2 a) { return mk(a.y, -a.x); }  // a * (-i)
DWS_HD c2 mul_pos_i(c2 a) { return mk(-a.y, a.x); }  // a * (+i)
DWS_HD int pidx(int i) { return i + (i >> 4); }

DWS_HD int brev_bits(int k, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (int)(__brev((unsigned)k) >> (32 - bits)) : 0;
#else
    unsigned v = (unsigned)k, r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return (int)r;
#endif
}

// W_16^k = exp(-2 pi i k / 16), k = 0..3
template <int K>
DWS_HD c2 w16c() {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r = 0.70710678118654752440f;
    return K == 0 ? mk(1.f, 0.f) : K == 1 ? mk(c1, -s1) : K == 2 ? mk(r, -r) : mk(s1, -c1);
}

// Keeps a value opaque to the optimiser (device code): the twiddles derived from a pass's base twiddle are loop
// invariant over the rows a persistent workgroup walks, and hoisting them out of that loop costs ~20 VGPRs per pass for
// the kernel's whole lifetime (spills at 1024 threads); recomputing them per row is a few packed instructions per pass.
DWS_HD c2 opaque(c2 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// nothing moves across this point in the instruction schedule (device code)
DWS_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

DWS_HD int opaque(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// One radix-2 butterfly in place: a <- a + W b, b <- a - W b (HALF: only a is needed), W = w (-i)^ROT / conj(w) i^ROT.
// UNIT: w = 1 (two packed adds).
template <bool INV, int ROT, bool UNIT, bool HALF, bool WS = false>
DWS_HD void bfly(c2& a, c2& b, c2 w) {
    if (UNIT) {
        const c2 s = ROT ? (INV ? cadd_pi(a, b) : cadd_mi(a, b)) : cadd(a, b);
        if (!HALF) b = ROT ? (INV ? cadd_mi(a, b) : cadd_pi(a, b)) : csub(a, b);
        a = s;
    } else {
        const c2 x0 = cfma<INV, ROT, WS>(a, w, b);
        if (!HALF) b = cmirror(a, x0);
        a = x0;
    }
}

// The twiddles of a 16-point transform with base twiddle tau: T1[k] = tau W16^k (k < 4), T2[k] = tau^2 W8^k (k < 2),
// T4 = tau^4, T8 = tau^8.  TW = false: tau = 1 (compile-time constants; k = 0 entries are never multiplied with).
template <bool TW>
struct Tw16 {
    c2 T1[4], T2[2], T4, T8;
    DWS_HD explicit Tw16(c2 tau) {
        if (TW) {
            T1[0] = tau; T1[1] = cmulk(tau, w16c<1>()); T1[2] = cmulk(tau, w16c<2>()); T1[3] = cmulk(tau, w16c<3>());
            T2[0] = csqr(tau); T2[1] = cmulk(T2[0], w16c<2>());
            T4 = csqr(T2[0]); T8 = csqr(T4);
        } else {
            T1[0] = w16c<0>(); T1[1] = w16c<1>(); T1[2] = w16c<2>(); T1[3] = w16c<3>();
            T2[0] = w16c<0>(); T2[1] = w16c<2>();
            T4 = T8 = w16c<0>();
        }
    }
};

// One stage of the 16-point transform: butterflies (i, i + 2^BIT) over all i with that bit clear.  E(i) = the stage's
// twiddle exponent in units of W_16 (see the header); POW selects tau^POW's table.
template <bool INV, bool TW, int BIT, bool HALF, bool COPY>
DWS_HD void stage16(c2 (&x)[16], const Tw16<TW>& t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i & (1 << BIT)) continue;
        const int j = i + (1 << BIT);
        if (COPY) {            // forward, upper input zero (the convolution's padding): both outputs equal the lower input
            x[j] = x[i];
            continue;
        }
        // forward: the bits ABOVE this one, reversed; inverse: the bits BELOW it
        int e;
        if (!INV) {
            const int hi = i >> (BIT + 1), nb = 3 - BIT;              // nb bits above
            int rv = 0;
            for (int q = 0; q < nb; ++q) rv |= ((hi >> q) & 1) << (nb - 1 - q);
            e = rv << BIT;
        } else {
            e = (i & ((1 << BIT) - 1)) << (3 - BIT);
        }
        const int rot = e >> 2, sel = e & 3;
        // tau-power of this stage: forward stage BIT uses tau^(2^BIT), inverse stage BIT uses tau^(2^(3-BIT))
        const int pw = INV ? (8 >> BIT) : (1 << BIT);
        const bool unit = (!TW && sel == 0);
        const c2 w = pw == 8 ? t.T8 : pw == 4 ? t.T4 : pw == 2 ? t.T2[sel >> 1] : t.T1[sel];
        // (TW = false: the remaining twiddles are the constants W16^1..3, read from scalar registers)
        if (rot) {
            if (unit) bfly<INV, 1, true, HALF>(x[i], x[j], w);
            else bfly<INV, 1, false, HALF, !TW>(x[i], x[j], w);
        } else {
            if (unit) bfly<INV, 0, true, HALF>(x[i], x[j], w);
            else bfly<INV, 0, false, HALF, !TW>(x[i], x[j], w);
        }
    }
}

// The four fused stages of a radix-16 pass on registers: x[r] is the point base + r*s, tau the pass's base twiddle
// (TW = false: tau = 1).  TAIL: only the two stages over r1, r0 (the radix-4 pass over index bits [0, 2)).
// ZERO_HI (forward): x[8..15] are zero on entry.  LO_ONLY (inverse): only x[0..7] are needed on exit.
template <bool INV, bool TW, bool ZERO_HI = false, bool LO_ONLY = false, bool TAIL = false>
DWS_HD void fft16(c2 (&x)[16], c2 tau_in) {
    const Tw16<TW> t(TW ? opaque(tau_in) : tau_in);
    if (!INV) {
        if (!TAIL) {
            stage16<false, TW, 3, false, ZERO_HI>(x, t);
            stage16<false, TW, 2, false, false>(x, t);
        }
        stage16<false, TW, 1, false, false>(x, t);
        stage16<false, TW, 0, false, false>(x, t);
    } else {
        stage16<true, TW, 0, false, false>(x, t);
        stage16<true, TW, 1, false, false>(x, t);
        if (!TAIL) {
            stage16<true, TW, 2, false, false>(x, t);
            stage16<true, TW, 3, LO_ONLY, false>(x, t);
        }
    }
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
    // does the pass over bits [b, b+4) carry a non-unit base twiddle?  forward: unless no bit lies above it;
    // inverse: unless none lies below
    static constexpr bool tw_fwd(int b) { return b + 4 < LOG2M; }
    static constexpr bool tw_inv(int b) { return b != 0; }
};

// Base twiddles of the passes for the 16-point groups g = tid + i*THREADS (i < NG) this thread owns (tw[k] = W_M^k):
//   theta[p] (inverse)  W_{16 s}^{g mod s},  s = 2^b0(p): loaded once, kept in registers over the rows a workgroup walks;
//   phi(p)   (forward)  W_M^{rev(g >> b0(p)) << b0(p)}, and the forward radix-4 tail's W_M^{rev(g)} (the thread's 16
//            contiguous points are group g of a pass over bits [0, 4)): fetched where they are used (one 8-byte load per
//            pass, an L2 hit) -- kept live beside theta they cost six more registers and spill at M = 16384.
template <int LOG2M, int NG = 1>
struct FftTw {
    using P = FftPlan<LOG2M>;
    c2 theta[P::N16 > 0 ? P::N16 : 1][NG];
    DWS_HD void load(const c2* __restrict__ tw, int tid) {
        constexpr int THREADS = (P::M / 16) / NG;
#pragma unroll
        for (int p = 0; p < P::N16; ++p) {
            const int b = P::b0(p);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int g = tid + i * THREADS;
                theta[p][i] = (b == 0) ? mk(1.f, 0.f) : tw[(g & ((1 << b) - 1)) * (P::M >> (b + 4))];
            }
        }
    }
    // theta[p] of group g fetched at its use instead of held (a kernel at its register limit)
    template <int PASS>
    static DWS_HD c2 theta_at(const c2* __restrict__ tw, int g) {
        constexpr int b = P::b0(PASS);
        return (b == 0) ? mk(1.f, 0.f) : tw[(g & ((1 << b) - 1)) * (P::M >> (b + 4))];
    }
    template <int B0>
    static DWS_HD c2 phi(const c2* __restrict__ tw, int g) {
        return !P::tw_fwd(B0) ? mk(1.f, 0.f) : tw[brev_bits(g >> B0, LOG2M - 4 - B0) << B0];
    }
    static DWS_HD c2 tail_twiddle(const c2* __restrict__ tw, int g) { return tw[brev_bits(g, LOG2M - 4)]; }
};

// Point r of the 16-point group `g` of the pass over bits [B0, B0+4): padded LDS index.
template <int B0>
DWS_HD int group_base(int g) {
    return ((g >> B0) << (B0 + 4)) + (g & ((1 << B0) - 1));
}

// One radix-16 pass LDS -> LDS (group index g = tid when THREADS = M/16); tau = phi (forward) / theta (inverse).
template <int LOG2M, int B0, bool INV>
DWS_HD void pass16_lds(c2* __restrict__ X, c2 tau, int g) {
    constexpr int S = 1 << B0;
    constexpr bool TW = INV ? FftPlan<LOG2M>::tw_inv(B0) : FftPlan<LOG2M>::tw_fwd(B0);
    // pidx(base + r S) = pidx(base) + r S + (r S >> 4): the group's bits [B0, B0+4) of `base` are clear, so adding r S never
    // carries into or out of the low four bits -- ONE address register per pass, the points at immediate offsets
    const int pb = pidx(group_base<B0>(g));
    c2 x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = X[pb + r * S + ((r * S) >> 4)];
    fft16<INV, TW>(x, tau);
#pragma unroll
    for (int r = 0; r < 16; ++r) X[pb + r * S + ((r * S) >> 4)] = x[r];
}

// Final (forward; tau = psi) / first (inverse; unit twiddles) radix-4 pass over bits [0, 2): thread g owns points
// 16g .. 16g+15.
template <bool INV>
DWS_HD void pass4_lds(c2* __restrict__ X, c2 tau, int g) {
    c2 x[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) x[d] = X[17 * g + d];  // pidx(16 g + d) = 17 g + d
    if (INV) fft16<true, false, false, false, true>(x, tau);
    else fft16<false, true, false, false, true>(x, tau);
#pragma unroll
    for (int d = 0; d < 16; ++d) X[17 * g + d] = x[d];
}

// Radix-2 pass over the top bit (odd LOG2M only): `t` runs over M/2 butterflies.  Forward: the first stage, unit
// twiddle; inverse: the last stage, conj(W_M^t).
template <int LOG2M, bool INV>
DWS_HD void pass2_top(c2* __restrict__ X, const c2* __restrict__ tw, int t) {
    constexpr int h = (1 << LOG2M) / 2;
    c2 u = X[pidx(t)], v = X[pidx(t + h)];
    if (!INV) bfly<false, 0, true, false>(u, v, u);
    else bfly<true, 0, false, false>(u, v, tw[t]);
    X[pidx(t)] = u;
    X[pidx(t + h)] = v;
}

// Pointwise stage of the real-input convolution in bit-reversed order for pair q (0 < q < M/2): positions p = 2q
// (k = brev(p) < M/2) and pm = position of M - k.  With N = 2M, Wk = exp(-2 pi i k / N):
//   Xe = (Zk + conj Zm)/2, Xo = -(i/2)(Zk - conj Zm), t = Wk Xo;  A[k] = Xe + t, A[M-k] = conj(Xe - t);  Y = A * Kf
//   Ye = (Yk + conj Ym)/2, Yo = (Yk - conj Ym)/2 * conj(Wk);  Zy[k] = Ye + i Yo, Zy[M-k] = conj(Ye - i Yo)
// csign = -1 multiplies by conj(K_f): the adjoint (correlation) of the convolution.
DWS_HD void pointwise_pair(c2& zk_io, c2& zm_io, c2 wk, c2 ka, c2 kb, float csign) {
    const c2 zk = zk_io, zm = zm_io;
    const c2 xe = mk(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const c2 d = mk(zk.x - zm.x, zk.y + zm.y);           // Zk - conj Zm
    const c2 xo = mk(0.5f * d.y, -0.5f * d.x);           // -(i/2) d
    const c2 t = cmul_(xo, wk);
    const c2 ak = cadd(xe, t), am = cconj(csub(xe, t));
    const c2 yk = cmul_(ak, mk(ka.x, csign * ka.y)), ym = cmul_(am, mk(kb.x, csign * kb.y));
    const c2 ye = mk(0.5f * (yk.x + ym.x), 0.5f * (yk.y - ym.y));
    const c2 e = mk(0.5f * (yk.x - ym.x), 0.5f * (yk.y + ym.y));  // (Yk - conj Ym)/2
    const c2 iyo = mul_pos_i(cmulc(e, wk));
    zk_io = cadd(ye, iyo);
    zm_io = cconj(csub(ye, iyo));
}

// The same pair on packed arithmetic, result scaled by FOUR (the two halvings of pointwise_pair are exact and commute with
// every rounding: they move into the caller's 1/M): 14 packed instructions + 2 multiplies against ~45 scalar ones.
//   s = Zk + conj Zm, d = Zk - conj Zm;  2 A[k] = s + Wk (-i d),  2 conj A[M-k] = u = 2 s - 2 A[k]
//   Yk = 2 A[k] K1,  v = u conj(K2) = 2 conj(Y[M-k]);  ye = Yk + v, e = Yk - v
//   4 Zy[k] = ye + conj(Wk) (i e),  4 Zy[M-k] = conj(2 ye - 4 Zy[k])
DWS_HD void pointwise_pair4(c2& zk_io, c2& zm_io, c2 wk, c2 ka, c2 kb, float csign) {
    const c2 s = cadd_conj(zk_io, zm_io), d = csub_conj(zk_io, zm_io);
    const c2 ak = cfma<false, 1>(s, wk, d);
    const c2 u = cmirror(s, ak);
    const c2 yk = cmul_(ak, mk(ka.x, csign * ka.y)), v = cmulc(u, mk(kb.x, csign * kb.y));
    const c2 ye = cadd(yk, v), e = csub(yk, v);
    const c2 zk = cfma<true, 1>(ye, wk, e);
    zk_io = zk;
    zm_io = cmirror_conj(ye, zk);
}
DWS_HD void pointwise_self4(c2& z0, c2& z1, c2 kf0, c2 kfM, c2 kfh, float csign) {
    const float y0 = (z0.x + z0.y) * kf0.x, ym = (z0.x - z0.y) * kfM.x;
    z0 = mk(2.f * (y0 + ym), 2.f * (y0 - ym));
    const c2 t = cmulc(z1, mk(kfh.x, csign * kfh.y));
    z1 = mk(4.f * t.x, 4.f * t.y);
}

// The two halves of pointwise_pair on their own (kernels that accumulate several spectra between them):
// bins A[k], A[M-k] of the real row from the packed spectrum, and the packed form of a real-row spectrum Y[k], Y[M-k].
DWS_HD void pair_bins(c2 zk, c2 zm, c2 wk, c2& ak, c2& am) {
    const c2 xe = mk(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const c2 d = mk(zk.x - zm.x, zk.y + zm.y);
    const c2 t = cmul_(mk(0.5f * d.y, -0.5f * d.x), wk);
    ak = cadd(xe, t);
    am = cconj(csub(xe, t));
}
DWS_HD void pair_repack(c2 yk, c2 ym, c2 wk, c2& zk, c2& zm) {
    const c2 ye = mk(0.5f * (yk.x + ym.x), 0.5f * (yk.y - ym.y));
    const c2 e = mk(0.5f * (yk.x - ym.x), 0.5f * (yk.y + ym.y));
    const c2 iyo = mul_pos_i(cmulc(e, wk));
    zk = cadd(ye, iyo);
    zm = cconj(csub(ye, iyo));
}

// q = 0: k = 0 (self-paired, carries DC and Nyquist, both real) and k = M/2 (position 1, self-paired).
DWS_HD void pointwise_self(c2& z0, c2& z1, c2 kf0, c2 kfM, c2 kfh, float csign) {
    const float y0 = (z0.x + z0.y) * kf0.x;   // A[0] = Re + Im; irfft ignores Im of DC / Nyquist
    const float ym = (z0.x - z0.y) * kfM.x;   // A[M] = Re - Im
    z0 = mk(0.5f * (y0 + ym), 0.5f * (y0 - ym));
    z1 = cmulc(z1, mk(kfh.x, csign * kfh.y));  // Zy[M/2] = Z[M/2] * conj(Kf[M/2])
}


// ---- fused tail: forward radix-4 tail + pointwise stage + inverse radix-4 tail in ONE LDS round trip ---------------
// (plans with TAIL4: the separate sequence reads and writes every point three times -- tail pass, pair stage, tail pass --
// with two barriers between them.)
// The pair stage couples frequency k with M - k.  In the bit-reversed layout a block of 16 contiguous positions 16 t ..
// 16 t + 15 holds the frequencies kb + (r0 M/2 + r1 M/4 + r2 M/8 + r3 M/16), kb = rev(t) < M/16, and their partners are
//   M - k = (M/16 - kb) + ((1 - r0) M/2 + (1 - r1) M/4 + (1 - r2) M/8 + (1 - r3) M/16)          (kb != 0)
// i.e. position 15 - r of the block whose base frequency is M/16 - kb.  Negating a frequency complements, in the
// reversed index, every bit below the highest set one: that block is t' = 3 msb(t) - 1 - t (t and t' mirror each other
// inside the octave [msb, 2 msb); t = 1 is its own mirror).  So thread t takes the r3 = 0 half of block t and the r3 = 1
// half of block t' -- 16 points, x[0..7] and x[8..15], the same index bits as a whole block -- and holds all eight pairs
// (x[e], x[15 - e]), e even, with the tables' pair index q = position / 2 = 8 t + e/2 (e < 8), 8 t' + e/2 (e >= 8); thread t'
// takes the other two halves.  Block 0 (kb = 0) pairs inside itself in a different pattern (k = 0 and M/2 are their own
// partners, group M/8 mirrors itself, groups M/16 and 3M/16 each other): thread 0 takes the whole block and routes its
// odd positions into the generic slots for the pair stage.
// The radix-4 butterflies only couple positions inside a group of four, so the two halves transform independently, each
// with its block's tail twiddle; the arithmetic per point is exactly that of pass4_lds / pointwise_pair / pass4_lds.
DWS_HD int mirror_block(int t) {
    if (t == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    const int msb = 1 << (31 - __clz(t));
#else
    int msb = 1;
    while (2 * msb <= t) msb *= 2;
#endif
    return 3 * msb - 1 - t;
}

// forward tail stage BIT (1, then 0) on x[0..7] (twiddles of tl) and x[8..15] (twiddles of th): stage16<false, true, BIT>
// with the table chosen by the half
template <int BIT>
DWS_HD void stage16_fwd_halves(c2 (&x)[16], const Tw16<true>& tl, const Tw16<true>& th) {
    static_assert(BIT == 0 || BIT == 1, "tail stages");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i & (1 << BIT)) continue;
        const int j = i + (1 << BIT);
        const int hi = i >> (BIT + 1), nb = 3 - BIT;
        int rv = 0;
        for (int q = 0; q < nb; ++q) rv |= ((hi >> q) & 1) << (nb - 1 - q);
        const int e = rv << BIT, rot = e >> 2, sel = e & 3;
        const Tw16<true>& t = (i & 8) ? th : tl;
        const c2 w = BIT == 1 ? t.T2[sel >> 1] : t.T1[sel];
        if (rot) bfly<false, 1, false, false>(x[i], x[j], w);
        else bfly<false, 0, false, false>(x[i], x[j], w);
    }
}

// the same stage on ONE half (HALF = 0: x[0..7], 1: x[8..15]) with that half's table alone: a caller short of registers builds the
// two Tw16 tables one after the other instead of holding both (fftcorr_blk_kernel)
template <int BIT, int HALF>
DWS_HD void stage16_fwd_half(c2 (&x)[16], const Tw16<true>& t) {
    static_assert((BIT == 0 || BIT == 1) && (HALF == 0 || HALF == 1), "tail stages");
#pragma unroll
    for (int i = 8 * HALF; i < 8 * HALF + 8; ++i) {
        if (i & (1 << BIT)) continue;
        const int j = i + (1 << BIT);
        const int hi = i >> (BIT + 1), nb = 3 - BIT;
        int rv = 0;
        for (int q = 0; q < nb; ++q) rv |= ((hi >> q) & 1) << (nb - 1 - q);
        const int e = rv << BIT, rot = e >> 2, sel = e & 3;
        const c2 w = BIT == 1 ? t.T2[sel >> 1] : t.T1[sel];
        if (rot) bfly<false, 1, false, false>(x[i], x[j], w);
        else bfly<false, 0, false, false>(x[i], x[j], w);
    }
}

// What the pass reads from global memory before it can start: the first half's tables and the two tail twiddles.  A kernel
// may request them ahead of the barrier in front of the pass (TailPre::load) and hand them in.
struct TailPre {
    c2 wk[4], ka[4], kb[4], taul, tauh;
    template <int LOG2M, bool TAILS>
    DWS_HD void load(const c2* __restrict__ tw, const c2* __restrict__ twp, const c2* __restrict__ kfa,
                     const c2* __restrict__ kfb, int t) {
        const int tm = mirror_block(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wk[i] = twp[8 * t + i];
            ka[i] = kfa[8 * t + i];
            kb[i] = kfb[8 * t + i];
        }
        taul = tauh = mk(1.f, 0.f);
        if (TAILS) {
            taul = tw[brev_bits(t, LOG2M - 4)];
            tauh = tw[brev_bits(tm, LOG2M - 4)];
        }
    }
};

// X4: the packed pair arithmetic, the row comes out scaled by four (pointwise_pair4).
// TAILS = false: the pair stage alone in the same block / mirror-block order (plans without a radix-4 tail): contiguous
// conflict-free LDS runs and adjacent table entries instead of eight scattered pairs per thread.
template <int LOG2M, bool X4 = false, bool TAILS = true>
DWS_HD void pass_tail_pointwise(c2* __restrict__ X, const c2* __restrict__ tw, const c2* __restrict__ twp,
                                const c2* __restrict__ kfa, const c2* __restrict__ kfb, const c2* __restrict__ kfs, int t,
                                float csign, const TailPre* pre = nullptr) {
    static_assert(FftPlan<LOG2M>::TAIL4 || !TAILS, "plans that end in a radix-4 pass");
    const int tm = mirror_block(t);
    // the tables of the four pairs whose even position lies in block t first: their L2 round trip runs under the LDS reads
    // and the tail butterflies; the other four (block t') are requested once the butterflies are done and arrive under
    // the first four pairs' arithmetic (all eight at once cost 24 more live registers: spills at 1024 threads)
    TailPre own;
    if (!pre) own.template load<LOG2M, TAILS>(tw, twp, kfa, kfb, t);
    const TailPre& in = pre ? *pre : own;
    c2 wk[4], ka[4], kb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wk[i] = in.wk[i];
        ka[i] = in.ka[i];
        kb[i] = in.kb[i];
    }
    const c2 taul = in.taul, tauh = in.tauh;
    c2 x[16];
#pragma unroll
    for (int d = 0; d < 8; ++d) x[d] = X[17 * t + d];            // pidx(16 t + d) = 17 t + d
#pragma unroll
    for (int d = 8; d < 16; ++d) x[d] = X[17 * tm + d];
    if (TAILS) {
        const Tw16<true> tl(opaque(taul)), th(opaque(tauh));
        stage16_fwd_halves<1>(x, tl, th);
        stage16_fwd_halves<0>(x, tl, th);
        sched_fence();
    }
    c2 wk2[4], ka2[4], kb2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wk2[i] = twp[8 * tm + 4 + i];
        ka2[i] = kfa[8 * tm + 4 + i];
        kb2[i] = kfb[8 * tm + 4 + i];
    }
    if (t == 0) {   // block 0: partners of 2, 4, 6, 8, 10, 12, 14 are 3, 7, 5, 15, 13, 11, 9; generic slot of e is 15 - e
        const c2 x1 = x[1], x3 = x[3], x5 = x[5], x7 = x[7], x9 = x[9], x11 = x[11], x13 = x[13], x15 = x[15];
        x[15] = x1; x[13] = x3; x[11] = x7; x[9] = x5; x[7] = x15; x[5] = x13; x[3] = x11; x[1] = x9;
    }
#pragma unroll
    for (int e = 2; e < 8; e += 2) {
        if (X4) pointwise_pair4(x[e], x[15 - e], wk[e / 2], ka[e / 2], kb[e / 2], csign);
        else pointwise_pair(x[e], x[15 - e], wk[e / 2], ka[e / 2], kb[e / 2], csign);
    }
    {
        c2 a = x[0], b = x[15];
        if (X4) pointwise_pair4(a, b, wk[0], ka[0], kb[0], csign);
        else pointwise_pair(a, b, wk[0], ka[0], kb[0], csign);
        if (t == 0) {   // k = 0 and k = M/2 are their own partners
            if (X4) pointwise_self4(x[0], x[15], kfs[0], kfs[1], kfs[2], csign);
            else pointwise_self(x[0], x[15], kfs[0], kfs[1], kfs[2], csign);
        } else {
            x[0] = a;
            x[15] = b;
        }
    }
    sched_fence();
#pragma unroll
    for (int e = 8; e < 16; e += 2) {
        if (X4) pointwise_pair4(x[e], x[15 - e], wk2[e / 2 - 4], ka2[e / 2 - 4], kb2[e / 2 - 4], csign);
        else pointwise_pair(x[e], x[15 - e], wk2[e / 2 - 4], ka2[e / 2 - 4], kb2[e / 2 - 4], csign);
    }
    if (t == 0) {
        const c2 n1 = x[1], n3 = x[3], n5 = x[5], n7 = x[7], n9 = x[9], n11 = x[11], n13 = x[13], n15 = x[15];
        x[1] = n15; x[3] = n13; x[7] = n11; x[5] = n9; x[15] = n7; x[13] = n5; x[11] = n3; x[9] = n1;
    }
    if (TAILS) fft16<true, false, false, false, true>(x, mk(1.f, 0.f));     // inverse tail: unit base twiddle in every block
#pragma unroll
    for (int d = 0; d < 8; ++d) X[17 * t + d] = x[d];
#pragma unroll
    for (int d = 8; d < 16; ++d) X[17 * tm + d] = x[d];
}

}  // namespace dws
