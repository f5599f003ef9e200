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
// reinterpreted as complex pairs).  The in-place forward (fft_core.h: twiddle-first radix-2 butterflies
// on packed fp32, fused four at a time into radix-16 register passes) leaves the spectrum in
// BIT-REVERSED order; the pointwise stage works in that order and a decimation-in-time inverse
// brings natural order back -- no reordering pass at all.
// LDS rows are padded by one complex per 16 so both the strided and the 16-contiguous access
// patterns are bank-conflict free for ds_read_b64 / ds_write_b64.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "fftconv.h"

#include "fft_core.h"

namespace dws {

__device__ __forceinline__ float gelu_f(float x) { return dws_gelu(x); }
__device__ __forceinline__ int brev(int k, int bits) { return brev_bits(k, bits); }

// (Round 3, measured and not kept: every pass over bits [B0, B0+4) with B0 <= 6, and the radix-4 tail, is wave-local -- the 64
// groups of a wave touch exactly the points [1024 w, 1024 w + 1024) -- so the workgroup barriers between those passes can be
// replaced by a wavefront-scope fence (4 of the 9 barriers per row at M = 16384).  All parity tests stay green and the time
// does not move: 86.4 vs 86.7 us.  The row is bound by its 2660 VALU instructions per wave (43.6 M per launch, ~56 % of the
// cycles at 2.7 cycles each) plus the LDS write path (~35 %), not by waves waiting for each other.)
// Phase stamps (DWS_FFT_TRACE=1: `fftconv_kernel<.., TRACE = true>`): the pass drivers below call st() after each
// barrier-closed phase; the product instances get NoStamp, which compiles to nothing.
struct NoStamp {
    __device__ __forceinline__ void operator()() const {}
    __device__ __forceinline__ void loads_landed() const {}
};
constexpr int FFT_TRACE_SLOTS = 16, FFT_TRACE_ROWS = 4;
struct WaveStamp {
    unsigned long long* slot;   // this wave's [FFT_TRACE_ROWS][FFT_TRACE_SLOTS]; null beyond the traced rows
    int k;
    bool lane0;
    __device__ __forceinline__ void operator()() {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (slot && lane0 && k < FFT_TRACE_SLOTS) slot[k] = t;
        ++k;
    }
    __device__ __forceinline__ void loads_landed() {   // the top pass's global loads: wait for them, then stamp
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (*this)();
    }
};

// Forward passes P0 .. of the plan on an LDS-resident row (a barrier after each).  A thread owns the NG groups
// g = tid + i*THREADS, THREADS = M/16/NG.
template <int LOG2M, int NG, int P0, class ST = NoStamp>
__device__ __forceinline__ void fft_forward_from(c2* X, const c2* __restrict__ tw, const FftTw<LOG2M, NG>& W, int tid,
                                                 ST&& st = ST()) {
    using P = FftPlan<LOG2M>;
    constexpr int THREADS = (P::M / 16) / NG;
    if constexpr (P0 < P::N16) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (i) __builtin_amdgcn_sched_barrier(0);   // one group's 16 points in registers at a time
            pass16_lds<LOG2M, P::b0(P0), false>(X, FftTw<LOG2M, NG>::template phi<P::b0(P0)>(tw, opaque(tid) + i * THREADS),
                                                tid + i * THREADS);
        }
        __syncthreads();
        st();
        fft_forward_from<LOG2M, NG, P0 + 1>(X, tw, W, tid, st);
    } else if constexpr (P::TAIL4) {
#pragma unroll
        for (int i = 0; i < NG; ++i)
            pass4_lds<false>(X, FftTw<LOG2M, NG>::tail_twiddle(tw, opaque(tid) + i * THREADS), tid + i * THREADS);
        __syncthreads();
        st();
    }
}

// Inverse passes in mirrored order down to (and including) radix-16 pass P0.
template <int LOG2M, int NG, int P0, int PCUR, class ST = NoStamp>
__device__ __forceinline__ void fft_inverse_passes(c2* X, const FftTw<LOG2M, NG>& W, int tid, ST&& st = ST()) {
    constexpr int THREADS = (FftPlan<LOG2M>::M / 16) / NG;
    if constexpr (PCUR > P0) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (i) __builtin_amdgcn_sched_barrier(0);
            pass16_lds<LOG2M, FftPlan<LOG2M>::b0(PCUR - 1), true>(X, W.theta[PCUR - 1][i], tid + i * THREADS);
        }
        __syncthreads();
        st();
        fft_inverse_passes<LOG2M, NG, P0, PCUR - 1>(X, W, tid, st);
    }
}

template <int LOG2M, int NG, int P0, class ST = NoStamp>
__device__ __forceinline__ void fft_inverse_to(c2* X, const FftTw<LOG2M, NG>& W, int tid, ST&& st = ST()) {
    using P = FftPlan<LOG2M>;
    constexpr int THREADS = (P::M / 16) / NG;
    if constexpr (P::TAIL4) {
#pragma unroll
        for (int i = 0; i < NG; ++i) pass4_lds<true>(X, mk(1.f, 0.f), tid + i * THREADS);
        __syncthreads();
        st();
    }
    fft_inverse_passes<LOG2M, NG, P0, P::N16>(X, W, tid, st);
}

// The same passes for ONE group per thread with the base twiddles requested a pass ahead: `cur` is the twiddle of pass P0 (of
// the radix-4 tail when P0 == N16), already on its way; a pass requests the next one's before it starts its own LDS reads,
// so the L2 round trip runs under a whole pass instead of in front of its butterflies (the phase trace: a forward pass
// that fetches its twiddle itself takes 3.5 k cycles at M = 4096 against 1.7 k for an inverse pass that has it in a register).
template <int LOG2M, int P0, bool TAIL>
__device__ __forceinline__ c2 fwd_twiddle(const c2* __restrict__ tw, int tid) {
    using P = FftPlan<LOG2M>;
    if constexpr (P0 < P::N16) return FftTw<LOG2M, 1>::template phi<P::b0(P0)>(tw, tid);
    else if constexpr (P::TAIL4 && TAIL) return FftTw<LOG2M, 1>::tail_twiddle(tw, tid);
    else return mk(1.f, 0.f);
}
// `before_last_barrier()` runs after the last pass's LDS writes and before its barrier: the caller's requests for the stage
// that follows (the pair stage's tables), issued while the pass's registers are free and the waves wait for each other.
template <int LOG2M, int P0, bool TAIL, class ST, class F>
__device__ __forceinline__ void fft_forward_chain(c2* X, const c2* __restrict__ tw, int tid, ST&& st, c2 cur,
                                                  F&& before_last_barrier) {
    using P = FftPlan<LOG2M>;
    constexpr bool LAST = TAIL && P::TAIL4 ? (P0 == P::N16) : (P0 + 1 == P::N16);
    if constexpr (P0 < P::N16) {
        const c2 nxt = fwd_twiddle<LOG2M, P0 + 1, TAIL>(tw, opaque(tid));
        __builtin_amdgcn_sched_barrier(0);
        pass16_lds<LOG2M, P::b0(P0), false>(X, cur, tid);
        if constexpr (LAST) before_last_barrier();
        __syncthreads();
        st();
        fft_forward_chain<LOG2M, P0 + 1, TAIL>(X, tw, tid, st, nxt, before_last_barrier);
    } else if constexpr (P::TAIL4 && TAIL) {
        pass4_lds<false>(X, cur, tid);
        before_last_barrier();
        __syncthreads();
        st();
    }
}
// Inverse radix-16 passes PCUR-1 .. P0 with fetched twiddles, chained the same way; returns the twiddle of pass P0-1 (the
// caller's bottom pass), requested under pass P0.
template <int LOG2M, int P0, int PCUR, class ST>
__device__ __forceinline__ c2 fft_inverse_chain(c2* X, const c2* __restrict__ tw, int tid, ST&& st, c2 cur) {
    if constexpr (PCUR > P0) {
        c2 nxt = mk(1.f, 0.f);
        if constexpr (PCUR >= 2) nxt = FftTw<LOG2M, 1>::template theta_at<PCUR - 2>(tw, opaque(tid));
        __builtin_amdgcn_sched_barrier(0);
        pass16_lds<LOG2M, FftPlan<LOG2M>::b0(PCUR - 1), true>(X, cur, tid);
        __syncthreads();
        st();
        return fft_inverse_chain<LOG2M, P0, PCUR - 1>(X, tw, tid, st, nxt);
    } else {
        return cur;
    }
}

// Whole transforms of an LDS-resident row; the caller has synchronised after filling X.
template <int LOG2M, int NG>
__device__ __forceinline__ void fft_forward(c2* X, const c2* tw, const FftTw<LOG2M, NG>& W, int tid) {
    constexpr int THREADS = (1 << LOG2M) / 16 / NG;
    if constexpr (FftPlan<LOG2M>::ODD) {
        for (int t = tid; t < (1 << LOG2M) / 2; t += THREADS) pass2_top<LOG2M, false>(X, tw, t);
        __syncthreads();
    }
    fft_forward_from<LOG2M, NG, 0>(X, tw, W, tid);
}

template <int LOG2M, int NG>
__device__ __forceinline__ void fft_inverse(c2* X, const c2* tw, const FftTw<LOG2M, NG>& W, int tid) {
    constexpr int THREADS = (1 << LOG2M) / 16 / NG;
    fft_inverse_to<LOG2M, NG, 0>(X, W, tid);
    if constexpr (FftPlan<LOG2M>::ODD) {
        for (int t = tid; t < (1 << LOG2M) / 2; t += THREADS) pass2_top<LOG2M, true>(X, tw, t);
        __syncthreads();
    }
}

// Forward transform of a zero-padded row straight from global memory: `src` holds n_valid (<= M/2) packed complex points,
// the rest of the M-point row is zero.  Even sizes: the top radix-16 pass runs on registers (a thread's 16 points are
// g + (M/16) r, of which r >= 8 are padding and never loaded) and only its result goes to LDS; odd sizes stage the row
// in LDS first.  The caller must have passed a barrier since the last read of X.
template <int LOG2M, int NG, class ST = NoStamp>
__device__ __forceinline__ void fft_forward_global(c2* X, const c2* __restrict__ src, int n_valid, const c2* tw,
                                                   const FftTw<LOG2M, NG>& W, int tid, ST&& st = ST()) {
    constexpr int M = 1 << LOG2M, G = M / 16, THREADS = G / NG;
    if constexpr (!FftPlan<LOG2M>::ODD) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (i) __builtin_amdgcn_sched_barrier(0);
            const int g = tid + i * THREADS;
            c2 x[16];
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = (g + G * r < n_valid) ? src[g + G * r] : mk(0.f, 0.f);
#pragma unroll
            for (int r = 8; r < 16; ++r) x[r] = mk(0.f, 0.f);
            if (i == 0) st.loads_landed();
            fft16<false, false, true>(x, mk(1.f, 0.f));   // no bit above the top pass: its twiddles are the constants W16^k
#pragma unroll
            for (int r = 0; r < 16; ++r) X[pidx(g) + (G + G / 16) * r] = x[r];   // = pidx(g + G r): G is a multiple of 16
        }
        __syncthreads();
        st();
        fft_forward_from<LOG2M, NG, 1>(X, tw, W, tid, st);
    } else {
        for (int j = tid; j < M; j += THREADS) X[pidx(j)] = (j < n_valid) ? src[j] : mk(0.f, 0.f);
        __syncthreads();
        fft_forward<LOG2M, NG>(X, tw, W, tid);
    }
}

// Row schedule: a persistent grid whose block b sits on XCD b % 8 (observed dispatch order); every XCD gets a contiguous
// range of rows, i.e. whole channels -- the rows of one channel share K_f, which then lives in ONE L2.
struct RowSchedule {
    int first, end, step;
    __device__ __forceinline__ RowSchedule(int rows) {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, per_xcd = (rows + 7) >> 3;
        first = xcd * per_xcd + (blockIdx.x >> 3);
        end = min(rows, (xcd + 1) * per_xcd);
        step = nwg >> 3;    // the launcher makes nwg a multiple of 8
    }
};

// g[b,h,:] = GELU(conv(u, K_h)[:L] + D[h] * u)
//
// Even sizes: the top radix-16 pass works straight from / to global memory -- a thread's 16 points are tid + THREADS*r,
// of which r >= 8 are the zero padding (never loaded) on the way in and never needed on the way out.
// FUSED (plans that end in a radix-4 tail, one 16-point group per thread): forward tail + pair stage + inverse tail are ONE
// pass (fft_core.h: pass_tail_pointwise) -- a row of M = 16384 goes through LDS in six round trips and six barriers
// instead of eight and nine.
// RSUM (training, the adjoint pass): the row's sum over its L outputs leaves with it (FftConvArgs::rowsum) -- the values are in
// registers in the bottom pass, a separate rowsum_bc launch read the whole tensor again (1.5 ms of a C5 step).
template <int LOG2M, int THREADS, bool TRACE = false, bool FUSED = false, bool RSUM = false>
__global__ __launch_bounds__(THREADS) void fftconv_kernel(FftConvArgs a) {
    using P = FftPlan<LOG2M>;
    static_assert(!TRACE || !P::ODD, "phase stamps: even sizes only");
    static_assert(!FUSED || (P::TAIL4 && !P::ODD && THREADS == (1 << LOG2M) / 16), "fused tail: shape");
    constexpr int M = 1 << LOG2M;
    constexpr bool DIRECT = !P::ODD;
    extern __shared__ __attribute__((aligned(16))) c2 X[];  // M + M/16 complex
    static_assert(!RSUM || !FftPlan<LOG2M>::ODD, "row sums: even plans (the bottom pass holds the outputs in registers)");
    __shared__ float rsum_w[RSUM ? (THREADS + 63) / 64 : 1];
    const int tid = threadIdx.x;
    const int L = a.L, Lc = L / 2;  // L even
    constexpr int NG = (1 << LOG2M) / 16 / THREADS;
    // (Distinct s_setprio per wave of a SIMD, so that one wave is writing its pass back while the others still compute:
    // 86.6 vs 86.4 us, no effect.)
    FftTw<LOG2M, NG> W;
    if constexpr (!FUSED) W.load(a.tw, tid);   // (FUSED fetches every base twiddle at its use: no register held over a row)
    // (the packed pair arithmetic leaves the row scaled by four, see pointwise_pair4)
    const float scale = 0.25f / (float)M, csign = a.conj_k ? -1.f : 1.f;
    const RowSchedule sch(a.B * a.H);
    // (Fetching the next row's points before the last pass of the current one was measured SLOWER: 99 vs 88 us at
    // M = 16384 -- 16 more live registers at the 128-VGPR limit of a 1024-thread workgroup.  Round 4: requested AFTER the
    // bottom pass, when its registers are free, and BEFORE the closing barrier, so that the waves that reach the barrier early
    // -- the trace shows 2-3 k cycles of mean wait -- have the round trip covered: the trace's "loads landed" phase drops
    // from 3.5 k to 0.6 k cycles, but __syncthreads() waits for vmcnt(0), i.e. for the requests, and the row gets no
    // shorter (75.7 vs 74.4 us); with `s_waitcnt lgkmcnt(0); s_barrier` in its place (no spill when the choice is made at
    // compile time) the loads do land under the barrier, and the bottom pass and the barrier grow by more than the top pass
    // saves: 74.9 vs 73.6 us (`profiles/r04_fft_trace_c3_early_loads_dropped.txt`).  Dropped: a row's points are requested
    // at its own top pass.)
    // TRACE: stamps of the first FFT_TRACE_ROWS rows this workgroup walks: row start | top pass: loads landed, done | forward
    // passes | pointwise | inverse passes | bottom pass + stores | closing barrier
    typename std::conditional<TRACE, WaveStamp, NoStamp>::type st;
    int ri = 0;
#pragma unroll 1
    for (int row = sch.first; row < sch.end; row += sch.step) {
        // rows of one channel h are adjacent (the kernel spectrum is shared by the batch)
        const int h = row / a.B, b = row % a.B;
        const size_t off = ((size_t)b * a.H + h) * L;
        const c2* __restrict__ u2 = reinterpret_cast<const c2*>(a.u + off);
        if constexpr (TRACE) {
            st.slot = ri < FFT_TRACE_ROWS ? a.trace + (((size_t)blockIdx.x * (THREADS / 64) + __builtin_amdgcn_readfirstlane(tid >> 6)) * FFT_TRACE_ROWS + ri) *
                                                          FFT_TRACE_SLOTS
                                          : nullptr;
            st.k = 0;
            st.lane0 = (tid & 63) == 0;
            ++ri;
            st();
        }
        c2 th_inv = mk(1.f, 0.f);   // FUSED: the first inverse pass's twiddle, requested under the pair pass's barrier
        if constexpr (DIRECT) {
            static_assert(NG == 1, "one block of 16 positions per thread");
            // top pass on registers (fft_forward_global's); buffer loads: a point beyond the row's L/2 comes back as zero
            __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)(a.u + off), 0, L * 4, 0x00020000);
            c2 x[16];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                x[r] = __builtin_bit_cast(c2, __builtin_amdgcn_raw_buffer_load_b64(rU, (tid + (M / 16) * r) * 8, 0, 0));
            const c2 ph1 = fwd_twiddle<LOG2M, 1, !FUSED>(a.tw, opaque(tid));   // the next pass's base twiddle, a pass ahead
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 8; r < 16; ++r) x[r] = mk(0.f, 0.f);
            st.loads_landed();
            fft16<false, false, true>(x, mk(1.f, 0.f));
#pragma unroll
            for (int r = 0; r < 16; ++r) X[pidx(tid) + (M / 16 + M / 256) * r] = x[r];
            __syncthreads();
            st();
            // the pair stage's first tables (and the fused tails' twiddles) are requested under the last forward pass's barrier
            TailPre pre;
            fft_forward_chain<LOG2M, 1, !FUSED>(X, a.tw, tid, st, ph1, [&] {
                pre.template load<LOG2M, FUSED>(a.tw, a.twp, a.kfa + (size_t)h * (M / 2), a.kfb + (size_t)h * (M / 2), opaque(tid));
            });
            // FUSED: tail + pair stage + tail; otherwise the pair stage alone, in the same block / mirror-block order
            // (contiguous LDS runs, adjacent table entries)
            pass_tail_pointwise<LOG2M, true, FUSED>(X, a.tw, a.twp, a.kfa + (size_t)h * (M / 2), a.kfb + (size_t)h * (M / 2),
                                                    a.kfs + (size_t)h * 3, opaque(tid), csign, &pre);
            if constexpr (FUSED) th_inv = FftTw<LOG2M, 1>::template theta_at<P::N16 - 1>(a.tw, opaque(tid));
        } else {
            fft_forward_global<LOG2M, NG>(X, u2, Lc, a.tw, W, tid, st);
            static_assert(NG == 1, "one block of 16 positions per thread");
            pass_tail_pointwise<LOG2M, true, false>(X, a.tw, a.twp, a.kfa + (size_t)h * (M / 2), a.kfb + (size_t)h * (M / 2),
                                                    a.kfs + (size_t)h * 3, opaque(tid), csign);
        }
        // (Round 4, measured and dropped, twice: touching the NEXT row's input -- one LDS-DMA dword per lane, 64 bytes apart,
        // into a scratch slot -- so that its top pass would find the points in L2.  Before the pair stage: 87.8 -> 94.5 us
        // (vector-memory loads return in order: the stage's K_f loads queue behind the requests); after it, with no
        // vector-memory instruction for three passes: 77.4 -> 80.4 us (`profiles/r04_ab_fftconv_fused_tail.txt`).)
        __syncthreads();
        st();
        const float Dh = a.D[h];
        c2* __restrict__ g2 = reinterpret_cast<c2*>(a.g + off);
        c2* __restrict__ p2 = a.pre ? reinterpret_cast<c2*>(a.pre + off) : nullptr;
        auto finish = [&](int j, c2 y, c2 uu) {
            const c2 v = mk(fmaf(y.x, scale, Dh * uu.x), fmaf(y.y, scale, Dh * uu.y));
            if (p2) p2[j] = v;
            g2[j] = a.no_act ? v : mk(gelu_f(v.x), gelu_f(v.y));
        };
        if constexpr (DIRECT) {
            // (FUSED: no base twiddle is held over a row -- the one value too many for the 128 registers of a 1024-thread
            // workgroup; they are fetched a pass ahead instead)
            c2 th_bottom = mk(1.f, 0.f);
            if constexpr (FUSED) th_bottom = fft_inverse_chain<LOG2M, 1, P::N16>(X, a.tw, tid, st, th_inv);
            else fft_inverse_to<LOG2M, NG, 1>(X, W, tid, st);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                if (i) __builtin_amdgcn_sched_barrier(0);
                const int g = tid + THREADS * i;
                c2 x[16];
                const c2 th0 = FUSED ? th_bottom : W.theta[0][i];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = X[pidx(g) + (M / 16 + M / 256) * r];   // = pidx(g + (M/16) r)
                fft16<true, true, false, true>(x, th0);
                // buffer instructions clip at the row's end (loads return zero, stores are dropped): no predicate, no
                // branch per output; the row's input once more for the D u term (an L2 hit), requested together
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __amdgpu_buffer_rsrc_t rUu = __builtin_amdgcn_make_buffer_rsrc((void*)(a.u + off), 0, L * 4, 0x00020000);
                __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + off), 0, L * 4, 0x00020000);
                __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)((a.pre ? a.pre : a.g) + off), 0,
                                                                               a.pre ? L * 4 : 0, 0x00020000);
                c2 uu[8];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    uu[r] = __builtin_bit_cast(c2, __builtin_amdgcn_raw_buffer_load_b64(rUu, (g + (M / 16) * r) * 8, 0, 0));
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int vo = (g + (M / 16) * r) * 8;
                    const c2 v = mk(fmaf(x[r].x, scale, Dh * uu[r].x), fmaf(x[r].y, scale, Dh * uu[r].y));
                    if (a.pre) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rP, vo, 0, 0);
                    const c2 o = a.no_act ? v : mk(gelu_f(v.x), gelu_f(v.y));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rG, vo, 0, 0);
                    if constexpr (RSUM) rs += (vo < L * 4) ? (o.x + o.y) : 0.f;      // (L even: a pair lies inside the row or past it)
                }
                if constexpr (RSUM) {     // lanes -> wave (fixed order), waves meet in LDS behind the closing barrier
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) rs += __shfl_xor(rs, off);
                    if ((tid & 63) == 0) rsum_w[tid >> 6] = rs;
                }
            }
            st();
        } else {
            fft_inverse<LOG2M, NG>(X, a.tw, W, tid);
            for (int j = tid; j < Lc; j += THREADS) finish(j, X[pidx(j)], u2[j]);
        }
        __syncthreads();   // the next row overwrites X
        if constexpr (RSUM) {
            if (tid == 0) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < (THREADS + 63) / 64; ++w) t += rsum_w[w];
                a.rowsum[(size_t)b * a.rowsum_bstride + h] = t;
            }
        }
        // (lgkmcnt(0) + s_barrier instead -- nobody needs this row's global stores to have landed, and __syncthreads()
        // waits for them -- measured no different: 80.5 vs 80.6 us)
        st();
    }
}

// Kernel-gradient partials (FftCorrArgs): the block owns channel h and a chunk of the batch; per sample it
// transforms u then dA through the same LDS FFT and accumulates conj(U) * dA for the bins its threads own.
// At M = 16384 the accumulators, the bins of U and a radix-16 pass do not fit in the 128 VGPRs of a 1024-thread workgroup:
// that size runs with 512 threads (two 16-point groups per thread and pass, 256 VGPRs) and parks the bins of U in the
// block's own output slab (coalesced, re-read by the thread that wrote them; results overwrite it after the last sample).
template <int LOG2M, int THREADS>
__global__ __launch_bounds__(THREADS) void fftcorr_kernel(FftCorrArgs a) {
    constexpr int M = 1 << LOG2M, NP = M / 2 / THREADS;
    extern __shared__ __attribute__((aligned(16))) c2 X[];
    const int tid0 = threadIdx.x, h = blockIdx.x, bs = blockIdx.y;
    const int L = a.L, Lc = L / 2;
    constexpr int NG = (1 << LOG2M) / 16 / THREADS;
    FftTw<LOG2M, NG> W;
    W.load(a.tw, tid0);
    constexpr bool PARK = LOG2M >= 14;
    c2 ua[PARK ? 1 : NP], ub[PARK ? 1 : NP], pa[NP], pb[NP];
    c2* __restrict__ o = a.part + ((size_t)bs * a.H + h) * (M + 1);
    float u0 = 0.f, uM = 0.f, p0 = 0.f, pM = 0.f;
    c2 uh = mk(0.f, 0.f), ph = mk(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NP; ++i) pa[i] = pb[i] = mk(0.f, 0.f);
    const int b_end = min(a.B, (bs + 1) * a.bchunk);
    // real-FFT bins of pair q from the bit-reversed half-size spectrum (see pointwise_pairs)
    auto bins = [&](int q, c2& ak, c2& am) {
        const int p = 2 * q;
        const int k = brev(p, LOG2M);
        const int pm = brev(M - k, LOG2M);
        const c2 zk = X[pidx(p)], zm = X[pidx(pm)];
        const c2 wk = a.twp[q];
        const c2 xe = mk(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const c2 d = mk(zk.x - zm.x, zk.y + zm.y);
        const c2 t = cmul_(wk, mk(0.5f * d.y, -0.5f * d.x));
        ak = cadd(xe, t);
        am = cconj(csub(xe, t));
    };
#pragma unroll 1
    for (int b = bs * a.bchunk; b < b_end; ++b) {
        // opaque copies of the thread index per phase: addresses derived from it (pair positions, table offsets) are
        // loop invariant and would otherwise be hoisted and kept in ~100 VGPRs across the transforms
        int tid = opaque(tid0);
        const c2* __restrict__ u2 = reinterpret_cast<const c2*>(a.u + ((size_t)b * a.H + h) * L);
        const c2* __restrict__ d2 = reinterpret_cast<const c2*>(a.da + ((size_t)b * a.H + h) * L);
        fft_forward_global<LOG2M, NG>(X, u2, Lc, a.tw, W, tid);
        tid = opaque(tid0);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * THREADS;
            if (q == 0) {
                const c2 z0 = X[pidx(0)];
                u0 = z0.x + z0.y; uM = z0.x - z0.y; uh = cconj(X[pidx(1)]);   // A[0], A[M], A[M/2]
            } else if (PARK) {
                c2 k1, k2;
                bins(q, k1, k2);
                o[q] = k1;
                o[M / 2 + q] = k2;
            } else {
                bins(q, ua[i], ub[i]);
            }
            if (NP > 8 && (i & 1)) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight (registers)
        }
        __syncthreads();
        tid = opaque(tid0);
        fft_forward_global<LOG2M, NG>(X, d2, Lc, a.tw, W, tid);
        tid = opaque(tid0);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * THREADS;
            if (q == 0) {
                const c2 z0 = X[pidx(0)];
                p0 = fmaf(u0, z0.x + z0.y, p0);
                pM = fmaf(uM, z0.x - z0.y, pM);
                ph = cadd(ph, cmulc(cconj(X[pidx(1)]), uh));
            } else {
                c2 dk, dm;
                bins(q, dk, dm);
                const c2 uka = PARK ? o[q] : ua[i], ukb = PARK ? o[M / 2 + q] : ub[i];
                pa[i] = cadd(pa[i], cmulc(dk, uka));   // dA * conj(U)
                pb[i] = cadd(pb[i], cmulc(dm, ukb));
            }
            if (NP > 8 && (i & 1)) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = tid0 + i * THREADS;
        if (q == 0) {
            o[0] = mk(p0, 0.f);
            o[M] = mk(pM, 0.f);
            o[M / 2] = ph;
        } else {
            const int k = brev(2 * q, LOG2M);
            o[k] = pa[i];
            o[M - k] = pb[i];
        }
    }
}

// The same kernel-gradient partials on the BLOCK / MIRROR-BLOCK pair stage of fftconv_kernel (round 6; even plans, one 16-point
// block per thread).  Per transform the forward radix-4 tail and the real-row bins are ONE pass straight from registers: thread t
// reads the r3 = 0 half of block t and the r3 = 1 half of its mirror block t' (fft_core.h: pass_tail_pointwise -- contiguous,
// conflict-free LDS runs), applies the two tail stages with each half's tail twiddle and holds all eight pairs (x[e], x[15 - e]),
// e even, whose tables are adjacent entries twp[8 t + e/2] / twp[8 t' + e/2].  Nothing is written back to LDS: the separate tail
// pass, its barrier and the eight scattered bit-reversed pair reads per thread of fftcorr_kernel are gone.  The bins of U are parked
// per thread ([j][t]: coalesced) in the block's own output slab at M = 16384 (registers), results overwrite it after the last sample.
template <int LOG2M>
__global__ __launch_bounds__((1 << LOG2M) / 16) void fftcorr_blk_kernel(FftCorrArgs a) {
    using P = FftPlan<LOG2M>;
    static_assert(!P::ODD, "block-order pair stage: even plans");
    constexpr int M = 1 << LOG2M, THREADS = M / 16;
    constexpr bool TAILS = P::TAIL4;
    constexpr bool PARK = LOG2M >= 14;
    extern __shared__ __attribute__((aligned(16))) c2 X[];
    const int tid0 = threadIdx.x, h = blockIdx.x, bs = blockIdx.y;
    const int L = a.L;
    c2 ua[PARK ? 1 : 8], ub[PARK ? 1 : 8], pa[8], pb[8];
    c2* __restrict__ o = a.part + ((size_t)bs * a.H + h) * (M + 1);
    // the self-paired bins (k = 0, M, M/2: thread 0's pair 0) live in LDS, not in eight registers of every thread: the M = 16384
    // instance runs at the 128-VGPR limit of a 1024-thread workgroup.  [u0, uM, uh.x, uh.y | p0, pM, ph.x, ph.y]; thread 0 only.
    __shared__ float selfp[8];
    if (tid0 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) selfp[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) pa[i] = pb[i] = mk(0.f, 0.f);
    const int b_end = min(a.B, (bs + 1) * a.bchunk);
    NoStamp st;

    // forward transform of one zero-padded row up to (not including) the radix-4 tail: fftconv_kernel's sequence
    auto forward = [&](const float* row) {
        const int tid = opaque(tid0);
        __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, L * 4, 0x00020000);
        c2 x[16];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            x[r] = __builtin_bit_cast(c2, __builtin_amdgcn_raw_buffer_load_b64(rU, (tid + (M / 16) * r) * 8, 0, 0));
        const c2 ph1 = fwd_twiddle<LOG2M, 1, false>(a.tw, opaque(tid));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 8; r < 16; ++r) x[r] = mk(0.f, 0.f);
        fft16<false, false, true>(x, mk(1.f, 0.f));
#pragma unroll
        for (int r = 0; r < 16; ++r) X[pidx(tid) + (M / 16 + M / 256) * r] = x[r];
        __syncthreads();
        fft_forward_chain<LOG2M, 1, false>(X, a.tw, tid, st, ph1, [] {});
    };
    // tail (plans with one) + the eight pairs of this thread: f(j, is_self, zk, zm, wk) for j = 0..7; pair j's even position is
    // 16 t + 2 j (j < 4) / 16 t' + 2 j (j >= 4), its table index 8 t + j / 8 t' + j
    auto pairs = [&](auto&& f) {
        const int t = opaque(tid0), tm = mirror_block(t);
        // tables of the four pairs whose even position lies in block t first (their round trip runs under the LDS reads and
        // the tail butterflies); the other four (block t') are requested once the butterflies are done: all eight at once
        // cost eight more live registers at the 128-VGPR limit of a 1024-thread workgroup (pass_tail_pointwise does the same)
        c2 wk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wk[i] = a.twp[8 * t + i];
        c2 x[16];
#pragma unroll
        for (int d = 0; d < 8; ++d) x[d] = X[17 * t + d];
#pragma unroll
        for (int d = 8; d < 16; ++d) x[d] = X[17 * tm + d];
        if constexpr (TAILS) {
            const c2 taul = a.tw[brev_bits(t, LOG2M - 4)], tauh = a.tw[brev_bits(tm, LOG2M - 4)];
            {   // one half at a time: each tail-twiddle table (6 complex) lives only while its half is transformed
                const Tw16<true> tl(opaque(taul));
                stage16_fwd_half<1, 0>(x, tl);
                stage16_fwd_half<0, 0>(x, tl);
            }
            sched_fence();
            {
                const Tw16<true> th(opaque(tauh));
                stage16_fwd_half<1, 1>(x, th);
                stage16_fwd_half<0, 1>(x, th);
            }
            sched_fence();
        }
        c2 wk2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wk2[i] = a.twp[8 * tm + 4 + i];
        if (t == 0) {   // block 0 pairs inside itself (pass_tail_pointwise): route its odd positions into the generic slots
            const c2 x1 = x[1], x3 = x[3], x5 = x[5], x7 = x[7], x9 = x[9], x11 = x[11], x13 = x[13], x15 = x[15];
            x[15] = x1; x[13] = x3; x[11] = x7; x[9] = x5; x[7] = x15; x[5] = x13; x[3] = x11; x[1] = x9;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f(j, j == 0 && t == 0, x[2 * j], x[15 - 2 * j], wk[j]);
            if (j & 1) __builtin_amdgcn_sched_barrier(0);     // bound the parked-bin loads in flight (registers)
        }
#pragma unroll
        for (int j = 4; j < 8; ++j) {
            f(j, false, x[2 * j], x[15 - 2 * j], wk2[j - 4]);
            if (j & 1) __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma unroll 1
    for (int b = bs * a.bchunk; b < b_end; ++b) {
        forward(a.u + ((size_t)b * a.H + h) * L);
        pairs([&](int j, bool self, c2 zk, c2 zm, c2 wk) {
            if (self) {
                selfp[0] = zk.x + zk.y; selfp[1] = zk.x - zk.y; selfp[2] = zm.x; selfp[3] = -zm.y;      // A[0], A[M], A[M/2] = conj(position 1)
            } else {
                c2 k1, k2;
                pair_bins(zk, zm, wk, k1, k2);
                if (PARK) {
                    o[(2 * j) * THREADS + tid0] = k1;
                    o[(2 * j + 1) * THREADS + tid0] = k2;
                } else {
                    ua[PARK ? 0 : j] = k1;
                    ub[PARK ? 0 : j] = k2;
                }
            }
        });
        __syncthreads();     // every thread has read its halves: the next transform may overwrite X
        forward(a.da + ((size_t)b * a.H + h) * L);
        pairs([&](int j, bool self, c2 zk, c2 zm, c2 wk) {
            if (self) {
                selfp[4] = fmaf(selfp[0], zk.x + zk.y, selfp[4]);
                selfp[5] = fmaf(selfp[1], zk.x - zk.y, selfp[5]);
                const c2 ph = cadd(mk(selfp[6], selfp[7]), cmulc(cconj(zm), mk(selfp[2], selfp[3])));
                selfp[6] = ph.x; selfp[7] = ph.y;
            } else {
                c2 dk, dm;
                pair_bins(zk, zm, wk, dk, dm);
                const c2 uka = PARK ? o[(2 * j) * THREADS + tid0] : ua[PARK ? 0 : j];
                const c2 ukb = PARK ? o[(2 * j + 1) * THREADS + tid0] : ub[PARK ? 0 : j];
                pa[j] = cadd(pa[j], cmulc(dk, uka));   // dA * conj(U)
                pb[j] = cadd(pb[j], cmulc(dm, ukb));
            }
        });
        __syncthreads();
    }
    // natural-order bins: pair j of thread t holds A[k], A[M - k], k = brev(its even position)
    {
        const int t = tid0, tm = mirror_block(t);
        if (PARK) __syncthreads();      // (the parked bins of the last sample have been read by their owner only: no hazard; kept cheap)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j == 0 && t == 0) {
                o[0] = mk(selfp[4], 0.f);
                o[M] = mk(selfp[5], 0.f);
                o[M / 2] = mk(selfp[6], selfp[7]);
            } else {
                const int p = (j < 4 ? 16 * t : 16 * tm) + 2 * j;
                const int k = brev(p, LOG2M);
                o[k] = pa[j];
                o[M - k] = pb[j];
            }
        }
    }
}

// Rows longer than the transform (FftConvSegArgs): block = (row, output segment j).  Three input segments (j, j-1, j+1)
// go through the forward transform; their real-row bins times the matching kernel spectrum accumulate in registers
// (16 pairs per thread at 512 threads); the sum is re-packed into LDS, inverse-transformed, and its first half is output
// segment j (+ D u, GELU).  Twice the transform work of a short row, 16-20 bytes of HBM traffic per sample.
template <int LOG2M, int THREADS>
__global__ __launch_bounds__(THREADS) void fftconv_seg_kernel(FftConvSegArgs a) {
    constexpr int M = 1 << LOG2M, S = M, NP = M / 2 / THREADS;
    constexpr int NG = M / 16 / THREADS;
    extern __shared__ __attribute__((aligned(16))) c2 X[];
    const int tid0 = threadIdx.x, row = blockIdx.x, j = blockIdx.y;
    const int h = row / a.B, b = row % a.B;
    const int L = a.L, nseg = (L + S - 1) / S;
    const size_t off = ((size_t)b * a.H + h) * L;
    const c2* __restrict__ u2 = reinterpret_cast<const c2*>(a.u + off);   // L even: samples (2i, 2i+1)
    FftTw<LOG2M, NG> W;
    W.load(a.tw, tid0);
    c2 ya[NP], yb[NP];
    float y0 = 0.f, yM = 0.f;
    c2 yh = mk(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NP; ++i) ya[i] = yb[i] = mk(0.f, 0.f);
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {
        const int sj = j + (t == 0 ? 0 : t == 1 ? -1 : 1);
        if (sj < 0 || sj >= nseg) continue;        // uniform over the block
        int tid = opaque(tid0);
        const int c0 = sj * (S / 2), cn = min(S / 2, L / 2 - c0);   // packed points of this segment that exist
        fft_forward_global<LOG2M, NG>(X, u2 + c0, cn, a.tw, W, tid);
        tid = opaque(tid0);
        const c2* __restrict__ kfa = a.kfa[t] + (size_t)h * (M / 2);
        const c2* __restrict__ kfb = a.kfb[t] + (size_t)h * (M / 2);
        const c2* __restrict__ kfs = a.kfs[t] + (size_t)h * 3;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * THREADS;
            if (q == 0) {
                const c2 z0 = X[pidx(0)];
                y0 = fmaf(z0.x + z0.y, kfs[0].x, y0);
                yM = fmaf(z0.x - z0.y, kfs[1].x, yM);
                yh = cadd(yh, cmul_(cconj(X[pidx(1)]), kfs[2]));
            } else {
                const int p = 2 * q, pm = brev(M - brev(p, LOG2M), LOG2M);
                c2 ak, am;
                pair_bins(X[pidx(p)], X[pidx(pm)], a.twp[q], ak, am);
                ya[i] = cadd(ya[i], cmul_(ak, kfa[q]));
                yb[i] = cadd(yb[i], cmul_(am, kfb[q]));
            }
        }
        __syncthreads();
    }
    {
        const int tid = opaque(tid0);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * THREADS;
            if (q == 0) {
                X[pidx(0)] = mk(0.5f * (y0 + yM), 0.5f * (y0 - yM));
                X[pidx(1)] = cconj(yh);
            } else {
                const int p = 2 * q, pm = brev(M - brev(p, LOG2M), LOG2M);
                c2 zk, zm;
                pair_repack(ya[i], yb[i], a.twp[q], zk, zm);
                X[pidx(p)] = zk;
                X[pidx(pm)] = zm;
            }
        }
    }
    __syncthreads();
    fft_inverse<LOG2M, NG>(X, a.tw, W, opaque(tid0));
    const float scale = 1.f / (float)M, Dh = a.D[h];
    c2* __restrict__ g2 = reinterpret_cast<c2*>(a.g + off);
    const int c0 = j * (S / 2), cn = min(S / 2, L / 2 - c0);
    for (int i = tid0; i < cn; i += THREADS) {
        const c2 y = X[pidx(i)], uu = u2[c0 + i];
        g2[c0 + i] = mk(gelu_f(fmaf(y.x, scale, Dh * uu.x)), gelu_f(fmaf(y.y, scale, Dh * uu.y)));
    }
}

// Forward real FFT only (weight time): spectrum of the re-placed two-sided kernel, natural order
// out[h][k], k = 0..M (Nf/2+1 bins).  Used to build K_f with the SAME transform the convolution uses.
template <int LOG2M, int THREADS>
// (kernel SIGNATURES keep HIP's float2: `c2` is a register-pair vector type in the device pass and float2 in the host pass,
// and a kernel's mangled name must be the same in both)
__global__ __launch_bounds__(THREADS) void rfft_rows_kernel(const float* __restrict__ in, ::float2* __restrict__ out_,
                                                            const ::float2* __restrict__ tw_,
                                                            const ::float2* __restrict__ twn_) {
    c2* __restrict__ out = reinterpret_cast<c2*>(out_);
    const c2* __restrict__ tw = reinterpret_cast<const c2*>(tw_);
    const c2* __restrict__ twn = reinterpret_cast<const c2*>(twn_);
    constexpr int M = 1 << LOG2M;
    extern __shared__ __attribute__((aligned(16))) c2 X[];
    const int tid = threadIdx.x, h = blockIdx.x;
    const c2* __restrict__ r2 = reinterpret_cast<const c2*>(in + (size_t)h * 2 * M);
    constexpr int NG = (1 << LOG2M) / 16 / THREADS;
    FftTw<LOG2M, NG> W;
    W.load(tw, tid);
    for (int j = tid; j < M; j += THREADS) X[pidx(j)] = r2[j];
    __syncthreads();
    fft_forward<LOG2M, NG>(X, tw, W, tid);
    c2* __restrict__ o = out + (size_t)h * (M + 1);
    for (int k = tid; k <= M / 2; k += THREADS) {
        if (k == 0) {
            const c2 z0 = X[pidx(0)];
            o[0] = mk(z0.x + z0.y, 0.f);
            o[M] = mk(z0.x - z0.y, 0.f);
            continue;
        }
        const c2 zk = X[pidx(brev(k, LOG2M))], zm = X[pidx(brev(M - k, LOG2M))];
        const c2 wk = twn[k];  // exp(-2 pi i k / (2M))
        const c2 xe = mk(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const c2 d = mk(zk.x - zm.x, zk.y + zm.y);
        const c2 t = cmul_(wk, mk(0.5f * d.y, -0.5f * d.x));
        o[k] = cadd(xe, t);
        o[M - k] = cconj(csub(xe, t));
    }
}

// Two-sided kernel re-placed for transform size Nf >= 2L (s4.py:1391-1394 generalised):
//   K[j] = k0[j]/L (j < L);  K[Nf - m] = k1[m-1]/L (m = 1..L);  0 elsewhere.
// k rows have length Lk (the kernel's own length); the first Lt = min(run length, Lk) taps of each direction are used
// (`L_kernel`, s4.py:1387,805); 1/Lk is the irfft normalisation the unnormalised C2R left out.
// which = 0: both directions, 1: causal taps only, 2: anti-causal taps only (the segmented long-row path).
__global__ void s4_twosided_pow2_kernel(const float* __restrict__ k, float* __restrict__ K, int H, int Lt, int Nf, int Lk,
                                        int which) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (j >= Nf) return;
    const float inv = 1.f / (float)Lk;
    float v = 0.f;
    if (j < Lt && which != 2) v = k[(size_t)h * Lk + j] * inv;
    else if (j >= Nf - Lt && j >= Lt && which != 1) v = k[((size_t)H + h) * Lk + (Nf - j - 1)] * inv;
    K[(size_t)h * Nf + j] = v;
}

// Pair-ordered copies of the spectrum for the pointwise stage: q -> k = brev(2q):
//   kfa[h][q] = Kf[h][k], kfb[h][q] = Kf[h][M-k];  kfs[h] = {Kf[0], Kf[M], Kf[M/2]}
// sign_alt: bin k is multiplied by (-1)^k, i.e. the kernel is shifted by half the transform (segmented long rows).
__global__ void kf_permute_kernel(const ::float2* __restrict__ kf_, ::float2* __restrict__ kfa_, ::float2* __restrict__ kfb_,
                                  ::float2* __restrict__ kfs_, int log2m, int sign_alt) {
    const c2* __restrict__ kf = reinterpret_cast<const c2*>(kf_);
    c2* __restrict__ kfa = reinterpret_cast<c2*>(kfa_);
    c2* __restrict__ kfb = reinterpret_cast<c2*>(kfb_);
    c2* __restrict__ kfs = reinterpret_cast<c2*>(kfs_);
    const int M = 1 << log2m;
    const int q = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y;
    if (q >= M / 2) return;
    const c2* r = kf + (size_t)h * (M + 1);
    auto bin = [&](int k) {
        const c2 v = r[k];
        return (sign_alt && (k & 1)) ? mk(-v.x, -v.y) : v;
    };
    if (q == 0) {
        kfs[h * 3 + 0] = bin(0); kfs[h * 3 + 1] = bin(M); kfs[h * 3 + 2] = bin(M / 2);
        kfa[(size_t)h * (M / 2)] = bin(0); kfb[(size_t)h * (M / 2)] = bin(M);
        return;
    }
    const int k = brev(2 * q, log2m);
    kfa[(size_t)h * (M / 2) + q] = bin(k);
    kfb[(size_t)h * (M / 2) + q] = bin(M - k);
}

template <int LOG2M>
struct FcCfg {
    static constexpr int M = 1 << LOG2M;
    // one 16-point group / four radix-4 butterflies per thread.  (M = 16384 with 512 threads and two groups per thread --
    // 256 VGPRs, room to overlap one group's LDS traffic with the other's butterflies -- measured 121 us against 87 us.)
    static constexpr int THREADS = M / 16;
    static constexpr size_t LDS = (size_t)(M + M / 16) * 8;
    static constexpr bool FUSED = FftPlan<LOG2M>::TAIL4 && !FftPlan<LOG2M>::ODD;
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

static int cu_count() {
    static int n_dev[DWS_MAX_DEVICES] = {};
    int& n = n_dev[current_device_slot()];
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// DWS_FFT_TRACE=1: the launch runs the stamped instance, waits, and prints the mean ticks per phase (one line per traced row
// index: a workgroup's first row starts cold, the later ones show the steady state) on stderr -- the per-phase budget of a
// row.  s_memtime is per XCD: only differences inside one wave / workgroup are formed.
template <int LOG2M>
static int fft_trace_launch(FftConvArgs a, int nwg, hipStream_t s) {
    using C = FcCfg<LOG2M>;
    static const char* names[12] = {"", "loads landed", "top pass", "fwd P1", "fwd P2", "fwd tail", "pointwise", "inv tail",
                                    "inv P2", "inv P1", "bottom + stores", "closing barrier"};
    static const bool fused = C::FUSED && getenv("DWS_FFT_NO_FUSED_TAIL") == nullptr;
    auto kern = fused ? fftconv_kernel<LOG2M, C::THREADS, true, C::FUSED> : fftconv_kernel<LOG2M, C::THREADS, true>;
    constexpr int NP = FftPlan<LOG2M>::N16 - 1;
    const int T4 = (FftPlan<LOG2M>::TAIL4 && !fused) ? 1 : 0, NST = 3 + NP + T4 + 1 + T4 + NP + 2;   // fused: the pair stage carries both tails
    static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr = true;
    }
    constexpr int WAVES = C::THREADS / 64, PER = FFT_TRACE_ROWS * FFT_TRACE_SLOTS;
    // stamps per row: start, loads, top, the plan's further radix-16 passes, (tail), pointwise, (tail), inverse passes, bottom, barrier
    const size_t n = (size_t)nwg * WAVES * PER;
    unsigned long long* d = nullptr;
    DWS_HIP(hipMalloc(&d, n * 8));
    DWS_HIP(hipMemsetAsync(d, 0, n * 8, s));
    a.trace = d;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(C::THREADS), C::LDS, s, a);
    DWS_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(n);
    DWS_HIP(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    DWS_HIP(hipFree(d));
    for (int ri = 0; ri < FFT_TRACE_ROWS; ++ri) {
        double ph[FFT_TRACE_SLOTS] = {0}, life = 0, span = 0;
        long waves = 0, wgs = 0;
        for (int g = 0; g < nwg; ++g) {
            unsigned long long g0 = ~0ull, g1 = 0;
            for (int w = 0; w < WAVES; ++w) {
                const unsigned long long* t = &h[((size_t)g * WAVES + w) * PER + (size_t)ri * FFT_TRACE_SLOTS];
                if (t[0] == 0 || t[NST - 1] == 0) continue;    // this workgroup walked fewer rows
                for (int i = 1; i < NST; ++i) ph[i] += (double)(t[i] - t[i - 1]);
                life += (double)(t[NST - 1] - t[0]);
                g0 = std::min(g0, t[0]);
                g1 = std::max(g1, t[NST - 1]);
                ++waves;
            }
            if (g1 > g0 && g0 != ~0ull) { span += (double)(g1 - g0); ++wgs; }
        }
        if (!waves) continue;
        fprintf(stderr, "[fft trace] M=%d L=%d rows=%d wgs=%d row#%d (%ld waves) mean ticks:", 1 << LOG2M, a.L, a.B * a.H, nwg, ri, waves);
        int name = 1;
        for (int i = 1; i < NST; ++i) {
            // names[] lists the M = 16384 plan (two further passes, a tail); shorter plans skip the entries they do not have
            const char* nm = names[std::min(name, 11)];
            fprintf(stderr, " %s %.0f |", nm, ph[i] / waves);
            ++name;
            if (NP < 2 && name == 4) name = 5;
            if (!T4 && (name == 5 || name == 7)) ++name;
            if (NP < 2 && name == 8) name = 9;
        }
        fprintf(stderr, " row %.0f, workgroup span %.0f\n", life / waves, span / std::max(1l, wgs));
    }
    return DWS_OK;
}

template <int LOG2M>
static int launch_fc(const FftConvArgs& a_in, hipStream_t s) {
    using C = FcCfg<LOG2M>;
    // A/B switch, read once: DWS_FFT_NO_FUSED_TAIL=1 (separate tail / pair / tail passes, scalar pair arithmetic)
    static const bool fused = C::FUSED && getenv("DWS_FFT_NO_FUSED_TAIL") == nullptr;
    auto kern = fused ? fftconv_kernel<LOG2M, C::THREADS, false, C::FUSED> : fftconv_kernel<LOG2M, C::THREADS>;
    if constexpr (!FftPlan<LOG2M>::ODD) {
        if (a_in.rowsum) kern = fftconv_kernel<LOG2M, C::THREADS, false, C::FUSED, true>;     // the training adjoint pass
    } else {
        DWS_CHECK(a_in.rowsum == nullptr, DWS_ERR_UNSUPPORTED, "fftconv: row sums need an even plan (log2 M = %d)", LOG2M);
    }
    static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
    if (!attr) {
        if constexpr (!FftPlan<LOG2M>::ODD)
            DWS_HIP(hipFuncSetAttribute((const void*)fftconv_kernel<LOG2M, C::THREADS, false, C::FUSED, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        DWS_HIP(hipFuncSetAttribute((const void*)fftconv_kernel<LOG2M, C::THREADS, false, C::FUSED>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        DWS_HIP(hipFuncSetAttribute((const void*)fftconv_kernel<LOG2M, C::THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)C::LDS));
        attr = true;
    }
    const FftConvArgs& a = a_in;
    // one resident workgroup per LDS slot of every CU walks several rows (RowSchedule); small rows: one row per block
    // (the persistent schedule from M = 4096 / 1024 up: 32.0-32.2 vs 31.7-31.9 us and 16.4 vs 16.3 us, no gain)
    const int rows = a.B * a.H;
    const int slots = cu_count() * std::max(1, std::min((int)(160 * 1024 / C::LDS), 2048 / C::THREADS));
    const int nwg = (std::min(rows, LOG2M >= 13 ? slots : rows) + 7) / 8 * 8;
    if constexpr (!FftPlan<LOG2M>::ODD) {
        static const bool trace = getenv("DWS_FFT_TRACE") != nullptr;
        if (trace) return fft_trace_launch<LOG2M>(a, nwg, s);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(C::THREADS), C::LDS, s, a);
    return DWS_OK;
}

template <int LOG2M>
static int launch_fcorr(const FftCorrArgs& a, hipStream_t s) {
    using C = FcCfg<LOG2M>;
    // M = 16384: one 16-point group per thread (1024 threads, 111 VGPRs with the packed FFT core and the bins of U parked
    // in the output slab) -- rounds 2-3 ran two groups per thread on 512 threads there (the scalar core needed 256 VGPRs).
    // DWS_FFTCORR_512=1 keeps that shape (A/B runs).
    static const bool half = std::getenv("DWS_FFTCORR_512") != nullptr;
    static const bool old = std::getenv("DWS_FFTCORR_OLD") != nullptr;      // same-box A/B: the scattered pair stage of rounds 1-5
    if constexpr (!FftPlan<LOG2M>::ODD) {
        if (!old && !half) {
            auto kern = fftcorr_blk_kernel<LOG2M>;
            static bool attrb_dev[DWS_MAX_DEVICES] = {};
            bool& attrb = attrb_dev[current_device_slot()];
            if (!attrb) {
                DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
                attrb = true;
            }
            hipLaunchKernelGGL(kern, dim3(a.H, ceil_div(a.B, a.bchunk)), dim3(C::M / 16), C::LDS, s, a);
            return DWS_OK;
        }
    }
    if (LOG2M >= 14 && half) {
        constexpr int TH = C::THREADS > 512 ? 512 : C::THREADS;
        auto kern = fftcorr_kernel<LOG2M, TH>;
        static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
        if (!attr) {
            DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(a.H, ceil_div(a.B, a.bchunk)), dim3(TH), C::LDS, s, a);
        return DWS_OK;
    }
    constexpr int TH = C::THREADS;
    auto kern = fftcorr_kernel<LOG2M, TH>;
    static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.H, ceil_div(a.B, a.bchunk)), dim3(TH), C::LDS, s, a);
    return DWS_OK;
}

template <int LOG2M>
static int launch_rf(const float* in, float* out, const float* tw, const float* twn, int H, hipStream_t s) {
    using C = FcCfg<LOG2M>;
    auto kern = rfft_rows_kernel<LOG2M, C::THREADS>;
    static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(H), dim3(C::THREADS), C::LDS, s, in, (::float2*)out, (const ::float2*)tw,
                       (const ::float2*)twn);
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

bool fftconv_rowsum_supported(int log2m) { return log2m >= 10 && log2m <= 14 && (log2m & 1) == 0; }

int launch_fftcorr(int log2m, const FftCorrArgs& a, hipStream_t s) {
    ProfileScope ps("fftcorr", s);
    DWS_FC_DISPATCH(launch_fcorr, a, s);
}

int launch_rfft_rows(int log2m, const float* in, float* out, const float* tw, const float* twn, int H,
                     hipStream_t s) {
    DWS_FC_DISPATCH(launch_rf, in, out, tw, twn, H, s);
}

int launch_s4_twosided_pow2_part(const float* k, float* K, int H, int Lt, int Nf, int Lk, int which, hipStream_t s) {
    hipLaunchKernelGGL(s4_twosided_pow2_kernel, dim3(ceil_div(Nf, 256), H), dim3(256), 0, s, k, K, H, Lt, Nf, Lk, which);
    return DWS_OK;
}

int launch_s4_twosided_pow2(const float* k, float* K, int H, int Lt, int Nf, int Lk, hipStream_t s) {
    return launch_s4_twosided_pow2_part(k, K, H, Lt, Nf, Lk, 0, s);
}

int launch_kf_permute_signed(const float* kf, float* kfa, float* kfb, float* kfs, int H, int log2m, int sign_alt,
                             hipStream_t s) {
    const int M = 1 << log2m;
    hipLaunchKernelGGL(kf_permute_kernel, dim3(ceil_div(M / 2, 256), H), dim3(256), 0, s, (const ::float2*)kf,
                       (::float2*)kfa, (::float2*)kfb, (::float2*)kfs, log2m, sign_alt);
    return DWS_OK;
}

int launch_kf_permute(const float* kf, float* kfa, float* kfb, float* kfs, int H, int log2m, hipStream_t s) {
    return launch_kf_permute_signed(kf, kfa, kfb, kfs, H, log2m, 0, s);
}

bool fftconv_seg_supported(int L, int taps) {
    return L > (1 << FFTCONV_SEG_LOG2M) && (L & 1) == 0 && taps <= (1 << FFTCONV_SEG_LOG2M);
}

int launch_fftconv_seg(const FftConvSegArgs& a, hipStream_t s) {
    using C = FcCfg<FFTCONV_SEG_LOG2M>;
    constexpr int TH = 512, S = 1 << FFTCONV_SEG_LOG2M;
    ProfileScope ps("fftconv_seg", s);
    auto kern = fftconv_seg_kernel<FFTCONV_SEG_LOG2M, TH>;
    static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.B * a.H, ceil_div(a.L, S)), dim3(TH), C::LDS, s, a);
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
