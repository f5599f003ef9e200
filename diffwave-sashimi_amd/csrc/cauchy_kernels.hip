// Cauchy multiply for gfx950 -- the replacement of the reference's only native
// code (`extensions/cauchy/cauchy_cuda.cu`).  Wave64 throughout: the block
// reductions use 64-wide shuffles and an LDS scratch of blockDim/64 entries
// (the reference's BlockReduceSum assumes 32-lane warps, `cauchy_cuda.cu:441-442,472-476`).
//
// VALU-bound (2 complex reciprocals x N per output, ~no memory traffic): each
// thread keeps IPT outputs in registers so one LDS broadcast read of
// (v_n, w_n) feeds IPT x ~24 VALU ops.
#include "dws_common.h"

namespace dws {

constexpr int CAUCHY_MAX_N = 1024;

__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// out[b,l] = sum_n v/(z-w)  (+ conj(v)/(z-conj(w)) if SYM)
//   `cauchy_cuda.cu:44-115` (non-symmetric), `:242-347` (symmetric, the one the model uses, `s4.py:758`)
template <bool SYM, int IPT>
__global__ __launch_bounds__(256) void cauchy_fwd_kernel(const float2* __restrict__ v, const float2* __restrict__ z,
                                                         const float2* __restrict__ w, float2* __restrict__ out,
                                                         int N, int L, int wmod) {
    __shared__ float4 vw[CAUCHY_MAX_N];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int bw = wmod ? b % wmod : b;  // w broadcast over leading dims of v (s4.py:752-758: v (2,3,H,N), w (H,N))
    for (int n = tid; n < N; n += 256) {
        const float2 vv = v[(size_t)b * N + n], ww = w[(size_t)bw * N + n];
        vw[n] = make_float4(vv.x, vv.y, ww.x, ww.y);
    }
    // Two outputs per packed lane pair: plain fp32 FMAs issue at 4 cycles per wave64 instruction on gfx950, v_pk_fma_f32
    // does two per lane in 4.6 (tools/valu_rate.hip), so the arithmetic is written on float2 vectors (v_pk_*_f32).
    constexpr int IP = (IPT + 1) / 2;
    v2f zr[IP], zi[IP], ar[IP], ai[IP];
    const int l0 = blockIdx.y * 256 * IPT + tid;
#pragma unroll
    for (int i = 0; i < IP; ++i) {
        float2 zz[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int l = l0 + (2 * i + h) * 256;
            zz[h] = (2 * i + h < IPT && l < L) ? z[l] : make_float2(1.f, 0.f);
        }
        zr[i] = v2f{zz[0].x, zz[1].x}; zi[i] = v2f{zz[0].y, zz[1].y};
        ar[i] = v2f{0.f, 0.f}; ai[i] = v2f{0.f, 0.f};
    }
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        const float4 p = vw[n];  // LDS broadcast
        const v2f vx = {p.x, p.x}, vy = {p.y, p.y}, wx = {p.z, p.z}, wy = {p.w, p.w};
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            const v2f dr = zr[i] - wx;
            const v2f d1 = zi[i] - wy;                         // z - w
            const v2f dr2 = dr * dr;
            v2f n1 = pk_fma(d1, d1, dr2);
            const v2f i1 = v2f{rcp_(n1.x), rcp_(n1.y)};
            const v2f xr = vx * dr, yr = vy * dr;
            // v * conj(d) / |d|^2
            ar[i] = pk_fma(pk_fma(vy, d1, xr), i1, ar[i]);
            ai[i] = pk_fma(pk_fma(-vx, d1, yr), i1, ai[i]);
            if (SYM) {
                const v2f d2 = zi[i] + wy;                     // z - conj(w)
                v2f n2 = pk_fma(d2, d2, dr2);
                const v2f i2 = v2f{rcp_(n2.x), rcp_(n2.y)};
                // conj(v) * conj(d2) / |d2|^2
                ar[i] = pk_fma(pk_fma(-vy, d2, xr), i2, ar[i]);
                ai[i] = pk_fma(-pk_fma(vx, d2, yr), i2, ai[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < IP; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int l = l0 + (2 * i + h) * 256;
            if (2 * i + h < IPT && l < L) out[(size_t)b * L + l] = make_float2(h ? ar[i].y : ar[i].x, h ? ai[i].y : ai[i].x);
        }
}

__device__ __forceinline__ float wave_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));  // row_mirror
    const int xi = __builtin_bit_cast(int, x);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48)));
}

// dv, dw for NG consecutive n of one batch row; the block's NT threads stride over l (NT = 64 for short rows: a single
// wave, no cross-wave pass), the next (z, dout) pair is requested before the current one is consumed.
//   SYM  (`cauchy_cuda.cu:377-449`): term1 = dout/(conj z - conj w), term2 = conj(dout)/(z - conj w)
//        dv = sum term1 + term2 ; dw = conj(v) * sum term1/(conj z - conj w) + term2/(z - conj w)
//   !SYM (`cauchy_cuda.cu:141-209`): q = 1/conj(z - w); dv = sum dout*q ; dw = conj(v) * sum dout*q*q
template <bool SYM, int NG, int NT>
__global__ __launch_bounds__(NT) void cauchy_bwd_kernel(const float2* __restrict__ v, const float2* __restrict__ z,
                                                        const float2* __restrict__ w, const float2* __restrict__ dout,
                                                        float2* __restrict__ dv, float2* __restrict__ dw, int N,
                                                        int L, int wmod, long long nrows) {
    // Block -> (row, n group): the N / NG blocks that re-read one dout row run back to back on ONE XCD (workgroups go
    // to the XCDs round-robin, block i to XCD i % 8), so the row comes from HBM once and from that XCD's L2 afterwards.
    // In (row, group) launch order the re-reads are B blocks apart and, at 6H x 8001 bins, miss every cache.
    const long long bid = blockIdx.x + (long long)blockIdx.y * gridDim.x;
    const int groups = (N + NG - 1) / NG;
    const long long q = bid >> 3;
    const long long b = (q / groups) * 8 + (bid & 7);
    if (b >= nrows) return;
    const int n0 = (int)(q % groups) * NG, tid = threadIdx.x;
    const long long bw = wmod ? b % wmod : b;   // w broadcast over leading dims of v; dw stays per row of v
    static_assert(NG % 2 == 0, "n in packed pairs");
    constexpr int NP = NG / 2;
    v2f wr[NP], wi[NP], svr[NP], svi[NP], swr[NP], swi[NP];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        const float2 w0 = w[(size_t)bw * N + min(n0 + 2 * g, N - 1)], w1 = w[(size_t)bw * N + min(n0 + 2 * g + 1, N - 1)];
        wr[g] = v2f{w0.x, w1.x}; wi[g] = v2f{w0.y, w1.y};
        svr[g] = svi[g] = swr[g] = swi[g] = v2f{0.f, 0.f};
    }
    const float2* __restrict__ drow = dout + (size_t)b * L;
    // a lane past the end never enters the loop; the values below only keep the prefetch registers defined
    float2 zz = make_float2(1.f, 0.f), dd = make_float2(0.f, 0.f);
    if (tid < L) { zz = z[tid]; dd = drow[tid]; }
    // the first pair is complete before the loop: otherwise hipcc's waitcnt pass, merging the preheader's pending loads
    // into the loop header, waits for the pair requested in the SAME iteration at the top of the body (no prefetch left)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int l = tid; l < L; l += NT) {
        float2 zn = make_float2(1.f, 0.f), dn = make_float2(0.f, 0.f);
        if (l + NT < L) { zn = z[l + NT]; dn = drow[l + NT]; }
        const v2f zx = {zz.x, zz.x}, zy = {zz.y, zz.y}, dx = {dd.x, dd.x}, dy = {dd.y, dd.y};
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            if (SYM) {
                // denom_1 = conj(z) - conj(w) = er - i a,  a = zi - wi;   denom_2 = z - conj(w) = er + i e2,  e2 = zi + wi
                // q1 = 1/denom_1 = (er + i a)/|.|^2,  q2 = 1/denom_2 = (er - i e2)/|.|^2
                const v2f er = zx - wr[g], a1 = zy - wi[g], e2 = zy + wi[g];
                const v2f er2 = er * er;
                const v2f m1 = pk_fma(a1, a1, er2), m2 = pk_fma(e2, e2, er2);
                const v2f i1 = v2f{rcp_(m1.x), rcp_(m1.y)}, i2 = v2f{rcp_(m2.x), rcp_(m2.y)};
                const v2f q1r = er * i1, q1i = a1 * i1, q2r = er * i2, q2i = -e2 * i2;
                // term_1 = dout * q1, term_2 = conj(dout) * q2
                const v2f t1r = pk_fma(dx, q1r, -dy * q1i), t1i = pk_fma(dx, q1i, dy * q1r);
                const v2f t2r = pk_fma(dx, q2r, dy * q2i), t2i = pk_fma(dx, q2i, -dy * q2r);
                svr[g] += t1r + t2r;
                svi[g] += t1i + t2i;
                swr[g] = pk_fma(t1r, q1r, pk_fma(-t1i, q1i, pk_fma(t2r, q2r, pk_fma(-t2i, q2i, swr[g]))));
                swi[g] = pk_fma(t1r, q1i, pk_fma(t1i, q1r, pk_fma(t2r, q2i, pk_fma(t2i, q2r, swi[g]))));
            } else {
                // q = 1/conj(z - w) = (z - w)/|z - w|^2
                const v2f er = zx - wr[g], ei = zy - wi[g];
                const v2f m1 = pk_fma(er, er, ei * ei);
                const v2f i1 = v2f{rcp_(m1.x), rcp_(m1.y)};
                const v2f qr = er * i1, qi = ei * i1;
                const v2f pr = pk_fma(dx, qr, -dy * qi), pi = pk_fma(dx, qi, dy * qr);
                svr[g] += pr;
                svi[g] += pi;
                swr[g] = pk_fma(pr, qr, pk_fma(-pi, qi, swr[g]));
                swi[g] = pk_fma(pr, qi, pk_fma(pi, qr, swi[g]));
            }
        }
        zz = zn; dd = dn;
    }
    constexpr int NW = NT / 64;
    __shared__ float red[NW][NG][4];
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float a = wave_sum(g & 1 ? svr[g / 2].y : svr[g / 2].x), bq = wave_sum(g & 1 ? svi[g / 2].y : svi[g / 2].x);
        const float c = wave_sum(g & 1 ? swr[g / 2].y : swr[g / 2].x), dq = wave_sum(g & 1 ? swi[g / 2].y : swi[g / 2].x);
        if (lane == 0) {
            red[wave][g][0] = a; red[wave][g][1] = bq; red[wave][g][2] = c; red[wave][g][3] = dq;
        }
    }
    __syncthreads();
    if (tid < NG && n0 + tid < N) {
        float s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = red[0][tid][k];
#pragma unroll
            for (int q = 1; q < NW; ++q) s[k] += red[q][tid][k];
        }
        const int n = n0 + tid;
        const float2 vv = v[(size_t)b * N + n];
        dv[(size_t)b * N + n] = make_float2(s[0], s[1]);
        // dw = t_dw * conj(v)
        dw[(size_t)b * N + n] = make_float2(s[2] * vv.x + s[3] * vv.y, s[3] * vv.x - s[2] * vv.y);
    }
}

static int check_shapes(const char* fn, const void* a, const void* b, const void* c, const void* d, int64_t B,
                        int64_t N, int64_t L) {
    DWS_CHECK(B >= 0 && N >= 1 && L >= 0, DWS_ERR_INVALID, "%s: bad shape B=%lld N=%lld L=%lld", fn, (long long)B,
              (long long)N, (long long)L);
    // empty tensors legitimately carry null data pointers
    DWS_CHECK((a && b && c && d) || B == 0 || L == 0, DWS_ERR_INVALID, "%s: null pointer", fn);
    DWS_CHECK(N <= CAUCHY_MAX_N, DWS_ERR_UNSUPPORTED, "%s: N=%lld > %d is not supported (`cauchy.py:95-98`)", fn,
              (long long)N, CAUCHY_MAX_N);
    DWS_CHECK(L <= (int64_t)1 << 31 && B <= 65535LL * 32768LL, DWS_ERR_UNSUPPORTED,
              "%s: only L <= 2^31 is supported (`cauchy.py:99-101`)", fn);
    return DWS_OK;
}

template <bool SYM>
static int cauchy_fwd(const float* v, const float* z, const float* w, float* out, int64_t B, int64_t N, int64_t L,
                      hipStream_t s, int wmod = 0) {
    DWS_TRY(check_shapes(SYM ? "cauchy_mult_sym_fwd" : "cauchy_mult_fwd", v, z, w, out, B, N, L));
    if (B == 0 || L == 0) return DWS_OK;
    ProfileScope ps(SYM ? "cauchy_sym_fwd" : "cauchy_fwd", s);
    // outputs per thread: 4 (one LDS read of (v_n, w_n) feeds four bins) unless the row length would leave more than
    // ~7 % of the lanes of the last block idle (L = 501 bins: two per thread, not four)
    auto padded = [&](int ipt) { return (int64_t)ceil_div(L, 256 * ipt) * 256 * ipt; };
    const int ipt = padded(4) * 100 <= padded(1) * 107 ? 4 : padded(2) * 100 <= padded(1) * 107 ? 2 : 1;
    // grid.x is limited to 2^31-1, grid.y to 65535
    DWS_CHECK(ceil_div(L, 256 * ipt) <= 65535, DWS_ERR_UNSUPPORTED, "L too large for one launch");
    dim3 grid((unsigned)B, (unsigned)ceil_div(L, 256 * ipt));
#define DWS_CAUCHY_FWD(IPT)                                                                                             \
    hipLaunchKernelGGL((cauchy_fwd_kernel<SYM, IPT>), grid, dim3(256), 0, s, (const float2*)v, (const float2*)z,         \
                       (const float2*)w, (float2*)out, (int)N, (int)L, wmod)
    if (ipt == 4) DWS_CAUCHY_FWD(4);
    else if (ipt == 2) DWS_CAUCHY_FWD(2);
    else DWS_CAUCHY_FWD(1);
#undef DWS_CAUCHY_FWD
    DWS_HIP(hipGetLastError());
    return DWS_OK;
}

template <bool SYM>
static int cauchy_bwd(const float* v, const float* z, const float* w, const float* dout, float* dv, float* dw,
                      int64_t B, int64_t N, int64_t L, hipStream_t s, int wmod = 0) {
    DWS_TRY(check_shapes(SYM ? "cauchy_mult_sym_bwd" : "cauchy_mult_bwd", v, z, w, dout, B, N, L));
    if (B == 0) return DWS_OK;
    DWS_CHECK(dv && dw, DWS_ERR_INVALID, "cauchy bwd: null output");
    DWS_CHECK(L == 0 || (v && z && w && dout), DWS_ERR_INVALID, "cauchy bwd: null pointer");
    ProfileScope ps(SYM ? "cauchy_sym_bwd" : "cauchy_bwd", s);
    // n per thread / threads per block (experiments: DWS_CAUCHY_BWD_NG = 2|4|8, DWS_CAUCHY_BWD_NT = 64|128|256).  Measured at
    // the config-5 shapes (tools/cauchy_times.py): 8 n per thread 5 % ahead of 4; single-wave blocks ahead up to 2001 bins
    // (no cross-wave pass), level with 256 threads at 8001.
    static const int ng_env = getenv("DWS_CAUCHY_BWD_NG") ? atoi(getenv("DWS_CAUCHY_BWD_NG")) : 8;
    static const int nt_env = getenv("DWS_CAUCHY_BWD_NT") ? atoi(getenv("DWS_CAUCHY_BWD_NT")) : 0;
    const int NG = ng_env;
    const long long blocks = (long long)ceil_div(B, 8) * 8 * ceil_div(N, NG);   // rows padded to whole sets of 8 (one per XCD)
    const unsigned gx = (unsigned)std::min<long long>(blocks, 1ll << 30);
    dim3 grid(gx, (unsigned)ceil_div(blocks, (long long)gx));
    DWS_CHECK(grid.y <= 65535, DWS_ERR_UNSUPPORTED, "cauchy bwd: B * N too large for one launch");
    const int nt = nt_env ? nt_env : (L <= 4096 ? 64 : 256);
#define DWS_CAUCHY_BWD(NG_, NT_)                                                                                       \
    hipLaunchKernelGGL((cauchy_bwd_kernel<SYM, NG_, NT_>), grid, dim3(NT_), 0, s, (const float2*)v, (const float2*)z,   \
                       (const float2*)w, (const float2*)dout, (float2*)dv, (float2*)dw, (int)N, (int)L, wmod, (long long)B)
    if (NG == 8) { if (nt == 64) DWS_CAUCHY_BWD(8, 64); else if (nt == 128) DWS_CAUCHY_BWD(8, 128); else DWS_CAUCHY_BWD(8, 256); }
    else if (NG == 2) { if (nt == 64) DWS_CAUCHY_BWD(2, 64); else if (nt == 128) DWS_CAUCHY_BWD(2, 128); else DWS_CAUCHY_BWD(2, 256); }
    else { if (nt == 64) DWS_CAUCHY_BWD(4, 64); else if (nt == 128) DWS_CAUCHY_BWD(4, 128); else DWS_CAUCHY_BWD(4, 256); }
#undef DWS_CAUCHY_BWD
    DWS_HIP(hipGetLastError());
    return DWS_OK;
}

// internal: symmetric forward with w[B % wmod] (no materialised broadcast of w)
int launch_cauchy_sym_fwd_bcast(const float* v, const float* z, const float* w, float* out, int64_t B, int64_t N,
                                int64_t L, int wmod, hipStream_t s) {
    return cauchy_fwd<true>(v, z, w, out, B, N, L, s, wmod);
}

// internal: symmetric backward with w[B % wmod]; dw is per row of v (the caller sums the broadcast)
int launch_cauchy_sym_bwd_bcast(const float* v, const float* z, const float* w, const float* dout, float* dv, float* dw,
                                int64_t B, int64_t N, int64_t L, int wmod, hipStream_t s) {
    return cauchy_bwd<true>(v, z, w, dout, dv, dw, B, N, L, s, wmod);
}

}  // namespace dws

extern "C" {

int dws_cauchy_sym_fwd(const float* v, const float* z, const float* w, float* out, int64_t B, int64_t N, int64_t L,
                       void* stream) {
    return dws::cauchy_fwd<true>(v, z, w, out, B, N, L, (hipStream_t)stream);
}
int dws_cauchy_fwd(const float* v, const float* z, const float* w, float* out, int64_t B, int64_t N, int64_t L,
                   void* stream) {
    return dws::cauchy_fwd<false>(v, z, w, out, B, N, L, (hipStream_t)stream);
}
int dws_cauchy_sym_bwd(const float* v, const float* z, const float* w, const float* dout, float* dv, float* dw,
                       int64_t B, int64_t N, int64_t L, void* stream) {
    return dws::cauchy_bwd<true>(v, z, w, dout, dv, dw, B, N, L, (hipStream_t)stream);
}
int dws_cauchy_bwd(const float* v, const float* z, const float* w, const float* dout, float* dv, float* dw,
                   int64_t B, int64_t N, int64_t L, void* stream) {
    return dws::cauchy_bwd<false>(v, z, w, dout, dv, dw, B, N, L, (hipStream_t)stream);
}

}  // extern "C"
