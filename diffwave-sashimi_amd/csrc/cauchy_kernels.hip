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
    float zr[IPT], zi[IPT], ar[IPT], ai[IPT];
    const int l0 = blockIdx.y * 256 * IPT + tid;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int l = l0 + i * 256;
        const float2 zz = (l < L) ? z[l] : make_float2(1.f, 0.f);
        zr[i] = zz.x; zi[i] = zz.y; ar[i] = 0.f; ai[i] = 0.f;
    }
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        const float4 p = vw[n];  // LDS broadcast
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const float dr = zr[i] - p.z;
            const float d1 = zi[i] - p.w;                      // z - w
            const float i1 = rcp_(fmaf(dr, dr, d1 * d1));
            // v * conj(d) / |d|^2
            ar[i] = fmaf(fmaf(p.x, dr, p.y * d1), i1, ar[i]);
            ai[i] = fmaf(fmaf(p.y, dr, -p.x * d1), i1, ai[i]);
            if (SYM) {
                const float d2 = zi[i] + p.w;                  // z - conj(w)
                const float i2 = rcp_(fmaf(dr, dr, d2 * d2));
                // conj(v) * conj(d2) / |d2|^2
                ar[i] = fmaf(fmaf(p.x, dr, -p.y * d2), i2, ar[i]);
                ai[i] = fmaf(-fmaf(p.y, dr, p.x * d2), i2, ai[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int l = l0 + i * 256;
        if (l < L) out[(size_t)b * L + l] = make_float2(ar[i], ai[i]);
    }
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// dv, dw for NG consecutive n of one batch row; threads stride over l.
//   SYM  (`cauchy_cuda.cu:377-449`): term1 = dout/(conj z - conj w), term2 = conj(dout)/(z - conj w)
//        dv = sum term1 + term2 ; dw = conj(v) * sum term1/(conj z - conj w) + term2/(z - conj w)
//   !SYM (`cauchy_cuda.cu:141-209`): q = 1/conj(z - w); dv = sum dout*q ; dw = conj(v) * sum dout*q*q
template <bool SYM, int NG>
__global__ __launch_bounds__(256) void cauchy_bwd_kernel(const float2* __restrict__ v, const float2* __restrict__ z,
                                                         const float2* __restrict__ w, const float2* __restrict__ dout,
                                                         float2* __restrict__ dv, float2* __restrict__ dw, int N,
                                                         int L, int wmod) {
    const int b = blockIdx.x, n0 = blockIdx.y * NG, tid = threadIdx.x;
    const int bw = wmod ? b % wmod : b;   // w broadcast over leading dims of v; dw stays per row of v
    float wr[NG], wi[NG], svr[NG], svi[NG], swr[NG], swi[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int n = min(n0 + g, N - 1);
        const float2 ww = w[(size_t)bw * N + n];
        wr[g] = ww.x; wi[g] = ww.y;
        svr[g] = svi[g] = swr[g] = swi[g] = 0.f;
    }
    for (int l = tid; l < L; l += 256) {
        const float2 zz = z[l], dd = dout[(size_t)b * L + l];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (SYM) {
                // denom_1 = conj(z) - conj(w) = (zr - wr) - i (zi - wi);  denom_2 = z - conj(w) = (zr - wr) + i (zi + wi)
                const float er = zz.x - wr[g];
                const float e1 = -(zz.y - wi[g]);
                const float e2 = zz.y + wi[g];
                const float i1 = rcp_(fmaf(er, er, e1 * e1)), i2 = rcp_(fmaf(er, er, e2 * e2));
                // q1 = 1/denom_1 = conj(denom_1)/|.|^2, q2 likewise
                const float q1r = er * i1, q1i = -e1 * i1, q2r = er * i2, q2i = -e2 * i2;
                // term_1 = dout * q1, term_2 = conj(dout) * q2
                const float t1r = dd.x * q1r - dd.y * q1i, t1i = dd.x * q1i + dd.y * q1r;
                const float t2r = dd.x * q2r + dd.y * q2i, t2i = dd.x * q2i - dd.y * q2r;
                svr[g] += t1r + t2r;
                svi[g] += t1i + t2i;
                swr[g] += (t1r * q1r - t1i * q1i) + (t2r * q2r - t2i * q2i);
                swi[g] += (t1r * q1i + t1i * q1r) + (t2r * q2i + t2i * q2r);
            } else {
                // q = 1/conj(z - w) = (z - w)/|z - w|^2
                const float er = zz.x - wr[g], ei = zz.y - wi[g];
                const float i1 = rcp_(fmaf(er, er, ei * ei));
                const float qr = er * i1, qi = ei * i1;
                const float pr = dd.x * qr - dd.y * qi, pi = dd.x * qi + dd.y * qr;
                svr[g] += pr;
                svi[g] += pi;
                swr[g] += pr * qr - pi * qi;
                swi[g] += pr * qi + pi * qr;
            }
        }
    }
    __shared__ float red[4][NG][4];
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float a = wave_sum(svr[g]), bq = wave_sum(svi[g]), c = wave_sum(swr[g]), dq = wave_sum(swi[g]);
        if (lane == 0) {
            red[wave][g][0] = a; red[wave][g][1] = bq; red[wave][g][2] = c; red[wave][g][3] = dq;
        }
    }
    __syncthreads();
    if (tid < NG && n0 + tid < N) {
        float s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = red[0][tid][k] + red[1][tid][k] + red[2][tid][k] + red[3][tid][k];
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
    constexpr int IPT = 4;
    // grid.x is limited to 2^31-1, grid.y to 65535
    DWS_CHECK(ceil_div(L, 256 * IPT) <= 65535, DWS_ERR_UNSUPPORTED, "L too large for one launch");
    dim3 grid((unsigned)B, (unsigned)ceil_div(L, 256 * IPT));
    hipLaunchKernelGGL((cauchy_fwd_kernel<SYM, IPT>), grid, dim3(256), 0, s, (const float2*)v, (const float2*)z,
                       (const float2*)w, (float2*)out, (int)N, (int)L, wmod);
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
    constexpr int NG = 4;
    dim3 grid((unsigned)B, (unsigned)ceil_div(N, NG));
    hipLaunchKernelGGL((cauchy_bwd_kernel<SYM, NG>), grid, dim3(256), 0, s, (const float2*)v, (const float2*)z,
                       (const float2*)w, (const float2*)dout, (float2*)dv, (float2*)dw, (int)N, (int)L, wmod);
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
