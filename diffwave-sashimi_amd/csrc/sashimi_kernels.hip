// SaShiMi / S4 kernels for gfx950 (everything except the FFTs, which are rocFFT
// plans driven from sashimi_model.hip, and the Cauchy multiply in cauchy_kernels.hip).
//
// Layout: activations [B, H, L] fp32 with L contiguous, as in the reference, so
// every per-position op is coalesced along time and the channel reductions
// (TransposedLN) run down a column held by one lane.
#include "sashimi.h"

namespace dws {

__device__ __forceinline__ float gelu_erf(float x) { return dws_gelu(x); }
__device__ __forceinline__ float sigmoid_s(float x) { return dws_sigmoid(x); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cdiv(float2 a, float2 b) {
    // Smith's algorithm (what torch's complex division uses): robust for tiny |b|
    if (fabsf(b.x) >= fabsf(b.y)) {
        const float r = b.y / b.x, den = b.x + b.y * r;
        return make_float2((a.x + a.y * r) / den, (a.y - a.x * r) / den);
    }
    const float r = b.x / b.y, den = b.x * r + b.y;
    return make_float2((a.x * r + a.y) / den, (a.y * r - a.x) / den);
}

// ---------------------------------------------------------------------------
// S4 kernel generation (`SSKernelNPLR.forward`, s4.py:704-807) -- weight-time only
// ---------------------------------------------------------------------------

// v[(a*3+c)*H + h][n] = Bt[a][h][n] * Ct[c][h][n],  Bt = [B; P], Ct = [C0; C1; conj P]   (s4.py:748-752)
// wdt[h][n] = (-exp(inv_w_real) + i w_imag) * exp(log_dt[h])                              (s4.py:704-745)
__global__ void s4_prep_kernel(const float2* __restrict__ C, const float2* __restrict__ Bp, const float2* __restrict__ P,
                               const float* __restrict__ iwr, const float* __restrict__ wim,
                               const float* __restrict__ log_dt, float2* __restrict__ v, float2* __restrict__ wdt,
                               float* __restrict__ dt_out, int H, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * N) return;
    const int h = i / N;
    const float dt = expf(log_dt[h]);
    if (i % N == 0) dt_out[h] = dt;
    const float wr = -expf(iwr[i]);
    wdt[i] = make_float2(wr * dt, wim[i] * dt);
    const float2 b = Bp[i], p = P[i];
    const float2 ct[3] = {C[i], C[(size_t)H * N + i], make_float2(p.x, -p.y)};
    const float2 bt[2] = {b, p};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[(size_t)(a * 3 + c) * H * N + i] = cmul(bt[a], ct[c]);
}

int launch_s4_prep(const float* C, const float* Bp, const float* P, const float* iwr, const float* wim,
                   const float* log_dt, float* v, float* wdt, float* dt, int H, int N, hipStream_t s) {
    hipLaunchKernelGGL(s4_prep_kernel, dim3(ceil_div(H * N, 256)), dim3(256), 0, s, (const float2*)C,
                       (const float2*)Bp, (const float2*)P, iwr, wim, log_dt, (float2*)v, (float2*)wdt, dt, H, N);
    return DWS_OK;
}

// r *= dt; k_f = r00 - r01 r10 / (1 + r11); k_f = k_f * 2 / (1 + omega)      (s4.py:763-793)
// r is [2][3][H][Lh]; k_f out [2][H][Lh].  irfft ignores Im of the DC and (even n) Nyquist bins:
// they are zeroed so the C2R transform sees a Hermitian-consistent spectrum.
__global__ void s4_woodbury_kernel(const float2* __restrict__ r, const float2* __restrict__ omega,
                                   const float* __restrict__ dt, float2* __restrict__ kf, int H, int Lh, int n_even) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (l >= Lh) return;
    const float d = dt[h];
    auto R = [&](int a, int c) {
        const float2 x = r[((size_t)(a * 3 + c) * H + h) * Lh + l];
        return make_float2(x.x * d, x.y * d);
    };
    const float2 r01 = R(0, 2), r11 = R(1, 2);
    const float2 one_r11 = make_float2(1.f + r11.x, r11.y);
    const float2 om = omega[l];
    const float2 one_om = make_float2(1.f + om.x, om.y);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float2 r00 = R(0, c), r10 = R(1, c);
        const float2 q = cdiv(cmul(r01, r10), one_r11);
        float2 k = make_float2(r00.x - q.x, r00.y - q.y);
        k = cdiv(make_float2(k.x * 2.f, k.y * 2.f), one_om);
        if (l == 0 || (n_even && l == Lh - 1)) k.y = 0.f;
        kf[((size_t)c * H + h) * Lh + l] = k;
    }
}

int launch_s4_woodbury(const float* r, const float* omega, const float* dt, float* kf, int H, int Lh, int n_even,
                       hipStream_t s) {
    hipLaunchKernelGGL(s4_woodbury_kernel, dim3(ceil_div(Lh, 256), H), dim3(256), 0, s, (const float2*)r,
                       (const float2*)omega, dt, (float2*)kf, H, Lh, n_even);
    return DWS_OK;
}

// Two-sided kernel (s4.py:1391-1394): K[h][j] = k0[h][j]/L for j < L; K[h][L+i] = k1[h][L-1-i]/L.
// (1/L is the irfft normalisation rocFFT's unnormalised C2R leaves out.)
__global__ void s4_twosided_kernel(const float* __restrict__ k, float* __restrict__ K, int H, int L, int Lk, int Lt) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (j >= 2 * L) return;
    const float inv = 1.f / (float)Lk;   // k rows have length Lk; Lt = min(L, Lk) taps per direction (s4.py:1387)
    float v = 0.f;
    if (j < Lt) v = k[(size_t)h * Lk + j];
    else if (j >= 2 * L - Lt) v = k[((size_t)H + h) * Lk + (2 * L - 1 - j)];
    K[(size_t)h * 2 * L + j] = v * inv;
}

int launch_s4_twosided(const float* k, float* K, int H, int L, int Lk, int Lt, hipStream_t s) {
    hipLaunchKernelGGL(s4_twosided_kernel, dim3(ceil_div(2 * L, 256), H), dim3(256), 0, s, k, K, H, L, Lk, Lt);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// Per-step kernels
// ---------------------------------------------------------------------------

// TransposedLN (sashimi.py:17-20) down the channel column of each position:
//   y = (s_p / std) * (x - mean + m_p)  (+ part_t[b,h])      population std, no eps.
// Output row stride `ostride` lets the S4 input land directly in the zero-padded
// FFT buffer (rows of length 2L).
__global__ void ln_kernel(const float* __restrict__ x, const float* __restrict__ m_p, const float* __restrict__ s_p,
                          const float* __restrict__ part_t, int pt_bstride, float* __restrict__ out, int H, int L,
                          size_t ostride, const int* __restrict__ step_idx, int pt_tstride) {
    const int b = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    if (part_t) part_t += (size_t)b * pt_bstride + step_row_off(step_idx, pt_tstride);
    const float* xb = x + (size_t)b * H * L + l;
    float sum = 0.f;
    for (int h = 0; h < H; ++h) sum += xb[(size_t)h * L];
    const float mean = sum / (float)H;
    float var = 0.f;
    for (int h = 0; h < H; ++h) {
        const float d = xb[(size_t)h * L] - mean;
        var = fmaf(d, d, var);
    }
    const float scale = s_p[0] / sqrtf(var / (float)H);
    const float shift = m_p[0] - mean;
    float* ob = out + (size_t)b * H * ostride + l;
    for (int h = 0; h < H; ++h) {
        float y = scale * (xb[(size_t)h * L] + shift);
        if (part_t) y += part_t[h];
        ob[(size_t)h * ostride] = y;
    }
}

// Same op, one HBM pass: a block owns 64 positions x all H channels; thread (col, part) keeps
// its H/4 values of column `col` in registers (H <= 512), partial sums meet in LDS.
template <int RPT>   // rows (channels) held per thread: H <= 4 * RPT
__global__ __launch_bounds__(256) void ln_tile_kernel(const float* __restrict__ x, const float* __restrict__ m_p,
                                                      const float* __restrict__ s_p, const float* __restrict__ part_t,
                                                      int pt_bstride, float* __restrict__ out, int H, int L,
                                                      size_t ostride, const int* __restrict__ step_idx, int pt_tstride) {
    __shared__ float red[2][4][64];
    const int b = blockIdx.y, col = threadIdx.x & 63, part = threadIdx.x >> 6;
    if (part_t) part_t += (size_t)b * pt_bstride + step_row_off(step_idx, pt_tstride);
    const int l = blockIdx.x * 64 + col;
    const bool ok = l < L;
    const float* __restrict__ xb = x + (size_t)b * H * L + (ok ? l : 0);
    float v[RPT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int h = i * 4 + part;
        v[i] = (h < H) ? xb[(size_t)h * L] : 0.f;
        sum += v[i];
    }
    red[0][part][col] = sum;
    __syncthreads();
    const float mean = (red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col]) / (float)H;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int h = i * 4 + part;
        const float d = (h < H) ? v[i] - mean : 0.f;
        var = fmaf(d, d, var);
    }
    red[1][part][col] = var;
    __syncthreads();
    var = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
    const float scale = s_p[0] / sqrtf(var / (float)H);
    const float shift = m_p[0] - mean;
    float* __restrict__ ob = out + (size_t)b * H * ostride + l;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int h = i * 4 + part;
        if (h < H && ok) {
            float y = scale * (v[i] + shift);
            if (part_t) y += part_t[h];
            ob[(size_t)h * ostride] = y;
        }
    }
}

// init_conv (`sashimi.py:281`: weight-normed 1x1 conv + ReLU from Cin channels) and the FIRST block's S4 input
//   y = LN1(x) + fc_t(e)     (`sashimi.py:148-152`)
// in one pass: a thread owns one position, the D channel values are recomputed from the Cin input samples for the mean,
// the variance and the two outputs (Cin <= 4: cheaper than holding D values), so x_init is written once and never
// re-read by a LayerNorm launch.  W / bias reads are wave-uniform (scalar loads).
template <int CIN>
__global__ __launch_bounds__(256) void init_conv_ln_kernel(const float* __restrict__ audio, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ m_p,
                                                           const float* __restrict__ s_p, const float* __restrict__ part_t,
                                                           int pt_bstride, const int* __restrict__ step_idx, int pt_tstride,
                                                           float* __restrict__ x, float* __restrict__ y, int D, int L) {
    const int b = blockIdx.y, l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    part_t += (size_t)b * pt_bstride + step_row_off(step_idx, pt_tstride);
    float a[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) a[ci] = audio[((size_t)b * CIN + ci) * L + l];
    auto chan = [&](int c) {
        float acc = bias[c];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc = fmaf(W[c * CIN + ci], a[ci], acc);
        return dws_relu(acc);
    };
    float sum = 0.f;
    for (int c = 0; c < D; ++c) sum += chan(c);
    const float mean = sum / (float)D;
    float var = 0.f;
    for (int c = 0; c < D; ++c) {
        const float d = chan(c) - mean;
        var = fmaf(d, d, var);
    }
    const float scale = s_p[0] / sqrtf(var / (float)D), shift = m_p[0] - mean;
    float* __restrict__ xb = x + (size_t)b * D * L + l;
    float* __restrict__ yb = y + (size_t)b * D * L + l;
    for (int c = 0; c < D; ++c) {
        const float v = chan(c);
        xb[(size_t)c * L] = v;
        yb[(size_t)c * L] = scale * (v + shift) + part_t[c];
    }
}

bool init_conv_ln_supported(int Cin) { return Cin >= 1 && Cin <= 4; }

int launch_init_conv_ln(const float* audio, const float* W, const float* bias, const float* m_p, const float* s_p,
                        const float* part_t, int pt_bstride, const int* step_idx, int pt_tstride, float* x, float* y, int B,
                        int Cin, int D, int L, hipStream_t s) {
    ProfileScope ps("init_conv", s);
    dim3 grid(ceil_div(L, 256), B);
#define DWS_ICL(CI)                                                                                                     \
    hipLaunchKernelGGL(init_conv_ln_kernel<CI>, grid, dim3(256), 0, s, audio, W, bias, m_p, s_p, part_t, pt_bstride, step_idx, \
                       pt_tstride, x, y, D, L)
    switch (Cin) {
        case 1: DWS_ICL(1); break;
        case 2: DWS_ICL(2); break;
        case 3: DWS_ICL(3); break;
        case 4: DWS_ICL(4); break;
        default: return set_error(DWS_ERR_UNSUPPORTED, "init_conv_ln: Cin=%d", Cin);
    }
#undef DWS_ICL
    return DWS_OK;
}

int launch_ln(const float* x, const float* m_p, const float* s_p, const float* part_t, int pt_bstride, float* out,
              int B, int H, int L, size_t ostride, hipStream_t s, const int* step_idx, int pt_tstride) {
    ProfileScope ps("ln_kernel", s);
    if (H <= 128)
        hipLaunchKernelGGL(ln_tile_kernel<32>, dim3(ceil_div(L, 64), B), dim3(256), 0, s, x, m_p, s_p, part_t,
                           pt_bstride, out, H, L, ostride, step_idx, pt_tstride);
    else if (H <= 256)
        hipLaunchKernelGGL(ln_tile_kernel<64>, dim3(ceil_div(L, 64), B), dim3(256), 0, s, x, m_p, s_p, part_t,
                           pt_bstride, out, H, L, ostride, step_idx, pt_tstride);
    else if (H <= 512)
        hipLaunchKernelGGL(ln_tile_kernel<128>, dim3(ceil_div(L, 64), B), dim3(256), 0, s, x, m_p, s_p, part_t,
                           pt_bstride, out, H, L, ostride, step_idx, pt_tstride);
    else
        hipLaunchKernelGGL(ln_kernel, dim3(ceil_div(L, 64), B), dim3(64), 0, s, x, m_p, s_p, part_t, pt_bstride, out,
                           H, L, ostride, step_idx, pt_tstride);
    return DWS_OK;
}

// U_f[b,h,k] *= K_f[h,k]     (s4.py:1405: contract('bhl,chl->bchl'), c = 1)
__global__ void spec_mul_kernel(float2* __restrict__ uf, const float2* __restrict__ kf, int H, int Lf) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    if (k >= Lf) return;
    const size_t i = ((size_t)b * H + h) * Lf + k;
    uf[i] = cmul(uf[i], kf[(size_t)h * Lf + k]);
}

int launch_spec_mul(float* uf, const float* kf, int B, int H, int Lf, hipStream_t s) {
    ProfileScope ps("spec_mul", s);
    hipLaunchKernelGGL(spec_mul_kernel, dim3(ceil_div(Lf, 256), H, B), dim3(256), 0, s, (float2*)uf, (const float2*)kf,
                       H, Lf);
    return DWS_OK;
}

// g = GELU(y_conv / (2L) + D[h] * u)    (s4.py:1406-1430); y_conv and u are rows of length 2L
__global__ void s4_post_kernel(const float* __restrict__ yc, const float* __restrict__ u, const float* __restrict__ D,
                               float* __restrict__ g, int H, int L) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    const size_t row = (size_t)b * H + h;
    const float y = yc[row * 2 * L + l] * (1.f / (float)(2 * L)) + u[row * 2 * L + l] * D[h];
    g[row * L + l] = gelu_erf(y);
}

int launch_s4_post(const float* yc, const float* u, const float* D, float* g, int B, int H, int L, hipStream_t s) {
    ProfileScope ps("s4_post", s);
    hipLaunchKernelGGL(s4_post_kernel, dim3(ceil_div(L, 256), H, B), dim3(256), 0, s, yc, u, D, g, H, L);
    return DWS_OK;
}

// Generic 1x1 convolution with the index maps and epilogues the backbone needs.
//   acc[o] = bias[o] + sum_k W[o,k] * in(b,k,l)
// IN_POOL:  in(b, k = h*p + j, l) = x[b, h, l*p + j]                    (DownPool, sashimi.py:37)
// EPI 0: out[b,o,l] = act(acc)               act: 0 none, 1 GELU(erf), 2 ReLU
// EPI 1: GLU + residual: out[b,h,l] = res[b,h,l] + acc[h] * sigmoid(acc[O/2 + h]) (+ mel)   (s4.py:1435, sashimi.py:177)
// EPI 2: residual: out[b,o,l] = res[b,o,l] + acc (+ addend)                             (sashimi.py:182)
// EPI 3: UpPool scatter: out[b, o/p, l*p + o%p] = acc (+ addend)                        (sashimi.py:57)
struct PwArgs {
    const float* in; const float* W; const float* bias; float* out;
    const float* res; const float* addend; const float* mel;
    int mel_bstride;
    int B, K, O, L, p, act;
};

template <bool IN_POOL, int EPI>
__global__ void pw_generic_kernel(PwArgs a) {
    const int b = blockIdx.z, o = blockIdx.y;
    const int L = a.L, K = a.K;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int Hin = IN_POOL ? K / a.p : K;
    const int Lin = IN_POOL ? L * a.p : L;
    const float* inb = a.in + (size_t)b * Hin * Lin;
    auto dot = [&](int row) {
        const float* w = a.W + (size_t)row * K;
        float acc = 0.f;
        if (IN_POOL) {
            for (int k = 0; k < K; ++k) acc = fmaf(w[k], inb[(size_t)(k / a.p) * Lin + (size_t)l * a.p + (k % a.p)], acc);
        } else {
            for (int k = 0; k < K; ++k) acc = fmaf(w[k], inb[(size_t)k * L + l], acc);
        }
        return acc + a.bias[row];
    };
    if (EPI == 0) {
        float v = dot(o);
        if (a.act == 1) v = gelu_erf(v);
        else if (a.act == 2) v = dws_relu(v);
        a.out[((size_t)b * a.O + o) * L + l] = v;
    } else if (EPI == 1) {
        const int Hh = a.O / 2;
        const float va = dot(o), vb = dot(Hh + o);
        const size_t idx = ((size_t)b * Hh + o) * L + l;
        float v = va * sigmoid_s(vb);
        if (a.mel) v += a.mel[((size_t)(a.mel_bstride ? b : 0) * Hh + o) * L + l];
        a.out[idx] = a.res[idx] + v;
    } else if (EPI == 2) {
        const size_t idx = ((size_t)b * a.O + o) * L + l;
        float v = a.res[idx] + dot(o);
        if (a.addend) v += a.addend[idx];
        a.out[idx] = v;
    } else {
        const int p = a.p, Ho = a.O / p;
        const size_t idx = ((size_t)b * Ho + o / p) * ((size_t)L * p) + (size_t)l * p + (o % p);
        float v = dot(o);
        if (a.addend) v += a.addend[idx];
        a.out[idx] = v;
    }
}

template <bool IN_POOL, int EPI>
static void pw_launch(const PwArgs& a, int rows, hipStream_t s) {
    hipLaunchKernelGGL((pw_generic_kernel<IN_POOL, EPI>), dim3(ceil_div(a.L, 128), rows, a.B), dim3(128), 0, s, a);
}

int launch_pw_conv(const float* in, const float* W, const float* bias, float* out, int B, int K, int O, int L, int act,
                   hipStream_t s) {
    ProfileScope ps("pw_conv", s);
    PwArgs a{in, W, bias, out, nullptr, nullptr, nullptr, 0, B, K, O, L, 1, act};
    pw_launch<false, 0>(a, O, s);
    return DWS_OK;
}

int launch_pw_glu_res(const float* in, const float* W, const float* bias, const float* res, const float* mel,
                      int mel_bstride, float* out, int B, int H, int L, hipStream_t s) {
    ProfileScope ps("pw_glu_res", s);
    PwArgs a{in, W, bias, out, res, nullptr, mel, mel_bstride, B, H, 2 * H, L, 1, 0};
    pw_launch<false, 1>(a, H, s);
    return DWS_OK;
}

int launch_pw_res(const float* in, const float* W, const float* bias, const float* res, const float* addend,
                  float* out, int B, int K, int O, int L, hipStream_t s) {
    ProfileScope ps("pw_res", s);
    PwArgs a{in, W, bias, out, res, addend, nullptr, 0, B, K, O, L, 1, 0};
    pw_launch<false, 2>(a, O, s);
    return DWS_OK;
}

int launch_pw_downpool(const float* x, const float* W, const float* bias, float* out, int B, int Hin, int p, int O,
                       int Lout, hipStream_t s) {
    ProfileScope ps("pw_downpool", s);
    PwArgs a{x, W, bias, out, nullptr, nullptr, nullptr, 0, B, Hin * p, O, Lout, p, 0};
    pw_launch<true, 0>(a, O, s);
    return DWS_OK;
}

int launch_pw_uppool(const float* x, const float* W, const float* bias, const float* addend, float* out, int B,
                     int Hin, int p, int Hout, int Lin, hipStream_t s) {
    ProfileScope ps("pw_uppool", s);
    PwArgs a{x, W, bias, out, nullptr, addend, nullptr, 0, B, Hin, Hout * p, Lin, p, 0};
    pw_launch<false, 3>(a, Hout * p, s);
    return DWS_OK;
}

}  // namespace dws
