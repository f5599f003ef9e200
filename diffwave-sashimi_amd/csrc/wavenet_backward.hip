// WaveNet backward kernels (training path) -- first, correctness-first
// version on plain FMAs: every reduction is deterministic (no atomics), every kernel is the direct
// adjoint of the forward op it cites.  The MFMA versions (transposed dilated conv as a position-tile
// GEMM, weight gradients as split-N GEMMs) replace the heavy ones in a later round.
//
// Forward of one residual layer (`models/wavenet.py:82-121`), with folded weights:
//   h = x + pt[b,:,None] (zero padded);  H = Wd (*) h + b1;  g = tanh(H[:C]) * sigmoid(H[C:])
//   res = Wr g + br;  skip = Ws g + bs;  x' = (x + res) * sqrt(.5);  skip_sum += skip
#include "model.h"
#include "wavenet_backward.h"

namespace dws {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// db[o] = sum_{b,l} dY[b,o,l]
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ dY, float* __restrict__ db, int B, int O,
                                                     int L, float scale, int accumulate) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* r = dY + ((size_t)b * O + o) * L;
        for (int l = threadIdx.x; l < L; l += 256) s += r[l];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) db[o] = (accumulate ? db[o] : 0.f) + s * scale;
}

int launch_rowsum(const float* dY, float* db, int B, int O, int L, float scale, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(rowsum_kernel, dim3(O), dim3(256), 0, s, dY, db, B, O, L, scale, accumulate);
    return DWS_OK;
}

// dW[o][c][t] = scale * sum_{b,l} dY[b,o,l] * Xh[b,c,l+(t-1)d],  Xh = X (+ addc[b,c]) inside [0,L), 0 outside.
// One block per (o-tile of TO, c-tile of TC); threads stride over (b,l); fixed-order block reduction.
template <int TO, int TC, int T>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                    const float* __restrict__ addc, int addc_bstride,
                                                    float* __restrict__ dW, int B, int O, int C, int L, int d,
                                                    float scale) {
    __shared__ float red[4];
    const int o0 = blockIdx.x * TO, c0 = blockIdx.y * TC;
    float acc[TO][TC][T];
#pragma unroll
    for (int i = 0; i < TO; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[i][j][t] = 0.f;
    for (int b = 0; b < B; ++b) {
        for (int l = threadIdx.x; l < L; l += 256) {
            float dy[TO];
#pragma unroll
            for (int i = 0; i < TO; ++i) dy[i] = (o0 + i < O) ? dY[((size_t)b * O + o0 + i) * L + l] : 0.f;
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                if (c0 + j >= C) continue;
                const float ad = addc ? addc[(size_t)b * addc_bstride + c0 + j] : 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int pos = l + (t - (T / 2)) * d;
                    const float xv = ((unsigned)pos < (unsigned)L) ? X[((size_t)b * C + c0 + j) * L + pos] + ad : 0.f;
#pragma unroll
                    for (int i = 0; i < TO; ++i) acc[i][j][t] = fmaf(dy[i], xv, acc[i][j][t]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TO; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float v = block_sum_256(acc[i][j][t], red);
                if (threadIdx.x == 0 && o0 + i < O && c0 + j < C) dW[((size_t)(o0 + i) * C + c0 + j) * T + t] = v * scale;
            }
}

int launch_wgrad(const float* dY, const float* X, const float* addc, int addc_bstride, float* dW, int B, int O, int C,
                 int L, int taps, int d, float scale, hipStream_t s) {
    DWS_CHECK(taps == 1 || taps == 3, DWS_ERR_UNSUPPORTED, "wgrad: taps=%d", taps);
    dim3 grid(ceil_div(O, 4), ceil_div(C, 4));
    if (taps == 3)
        hipLaunchKernelGGL((wgrad_kernel<4, 4, 3>), grid, dim3(256), 0, s, dY, X, addc, addc_bstride, dW, B, O, C, L, d, scale);
    else
        hipLaunchKernelGGL((wgrad_kernel<4, 4, 1>), grid, dim3(256), 0, s, dY, X, addc, addc_bstride, dW, B, O, C, L, d, scale);
    return DWS_OK;
}

// Adjoint of a (dilated) conv w.r.t. its input: dX[b,c,l] (+)= scale * sum_{o,t} W[o,c,t] * dY[b,o,l-(t-1)d]
template <int T>
__global__ void conv_t_kernel(const float* __restrict__ dY, const float* __restrict__ W, float* __restrict__ dX, int O,
                              int C, int L, int d, float scale, int accumulate) {
    const int b = blockIdx.z, c = blockIdx.y;
    const float* dyb = dY + (size_t)b * O * L;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int o = 0; o < O; ++o) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int pos = l - (t - (T / 2)) * d;
                if ((unsigned)pos < (unsigned)L) acc = fmaf(W[((size_t)o * C + c) * T + t], dyb[(size_t)o * L + pos], acc);
            }
        }
        const size_t idx = ((size_t)b * C + c) * L + l;
        dX[idx] = (accumulate ? dX[idx] : 0.f) + acc * scale;
    }
}

int launch_conv_t(const float* dY, const float* W, float* dX, int B, int O, int C, int L, int taps, int d, float scale,
                  int accumulate, hipStream_t s) {
    dim3 grid(min(ceil_div(L, 128), 512), C, B);
    if (taps == 3)
        hipLaunchKernelGGL(conv_t_kernel<3>, grid, dim3(128), 0, s, dY, W, dX, O, C, L, d, scale, accumulate);
    else
        hipLaunchKernelGGL(conv_t_kernel<1>, grid, dim3(128), 0, s, dY, W, dX, O, C, L, d, scale, accumulate);
    return DWS_OK;
}

// Gate adjoint: g = tanh(Ht) sig(Hs);  dHt = dg sig(Hs) (1 - tanh^2 Ht);  dHs = dg tanh(Ht) sig(Hs)(1 - sig(Hs))
__global__ void gate_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ H, float* __restrict__ dH,
                                float* __restrict__ g, int C, int L) {
    const int b = blockIdx.z, c = blockIdx.y;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        const size_t it = ((size_t)b * 2 * C + c) * L + l, is = ((size_t)b * 2 * C + C + c) * L + l;
        const float th = tanhf(H[it]), sg = sigm(H[is]);
        const float d = dg[((size_t)b * C + c) * L + l];
        g[((size_t)b * C + c) * L + l] = th * sg;
        dH[it] = d * sg * (1.f - th * th);
        dH[is] = d * th * sg * (1.f - sg);
    }
}

int launch_gate_bwd(const float* dg, const float* H, float* dH, float* g, int B, int C, int L, hipStream_t s) {
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(min(ceil_div(L, 256), 256), C, B), dim3(256), 0, s, dg, H, dH, g, C, L);
    return DWS_OK;
}

// dy[b,r,l] = (y > 0) * sum_oc Wz[oc,r] * dout[b,oc,l]      (final ReLU + zero conv, `wavenet.py:198-200`)
__global__ void final_dy_kernel(const float* __restrict__ dout, const float* __restrict__ Wz,
                                const float* __restrict__ y, float* __restrict__ dy, int S, int Cout, int L) {
    const int b = blockIdx.z, r = blockIdx.y;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        const size_t idx = ((size_t)b * S + r) * L + l;
        float acc = 0.f;
        for (int oc = 0; oc < Cout; ++oc) acc = fmaf(Wz[oc * S + r], dout[((size_t)b * Cout + oc) * L + l], acc);
        dy[idx] = y[idx] > 0.f ? acc : 0.f;
    }
}

int launch_final_dy(const float* dout, const float* Wz, const float* y, float* dy, int B, int S, int Cout, int L,
                    hipStream_t s) {
    hipLaunchKernelGGL(final_dy_kernel, dim3(min(ceil_div(L, 256), 256), S, B), dim3(256), 0, s, dout, Wz, y, dy, S, Cout, L);
    return DWS_OK;
}

// dx_in = dx_out * sqrt(.5) + dh   (dx_out may be null: last layer), in place into dh
__global__ void dx_combine_kernel(float* __restrict__ dh, const float* __restrict__ dx_out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dh[i] += dx_out[i] * 0.70710678118654752440f;
}

int launch_dx_combine(float* dh, const float* dx_out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(dx_combine_kernel, dim3((unsigned)std::min<size_t>(ceil_div(n, 256), 4096)), dim3(256), 0, s, dh, dx_out, n);
    return DWS_OK;
}

// dres = dx_out * sqrt(.5)
__global__ void scale_kernel(const float* __restrict__ in, float* __restrict__ out, float a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] * a;
}

int launch_scale(const float* in, float* out, float a, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)std::min<size_t>(ceil_div(n, 256), 4096)), dim3(256), 0, s, in, out, a, n);
    return DWS_OK;
}

// dpt[b,c] = sum_l dh[b,c,l]
__global__ __launch_bounds__(256) void rowsum_bc_kernel(const float* __restrict__ dh, float* __restrict__ out,
                                                        int out_bstride, int C, int L) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* r = dh + ((size_t)b * C + c) * L;
    float s = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) s += r[l];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[(size_t)b * out_bstride + c] = s;
}

int launch_rowsum_bc(const float* dh, float* out, int out_bstride, int B, int C, int L, hipStream_t s) {
    hipLaunchKernelGGL(rowsum_bc_kernel, dim3(C, B), dim3(256), 0, s, dh, out, out_bstride, C, L);
    return DWS_OK;
}

// init conv adjoint: dpre = dx0 * (x0 > 0), in place
__global__ void relu_bwd_kernel(float* __restrict__ dx, const float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dx[i] : 0.f;
}

int launch_relu_bwd(float* dx, const float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)std::min<size_t>(ceil_div(n, 256), 4096)), dim3(256), 0, s, dx, y, n);
    return DWS_OK;
}

// weight-norm adjoint (W = g v / ||v|| per output row, `wavenet.py:21`):
//   vhat = v / ||v||;  dg = <dW, vhat>;  dv = (g / ||v||) (dW - dg vhat)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dW, const float* __restrict__ v,
                                                              const float* __restrict__ g, float* __restrict__ dv,
                                                              float* __restrict__ dg, int inner) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const float* vr = v + (size_t)o * inner;
    const float* dr = dW + (size_t)o * inner;
    float nn = 0.f, dot = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) {
        nn = fmaf(vr[i], vr[i], nn);
        dot = fmaf(dr[i], vr[i], dot);
    }
    nn = block_sum_256(nn, red);
    dot = block_sum_256(dot, red);
    const float norm = sqrtf(nn), dgo = dot / norm, sc = g[o] / norm;
    if (threadIdx.x == 0) dg[o] = dgo;
    for (int i = threadIdx.x; i < inner; i += 256) dv[(size_t)o * inner + i] = sc * (dr[i] - dgo * vr[i] / norm);
}

int launch_weight_norm_bwd(const float* dW, const float* v, const float* g, float* dv, float* dg, int O, int inner,
                           hipStream_t s) {
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(O), dim3(256), 0, s, dW, v, g, dv, dg, inner);
    return DWS_OK;
}

// ---- small dense layers (embedding MLP, `wavenet.py:153-155,89`) ----
// dW[o][k] = sum_b dy[b,o] x[b,k];  db[o] = sum_b dy[b,o]
__global__ void lin_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dW,
                                 float* __restrict__ db, int B, int K, int O) {
    const int o = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(dy[(size_t)b * O + o], x[(size_t)b * K + k], acc);
        dW[(size_t)o * K + k] = acc;
    }
    if (threadIdx.x == 0 && db) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dy[(size_t)b * O + o];
        db[o] = acc;
    }
}

int launch_lin_bwd_w(const float* dy, const float* x, float* dW, float* db, int B, int K, int O, hipStream_t s) {
    hipLaunchKernelGGL(lin_bwd_w_kernel, dim3(O), dim3(128), 0, s, dy, x, dW, db, B, K, O);
    return DWS_OK;
}

// dx[b,k] = sum_o W[o,k] dy[b,o], then optionally through swish: dx *= swish'(pre[b,k]).
// O (up to n_layers * C = 9216 stacked fc_t rows) is split over grid.z; partial sums are added in a fixed order.
__global__ void lin_bwd_x_partial_kernel(const float* __restrict__ dy, const float* __restrict__ W, float* __restrict__ part,
                                         int K, int O, int ochunk) {
    const int b = blockIdx.y, z = blockIdx.z;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int o0 = z * ochunk, o1 = min(O, o0 + ochunk);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int o = o0;
    for (; o + 4 <= o1; o += 4) {
        a0 = fmaf(W[(size_t)o * K + k], dy[(size_t)b * O + o], a0);
        a1 = fmaf(W[(size_t)(o + 1) * K + k], dy[(size_t)b * O + o + 1], a1);
        a2 = fmaf(W[(size_t)(o + 2) * K + k], dy[(size_t)b * O + o + 2], a2);
        a3 = fmaf(W[(size_t)(o + 3) * K + k], dy[(size_t)b * O + o + 3], a3);
    }
    for (; o < o1; ++o) a0 = fmaf(W[(size_t)o * K + k], dy[(size_t)b * O + o], a0);
    part[((size_t)z * gridDim.y + b) * K + k] = (a0 + a1) + (a2 + a3);
}

__global__ void lin_bwd_x_finish_kernel(const float* __restrict__ part, const float* __restrict__ pre, float* __restrict__ dx,
                                        int n, int nz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int z = 0; z < nz; ++z) acc += part[(size_t)z * n + i];
    if (pre) {
        const float a = pre[i], sg = sigm(a);
        acc *= sg * (1.f + a * (1.f - sg));   // d/da [a sigmoid(a)]
    }
    dx[i] = acc;
}

int launch_lin_bwd_x(const float* dy, const float* W, const float* pre, float* dx, int B, int K, int O, DevBuf& part,
                     hipStream_t s) {   // part: [nz][B][K] scratch owned by the model
    const int ochunk = 128, nz = ceil_div(O, ochunk);
    DWS_TRY(part.ensure((size_t)nz * B * K * 4));
    hipLaunchKernelGGL(lin_bwd_x_partial_kernel, dim3(ceil_div(K, 128), B, nz), dim3(128), 0, s, dy, W, part.f(), K, O, ochunk);
    hipLaunchKernelGGL(lin_bwd_x_finish_kernel, dim3(ceil_div(B * K, 256)), dim3(256), 0, s, part.f(), pre, dx, B * K, nz);
    return DWS_OK;
}

}  // namespace dws
