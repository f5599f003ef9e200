// Reverse-diffusion sampler (`generate.py:23-55`) as a replayed hipGraph.
//
// Per step t = T-1 .. 0 the reference does (`generate.py:50-54`):
//   eps = net((x, t));  x = (x - (1-a_t)/sqrt(1-abar_t) * eps) / sqrt(a_t);  if t > 0: x += sigma_t * z
// with t, the noise and (partly) the tables crossing the host/device boundary
// every step.  Here the step index lives in device memory, the coefficient
// tables are device resident, and z comes either from an injected tensor
// (parity mode) or from an on-device Philox4x32-10 counter RNG, so that one
// reverse step is a fixed kernel sequence: captured once, replayed T times.
#include <cmath>

#include "model.h"

namespace dws {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// four N(0,1) samples for element group g of stream `t`
__device__ __forceinline__ void normal4(uint64_t seed, uint32_t t, uint64_t g, float z[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), t, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float k = 2.3283064365386963e-10f;  // 2^-32
    const float u0 = ((float)r[0] + 0.5f) * k, u1 = ((float)r[1] + 0.5f) * k;
    const float u2 = ((float)r[2] + 0.5f) * k, u3 = ((float)r[3] + 0.5f) * k;
    const float ra = sqrtf(-2.f * logf(fminf(u0, 0.99999994f)));
    const float rb = sqrtf(-2.f * logf(fminf(u2, 0.99999994f)));
    float s, c;
    sincospif(2.f * u1, &s, &c);
    z[0] = ra * c; z[1] = ra * s;
    sincospif(2.f * u3, &s, &c);
    z[2] = rb * c; z[3] = rb * s;
}

// sampler state in device memory: [0] the step index t, [1] number of update blocks that have finished this step
__global__ void smp_set_step_kernel(int* t_dev, int t) { t_dev[0] = t; t_dev[1] = 0; }

__global__ void smp_fill_normal_kernel(float* __restrict__ x, size_t n, uint64_t seed, uint32_t stream_id) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g * 4 < n; g += (size_t)gridDim.x * blockDim.x) {
        float z[4];
        normal4(seed, stream_id, g, z);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (g * 4 + j < n) x[g * 4 + j] = z[j];
    }
}

// x <- (x - c1[t]*eps) / c2[t]  (+ sigma[t]*z if t > 0); products and sums are
// rounded separately to match the reference's op-by-op fp32 evaluation (`generate.py:52,54`): contraction is switched
// off for this kernel and the arithmetic written with plain operators -- the __f*_rn intrinsics are inline functions
// whose operations hipcc fused into FMAs after inlining (found by
// tests/test_sampler_gpu.py::test_step_table_sampler_equals_the_per_step_loop: 1 ulp on most elements once t > 0).
// The last block to finish moves the step index on (t <- t - 1): every block has read t by then, and the next kernel
// that reads it is stream-ordered behind this one -- no separate one-thread launch per step.
__global__ void smp_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                  const float* __restrict__ tables, int* __restrict__ t_dev,
                                  const float* __restrict__ noise, uint64_t seed, size_t n, int T) {
#pragma clang fp contract(off)
    const int t = __builtin_amdgcn_readfirstlane(*(volatile int*)t_dev);
    const float c1 = tables[t], c2 = tables[T + t], sg = tables[2 * T + t];
    const float* nz = noise ? noise + (size_t)t * n : nullptr;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g * 4 < n; g += (size_t)gridDim.x * blockDim.x) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (t > 0 && !nz) normal4(seed, (uint32_t)t, g, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = g * 4 + j;
            if (i >= n) break;
            // plain operators under `fp contract(off)`: the products, the difference, the quotient and the sum are each
            // rounded once (the __f*_rn intrinsics are inline functions compiled with contraction allowed; after inlining
            // the backend fuses them)
            const float p = c1 * eps[i];
            float v = (x[i] - p) / c2;
            if (t > 0) {
                const float q = sg * (nz ? nz[i] : z[j]);
                v = v + q;
            }
            x[i] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(reinterpret_cast<unsigned*>(t_dev + 1), 1u) == gridDim.x - 1) {
            t_dev[1] = 0;
            t_dev[0] = t - 1;
        }
    }
}

static int upload_tables(dws_model* m, const float* alpha, const float* alpha_bar, const float* sigma, int T,
                         hipStream_t s) {
    std::vector<float> h(3 * (size_t)T);
    for (int t = 0; t < T; ++t) {
        // fp32 scalar arithmetic in the reference's order (`generate.py:52`)
        const float one_m_a = 1.0f - alpha[t];
        const float den = sqrtf(1.0f - alpha_bar[t]);
        h[t] = one_m_a / den;
        h[T + t] = sqrtf(alpha[t]);
        h[2 * T + t] = sigma[t];
    }
    // the device copy is keyed on the table CONTENTS (same T with another beta schedule must not reuse it)
    if (m->smp_T == T && m->smp_host_tables == h) return DWS_OK;
    DWS_TRY(m->smp_tables.ensure(h.size() * 4));
    DWS_TRY(m->smp_state.ensure(8));
    DWS_HIP(hipMemcpyAsync(m->smp_tables.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, s));
    DWS_HIP(hipStreamSynchronize(s));
    m->smp_T = T;
    m->smp_host_tables.swap(h);
    return DWS_OK;
}

static int one_step(dws_model* m, float* x, const float* noise, uint64_t seed, int T, hipStream_t s) {
    const size_t n = (size_t)m->B * m->d.out_channels * m->L;
    int* t_dev = static_cast<int*>(m->smp_state.p);
    // every clip is at step t (`generate.py:50`): the network reads row t of the step table built at sampler entry
    m->step_idx = t_dev;
    const int st = m->forward(x, nullptr, m->smp_eps.f(), s);
    m->step_idx = nullptr;
    DWS_TRY(st);
    const int blocks = (int)std::min<size_t>(ceil_div(n, 4 * 256), 4096);
    hipLaunchKernelGGL(smp_update_kernel, dim3(blocks), dim3(256), 0, s, x, m->smp_eps.f(), m->smp_tables.f(), t_dev,
                       noise, seed, n, T);
    return DWS_OK;
}

static int run_steps(dws_model* m, float* x, int T, int t_start, int n_steps, const float* noise, uint64_t seed,
                     int use_graph, hipStream_t s) {
    DWS_CHECK(m->B > 0, DWS_ERR_STATE, "sampler before dws_model_prepare");
    DWS_CHECK(m->d.in_channels == m->d.out_channels, DWS_ERR_INVALID,
              "sampler needs in_channels == out_channels (x and eps share a shape, `generate.py:52`)");
    DWS_CHECK(t_start < T && n_steps >= 0 && t_start - n_steps >= -1, DWS_ERR_INVALID, "bad step range");
    if (m->dirty) DWS_TRY(m->commit(s));
    const size_t n = (size_t)m->B * m->d.out_channels * m->L;
    DWS_TRY(m->smp_eps.ensure(n * 4));
    m->smp_eps_B = m->B; m->smp_eps_L = m->L;
    DWS_TRY(m->build_step_table(T, s));   // step-only part of the network for t = 0..T-1 (kept while weights and T stay)
    int* t_dev = static_cast<int*>(m->smp_state.p);

    if (!use_graph) {
        hipLaunchKernelGGL(smp_set_step_kernel, dim3(1), dim3(1), 0, s, t_dev, t_start);
        for (int i = 0; i < n_steps; ++i) DWS_TRY(one_step(m, x, noise, seed, T, s));
        DWS_HIP(hipGetLastError());
        return DWS_OK;
    }
    // The caller's stream may be the legacy null stream, which cannot be
    // captured: capture and replay on an engine-owned stream that is ordered
    // after / before the caller's stream with events.
    if (!m->smp_stream) {
        DWS_HIP(hipStreamCreateWithFlags(&m->smp_stream, hipStreamNonBlocking));
        DWS_HIP(hipEventCreateWithFlags(&m->smp_ev_in, hipEventDisableTiming));
        DWS_HIP(hipEventCreateWithFlags(&m->smp_ev_out, hipEventDisableTiming));
    }
    hipStream_t cs = m->smp_stream;
    DWS_HIP(hipEventRecord(m->smp_ev_in, s));
    DWS_HIP(hipStreamWaitEvent(cs, m->smp_ev_in, 0));
    hipLaunchKernelGGL(smp_set_step_kernel, dim3(1), dim3(1), 0, cs, t_dev, t_start);
    const bool reuse = m->smp_graph && m->g_B == m->B && m->g_L == m->L && m->g_T == T && m->g_x == x &&
                       m->g_noise == noise && m->g_seed == seed;
    if (!reuse) {
        m->drop_graph();
        hipGraph_t graph = nullptr;
        DWS_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int st = one_step(m, x, noise, seed, T, cs);
        hipError_t e = hipStreamEndCapture(cs, &graph);
        if (st != DWS_OK) {
            if (graph) hipGraphDestroy(graph);
            return st;
        }
        DWS_HIP(e);
        e = hipGraphInstantiate(&m->smp_graph, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        DWS_HIP(e);
        m->g_B = m->B; m->g_L = m->L; m->g_T = T; m->g_x = x; m->g_noise = noise; m->g_seed = seed;
    }
    for (int i = 0; i < n_steps; ++i) DWS_HIP(hipGraphLaunch(m->smp_graph, cs));
    DWS_HIP(hipEventRecord(m->smp_ev_out, cs));
    DWS_HIP(hipStreamWaitEvent(s, m->smp_ev_out, 0));
    return DWS_OK;
}

int sampler_run(dws_model* m, float* x, const float* alpha, const float* alpha_bar, const float* sigma, int T,
                const float* noise, uint64_t seed, int init_from_seed, int use_graph, hipStream_t s) {
    DWS_CHECK(T > 0 && alpha && alpha_bar && sigma, DWS_ERR_INVALID, "sampler: bad schedule tables");
    DWS_TRY(upload_tables(m, alpha, alpha_bar, sigma, T, s));
    if (init_from_seed) {
        const size_t n = (size_t)m->B * m->d.in_channels * m->L;
        const int blocks = (int)std::min<size_t>(ceil_div(n, 4 * 256), 4096);
        hipLaunchKernelGGL(smp_fill_normal_kernel, dim3(blocks), dim3(256), 0, s, x, n, seed, (uint32_t)T);
    }
    return run_steps(m, x, T, T - 1, T, noise, seed, use_graph, s);
}

int sampler_steps(dws_model* m, float* x, const float* alpha, const float* alpha_bar, const float* sigma, int T,
                  int t_start, int n_steps, uint64_t seed, int use_graph, hipStream_t s) {
    DWS_CHECK(T > 0 && alpha && alpha_bar && sigma, DWS_ERR_INVALID, "sampler: bad schedule tables");
    // re-uploaded only when the coefficients differ from the resident ones (3T host flops per call)
    DWS_TRY(upload_tables(m, alpha, alpha_bar, sigma, T, s));
    return run_steps(m, x, T, t_start, n_steps, nullptr, seed, use_graph, s);
}

}  // namespace dws
