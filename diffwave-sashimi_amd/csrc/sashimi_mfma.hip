// MFMA (exact-f32 v_mfma_f32_32x32x2_f32) kernels of the SaShiMi per-position path.
//
// s4_tail_mfma_kernel fuses everything of DiffWaveBlock.forward after the S4
// convolution (`sashimi.py:177-184`, `s4.py:1435`) for a tile of P positions:
//   o  = Wo g + bo                      (H -> 2H)          GEMM-o
//   x1 = x + o[:H] * sigmoid(o[H:]) (+ mel)                GLU + residual   -> LDS
//   LN2 statistics down each column; tile centred in place (x1 - mean)
//   u  = GELU(W1 LN2(x1) + b1)          (H -> ff*H)        GEMM-1, LN folded into the epilogue:
//        W1 (a (xc + m_p)) = a (W1 xc) + a m_p rowsum(W1)
//   f  = W2 u + b2                      (ff*H -> H)        GEMM-2, accumulated per H-row chunk of u
//   out = x1 + f (+ addend)
// so g, x are read once and out written once (3 tensor passes instead of ~15).
//
// pw_mfma_kernel: the pooling 1x1 convs (`sashimi.py:23-58`) with the index
// maps folded into the B-operand gather (DownPool) / the float4 scatter (UpPool).
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "sashimi.h"
#include "sashimi_mfma.h"
#include "bf16_split.h"
#include "wavenet.h"

namespace dws {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_erf_m(float x) { return dws_gelu(x); }
__device__ __forceinline__ float sigmoid_m(float x) { return dws_sigmoid(x); }

__device__ __forceinline__ float f4_get(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

template <int H, int WM, int WN, int NT, int FFE>
struct TailCfg {
    static constexpr int P = 32 * WN * NT;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int MT = H / 32 / WM;            // 32-row tiles per wave for an H-row GEMM
    static constexpr int PARTS = THREADS / P;         // row partitions of the LN column reduction
    static constexpr int NCONST = (3 + 2 * FFE) * H;  // per-row constants kept in LDS: bo [2H], rs1 [FFE H], b1 [FFE H], b2 [H]
    static constexpr int LDS_FLOATS = 2 * H * P + 2 * P + 2 * PARTS * P + NCONST;
    static_assert(H % (32 * WM) == 0, "H split");
    static_assert(THREADS % P == 0 && H % PARTS == 0, "LN split");
};

// One H-row x K GEMM slab on the wave's MT tiles: acc[m][n] += A[tile rows] . Bt[K x P]
// A packed as pack[mt][kg][lane][4]; Bt in LDS row-major [K][P].
// KGU: leave the k-group loop to hipcc's unroller (nkg is a constant at every call site: it unrolls fully and hoists
// ring loads, +40..100 VGPRs but a little faster where the registers are there); !KGU pins it rolled (`unroll 1`).
// (Round 4, measured and dropped: requesting the first ring entries of the NEXT slab before the epilogue that separates two
// slabs -- a slab that fills its own ring starts with an L2 round trip in front of its first MFMA, six times per tile.  Same box:
// H = 128 126.05 -> 126.1 us, H = 256 116.3 -> 117.2 us.  The partner workgroup's MFMAs already cover those waits.)
template <int MT, int NT, int P, bool KGU>
__device__ __forceinline__ void gemm_slab(f32x16 (&acc)[MT][NT], const float4* __restrict__ A, int nkg_total,
                                          int kg0, int nkg, const int (&mt)[MT], const float* __restrict__ bt, int wn,
                                          int lane) {
    // A fragments come from L2 (~1 us away) and one k-group feeds only 4*MT*NT MFMAs, so they are fetched through a
    // ring of D k-groups; the sched_barrier pins each refill where it is written (hipcc otherwise sinks the load
    // next to its use and the wave waits out the full latency every k-group: measured 43 % MFMA busy)
    constexpr int D = 4;   // nkg is a multiple of 4 (K % 32 == 0)
    const int l31 = lane & 31, lhi = lane >> 5;
    float4 buf[D][MT];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int m = 0; m < MT; ++m) buf[d][m] = A[((size_t)mt[m] * nkg_total + kg0 + d) * 64 + lane];
#define DWS_TAIL_GROUP4                                                                                               \
    _Pragma("unroll") for (int d = 0; d < D; ++d) {                                                                  \
        float4 cur[MT];                                                                                              \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) cur[m] = buf[d][m];                                           \
        const int kn = min(kg + d + D, nkg - 1);                                                                     \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                               \
            buf[d][m] = A[((size_t)mt[m] * nkg_total + kg0 + kn) * 64 + lane];                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
            const int krow = (kg + d) * 8 + j * 2 + lhi;                                                             \
            float bf[NT];                                                                                            \
            _Pragma("unroll") for (int n = 0; n < NT; ++n) bf[n] = bt[krow * P + (wn * NT + n) * 32 + l31];          \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                           \
                _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                       \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(cur[m], j), bf[n], acc[m][n], 0, 0, 0);  \
        }                                                                                                            \
    }
    // (Reading the B fragments one k-step ahead of their MFMAs instead of directly in front of them changed nothing:
    // C3 6.30 vs 6.33 ms, C4 5.35 vs 5.32 ms per step on the same box -- the partner waves already cover that latency.)
    if constexpr (KGU) {
        for (int kg = 0; kg < nkg; kg += D) { DWS_TAIL_GROUP4 }
    } else {
#pragma unroll 1
        for (int kg = 0; kg < nkg; kg += D) { DWS_TAIL_GROUP4 }
    }
#undef DWS_TAIL_GROUP4
}

// The same slab on the 16-bit matrix cores (precision = "bf16x6" / "f16x3", H >= 256: bf16_split.h): a k-block of 16 is
// two fp32 k-groups of the SAME packed A (slot e = 4 g + j of the block is k = 16 kb + 8 g + 2 j + lhi, the order the two
// float4 fragments already hold per lane), split by the wave that owns the rows; the B operand is read from the fp32 tile
// in that order (eight ds_read_b32 per column tile) and split in registers -- every wave of the workgroup splits the
// columns it multiplies.  Scaled splits: B enters times SX, A times ws; the caller multiplies the accumulators back.
template <typename SP, int MT, int NT, int P>
__device__ __forceinline__ void gemm_slab_split(f32x16 (&acc)[MT][NT], const float4* __restrict__ A, int nkg_total, int kg0,
                                                int nkg, const int (&mt)[MT], const float* __restrict__ bt, int wn, int lane,
                                                float ws) {
    using v8 = typename SP::v8;
    constexpr int TN = SP::NT, NPR = SP::NP, D = 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nkb = nkg / 2;            // nkg is a multiple of 4
    float4 buf[D][MT][2];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 2; ++g) buf[d][m][g] = A[((size_t)mt[m] * nkg_total + kg0 + 2 * d + g) * 64 + lane];
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float4 cur[MT][2];
#pragma unroll
            for (int m = 0; m < MT; ++m) { cur[m][0] = buf[d][m][0]; cur[m][1] = buf[d][m][1]; }
            const int kn = min(kb + d + D, nkb - 1);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 2; ++g) buf[d][m][g] = A[((size_t)mt[m] * nkg_total + kg0 + 2 * kn + g) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            const float* bk = bt + ((kb + d) * 16 + lhi) * P + wn * NT * 32 + l31;
            v8 bq[NT][TN], af[MT][TN];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = bk[(8 * (e >> 2) + 2 * (e & 3)) * P + n * 32];
                    SP::split1(SP::SCALED ? v * SP::SX : v, bq[n], e);
                }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = f4_get(cur[m][e >> 2], e & 3);
                    SP::split1(SP::SCALED ? v * ws : v, af[m], e);
                }
#pragma unroll
            for (int t = 0; t < NPR; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = SP::mfma(af[m][SP::ia(t)], bq[n][SP::ib(t)], acc[m][n]);
        }
    }
}

// which slab a kernel instance runs: SP = void -> the exact-f32 one
template <typename SP, int MT, int NT, int P, bool KGU>
struct SlabRun {
    static constexpr bool SCALED = SP::SCALED;
    static constexpr float SX = SP::SX;
    __device__ static __forceinline__ void run(f32x16 (&acc)[MT][NT], const float4* __restrict__ A, int nkg_total, int kg0, int nkg,
                                               const int (&mt)[MT], const float* __restrict__ bt, int wn, int lane, float ws) {
        gemm_slab_split<SP, MT, NT, P>(acc, A, nkg_total, kg0, nkg, mt, bt, wn, lane, ws);
    }
};
template <int MT, int NT, int P, bool KGU>
struct SlabRun<void, MT, NT, P, KGU> {
    static constexpr bool SCALED = false;
    static constexpr float SX = 1.f;
    __device__ static __forceinline__ void run(f32x16 (&acc)[MT][NT], const float4* __restrict__ A, int nkg_total, int kg0, int nkg,
                                               const int (&mt)[MT], const float* __restrict__ bt, int wn, int lane, float) {
        gemm_slab<MT, NT, P, KGU>(acc, A, nkg_total, kg0, nkg, mt, bt, wn, lane);
    }
};

template <int MT, int NT>
__device__ __forceinline__ void acc_scale(f32x16 (&acc)[MT][NT], float f) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] *= f;
}

// TransposedLN statistics of an [H][P] tile in LDS, down each column (population std, no eps; `sashimi.py:17-20`):
// the tile is centred in place, colmean[col] = mean, colalpha[col] = s / std.  THREADS = P * PARTS; ends with a barrier.
template <int H, int P, int PARTS>
__device__ __forceinline__ void column_stats(float* __restrict__ tile, float* __restrict__ red, float* __restrict__ colmean,
                                             float* __restrict__ colalpha, float s_scale, int tid) {
    const int col = tid % P, part = tid / P;
    constexpr int RP = H / PARTS;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RP; ++r) s += tile[(part * RP + r) * P + col];
    red[part * P + col] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) mean += red[q * P + col];
    mean *= (1.f / (float)H);
    float v = 0.f;
#pragma unroll 8
    for (int r = 0; r < RP; ++r) {
        const float d = tile[(part * RP + r) * P + col] - mean;
        tile[(part * RP + r) * P + col] = d;
        v = fmaf(d, d, v);
    }
    red[(PARTS + part) * P + col] = v;
    __syncthreads();
    if (part == 0) {
        float var = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) var += red[(PARTS + q) * P + col];
        colmean[col] = mean;
        colalpha[col] = s_scale / sqrtf(var * (1.f / (float)H));
    }
    __syncthreads();
}

// two waves per SIMD at least: without the bound hipcc spends > 256 registers per lane on one resident workgroup
//
// VEC (L % 4 == 0, chosen at launch): every tile-sized stream moves 16 bytes per lane -- g and the block input x are
// staged by dwordx4 LDS-DMA (x into the still unused u buffer), the output and the next block's S4 input leave through
// a row-major float4 pass over the tile in LDS.  In the accumulator layout (a lane owns single positions of 16 rows)
// the same traffic takes four times the VMEM instructions, and their issue was the larger part of the kernel's
// non-MFMA time.
// SP: void = exact-f32 MFMA; SplitBf16x3 / SplitF16x2 = the GEMMs on the 16-bit matrix cores (gemm_slab_split).
template <int H, int WM, int WN, int NT, int FFE, bool VEC, int OCC, bool KGU, typename SP = void>
__global__ __launch_bounds__(64 * WM * WN, OCC) void s4_tail_mfma_kernel(S4TailArgs a) {
    using SR = SlabRun<SP, TailCfg<H, WM, WN, NT, FFE>::MT, NT, TailCfg<H, WM, WN, NT, FFE>::P, KGU>;
    constexpr bool SCALED = SR::SCALED;
    auto slab = [&](auto& acc, const float4* A, int nkg_total, int kg0, int nkg, const auto& mts, const float* bt, int wn_, int lane_,
                    float ws) { SR::run(acc, A, nkg_total, kg0, nkg, mts, bt, wn_, lane_, ws); };
    const float sx = SR::SX;
    const float wso = SCALED ? a.wscale_c6[0] : 1.f, ws1 = SCALED ? a.wscale_c6[1] : 1.f, ws2 = SCALED ? a.wscale_c6[2] : 1.f;
    using T = TailCfg<H, WM, WN, NT, FFE>;
    constexpr int P = T::P, MT = T::MT, THREADS = T::THREADS, PARTS = T::PARTS;
    // row-major float4 view of an [H][P] tile: F4_ROW float4 per row, ROWS_PASS rows per pass of the workgroup
    constexpr int F4_ROW = P / 4, ROWS_PASS = THREADS / F4_ROW, NPASS = H / ROWS_PASS;
    static_assert(THREADS % F4_ROW == 0 && H % ROWS_PASS == 0, "row-major passes");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds;                 // [H][P]: g, then x1 centred
    float* ut = lds + H * P;           // [H][P]: one H-row chunk of u
    float* colmean = ut + H * P;       // [P]
    float* colalpha = colmean + P;     // [P]
    float* red = colalpha + P;         // [2][PARTS][P]
    // per-row constants of the four epilogues (biases, row sums of W1): copied into LDS while the tiles are staged.  Fetched
    // where they are used they cost one exposed L2 round trip per epilogue and tile (`profiles/r04_tail_phase_trace.txt`:
    // the GLU / GELU / output phases carried ~2 k cycles of waiting each); a ds_read_b128 serves four consecutive rows.
    float* cst_bo = red + 2 * PARTS * P;   // [2H]
    float* cst_rs = cst_bo + 2 * H;        // [FFE H]
    float* cst_b1 = cst_rs + FFE * H;      // [FFE H]
    float* cst_b2 = cst_b1 + FFE * H;      // [H]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L;
    const int ntl = (L + P - 1) / P;
    const int tix = xcd_remap(blockIdx.x, gridDim.x);
    // readfirstlane: keeps every buffer descriptor below provably wave-uniform (no waterfall loops)
    const int b = __builtin_amdgcn_readfirstlane(tix / ntl), l0 = __builtin_amdgcn_readfirstlane((tix % ntl) * P);
    const int L4 = L * 4;
    constexpr int OOB = 0x7ffffff0;   // lane offset past every descriptor: loads give 0, stores are dropped
    unsigned long long* __restrict__ trc = a.trace ? a.trace + ((size_t)blockIdx.x * (THREADS / 64) + wave) * 16 : nullptr;
    auto stamp = [&](int i) {
        if (trc) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) trc[i] = t;
        }
    };
    stamp(0);

    // fp32 MFMA and VALU do not co-issue on a SIMD (tools/ubench/coexec.hip), so every VALU instruction in this
    // kernel is MFMA time lost: all global traffic goes through buffer instructions whose row part is a scalar
    // offset (no per-lane 64-bit address math), and the g tile is staged by LDS-DMA (no VALU, no VGPRs).
    // ---- stage g tile: row r of the tile = P/64 DMA instructions of 64 positions
    if constexpr (VEC) {
        // one dwordx4 DMA instruction fills 256 consecutive floats of the tile = 256/P whole rows
        constexpr int RPI = 256 / P, NI = H / RPI / (THREADS / 64);
        static_assert(H % (RPI * (THREADS / 64)) == 0, "DMA split");
        __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rXs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * H * L), 0, H * L4, 0x00020000);
        const int pos = l0 + 4 * (lane % F4_ROW);
        const int voff = pos < L ? ((lane / F4_ROW) * L + pos) * 4 : OOB;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row0 = (wave + (THREADS / 64) * i) * RPI;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, tile + row0 * P, 16, voff, row0 * L4, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row0 = (wave + (THREADS / 64) * i) * RPI;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rXs, ut + row0 * P, 16, voff, row0 * L4, 0, 0);
        }
    } else {
        __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + (size_t)b * H * L), 0, H * L4, 0x00020000);
        if constexpr (P >= 64) {
            constexpr int SEG = P / 64, NI = H * SEG / (THREADS / 64);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int q = wave + (THREADS / 64) * i;     // (row, segment) index
                const int row = q / SEG, seg = q % SEG;
                const int pos = l0 + seg * 64 + lane;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, tile + row * P + seg * 64, 4, pos < L ? pos * 4 : OOB, row * L4, 0, 0);
            }
        } else {   // P == 32: one instruction moves two consecutive rows (they are contiguous in the tile)
            constexpr int NI = H / 2 / (THREADS / 64);
            const int pos = l0 + l31;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int q = wave + (THREADS / 64) * i;     // row pair
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, tile + 2 * q * P, 4, pos < L ? (lhi * L + pos) * 4 : OOB, 2 * q * L4, 0, 0);
            }
        }
    }
    // ---- GEMM-o: rows [tile m] pair with rows [H/32 + tile m] (GLU halves)
    int mt_a[MT], mt_b[MT], mt_h[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        mt_h[m] = wm * MT + m;
        mt_a[m] = mt_h[m];
        mt_b[m] = H / 32 + mt_h[m];
    }
    for (int i = tid; i < T::NCONST; i += THREADS) {
        const float* src = i < 2 * H ? a.bo + i : i < (2 + FFE) * H ? a.rs1 + (i - 2 * H)
                         : i < (2 + 2 * FFE) * H ? a.b1 + (i - (2 + FFE) * H) : a.b2 + (i - (2 + 2 * FFE) * H);
        cst_bo[i] = *src;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
    __syncthreads();
    stamp(1);    // staging (LDS-DMA round trip + barrier)

    const float4* Ao = reinterpret_cast<const float4*>(a.Ao);
    {
        f32x16 acc_a[MT][NT], acc_b[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc_a[m][n][r] = 0.f; acc_b[m][n][r] = 0.f; }
        slab(acc_a, Ao, H / 8, 0, H / 8, mt_a, tile, wn, lane, wso);
        slab(acc_b, Ao, H / 8, 0, H / 8, mt_b, tile, wn, lane, wso);
        if constexpr (SCALED) {
            acc_scale(acc_a, 1.f / (wso * sx));
            acc_scale(acc_b, 1.f / (wso * sx));
        }
        stamp(2);    // GEMM-o
        __syncthreads();  // every wave is done reading g
        stamp(3);    // barrier
        // x1 = x + GLU(o) (+ mel) -> tile.  (Requesting x together with the g tile, so that its HBM round trip overlaps the
        // staging barrier, was measured on the same box: 166.5 vs 165.1 us at H = 64 -- no gain, not kept.)
        __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.mel ? a.mel + (size_t)(a.mel_bstride ? b : 0) * H * L : a.x), 0, H * L4, 0x00020000);
        const bool has_mel = a.mel != nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float ba[16], bb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                ba[r] = cst_bo[row];
                bb[r] = cst_bo[H + row];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = (wn * NT + n) * 32 + l31;
                const int pos = l0 + col;
                const int voff = pos < L ? (4 * lhi * L + pos) * 4 : OOB;
                float xr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int soff = (mt_h[m] * 32 + (r & 3) + 8 * (r >> 2)) * L4;
                    if constexpr (VEC) xr[r] = ut[(mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P + col];
                    else xr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, voff, soff, 0));
                    if (has_mel) xr[r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voff, soff, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int h = mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float oa = acc_a[m][n][r] + ba[r], ob = acc_b[m][n][r] + bb[r];
                    tile[h * P + col] = xr[r] + oa * sigmoid_m(ob);
                }
            }
        }
    }
    stamp(4);        // GLU + residual epilogue
    __syncthreads();
    stamp(5);        // barrier

    // ---- LN2 statistics per column (population std, no eps; `sashimi.py:17-20`), centre in place
    column_stats<H, P, PARTS>(tile, red, colmean, colalpha, a.ln_s[0], tid);
    stamp(6);        // LayerNorm statistics (three barriers)

    // ---- FF: per H-row chunk q of u: GEMM-1 chunk -> GELU -> LDS -> GEMM-2 partial
    const float4* A1 = reinterpret_cast<const float4*>(a.A1);
    const float4* A2 = reinterpret_cast<const float4*>(a.A2);
    const float lnm = a.ln_m[0];
    f32x16 acc2[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][n][r] = 0.f;
    for (int q = 0; q < FFE; ++q) {
        f32x16 acc1[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[m][n][r] = 0.f;
        int mt_q[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) mt_q[m] = q * (H / 32) + mt_h[m];
        slab(acc1, A1, H / 8, 0, H / 8, mt_q, tile, wn, lane, ws1);
        if constexpr (SCALED) acc_scale(acc1, 1.f / (ws1 * sx));
        stamp(7 + 3 * (q & 1));     // GEMM-1 chunk q  (stamps 7..12 hold the first two chunks)
        if (q > 0) __syncthreads();  // previous chunk of u fully consumed by GEMM-2
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float rsv[16], b1v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = q * H + mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                rsv[r] = cst_rs[row] * lnm;
                b1v[r] = cst_b1[row];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = (wn * NT + n) * 32 + l31;
                const float al = colalpha[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int hr = mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;  // row inside the chunk
                    const float pre = fmaf(al, acc1[m][n][r] + rsv[r], b1v[r]);      // al * (W1 xc + m rowsum) + b1
                    ut[hr * P + col] = gelu_erf_m(pre);
                }
            }
        }
        __syncthreads();
        stamp(8 + 3 * (q & 1));     // GELU epilogue + barrier(s)
        slab(acc2, A2, FFE * H / 8, q * (H / 8), H / 8, mt_h, ut, wn, lane, ws2);
        stamp(9 + 3 * (q & 1));     // GEMM-2 partial
    }
    if constexpr (SCALED) acc_scale(acc2, 1.f / (ws2 * sx));

    // ---- out = x1 + f (+ addend);  x1 = centred tile + mean
    if constexpr (VEC) {
        __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.addend ? a.addend + (size_t)b * H * L : a.out), 0, H * L4, 0x00020000);
        const bool has_add = a.addend != nullptr;
        const int f4 = tid % F4_ROW, rsub = tid / F4_ROW;
        const int pos4 = l0 + 4 * f4;
        const int voff4 = pos4 < L ? (rsub * L + pos4) * 4 : OOB;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        // the U-Net skip rows of this thread's row-major float4s: requested before the accumulator pass
        f32x4 ad4[NPASS];
        if (has_add) {
#pragma unroll
            for (int i = 0; i < NPASS; ++i)
                ad4[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voff4, i * ROWS_PASS * L4, 0));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float b2v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) b2v[r] = cst_b2[mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = (wn * NT + n) * 32 + l31;
                const float mean = colmean[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int h = mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    tile[h * P + col] = (tile[h * P + col] + mean) + (acc2[m][n][r] + b2v[r]);   // own element
                }
            }
        }
        stamp(13);   // accumulators -> tile
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            float* tp = tile + (i * ROWS_PASS + rsub) * P + 4 * f4;
            f32x4 v = *reinterpret_cast<const f32x4*>(tp);
            if (has_add) {
                v += ad4[i];
                if (a.ynext) *reinterpret_cast<f32x4*>(tp) = v;   // own float4: the block output stays in LDS
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rO, voff4, i * ROWS_PASS * L4, 0);
        }
        stamp(14);   // barrier + output stores issued
        if (a.ynext == nullptr) return;   // uniform

        // ---- the next block's S4 input: LN1_next down the columns of the output tile + its step-embedding projection
        __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ynext + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc((void*)(a.e_next + (size_t)b * a.e_stride + step_row_off(a.e_step, a.e_tstride)), 0, H * 4, 0x00020000);
        float ev[NPASS];
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            ev[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rE, rsub * 4, i * ROWS_PASS * 4, 0));
        __syncthreads();
        column_stats<H, P, PARTS>(tile, red, colmean, colalpha, a.n1_s[0], tid);
        const float n1m = a.n1_m[0];
        const f32x4 al = *reinterpret_cast<const f32x4*>(colalpha + 4 * f4);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(tile + (i * ROWS_PASS + rsub) * P + 4 * f4);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaf(al[e], t[e] + n1m, ev[i]);     // (s/sd)(x - mu + m) + fc_t(e)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rY, voff4, i * ROWS_PASS * L4, 0);
        }
        stamp(15);   // next block's LayerNorm + its stores issued
    } else {
        __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.addend ? a.addend + (size_t)b * H * L : a.out), 0, H * L4, 0x00020000);
        const bool has_add = a.addend != nullptr;
    #pragma unroll
        for (int m = 0; m < MT; ++m) {
            float b2v[16];
    #pragma unroll
            for (int r = 0; r < 16; ++r) b2v[r] = cst_b2[mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
    #pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = (wn * NT + n) * 32 + l31;
                const int pos = l0 + col;
                const int voff = pos < L ? (4 * lhi * L + pos) * 4 : OOB;
                const float mean = colmean[col];
                float ad[16];
                if (has_add) {
    #pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ad[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, voff, (mt_h[m] * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));
                } else {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) ad[r] = 0.f;
                }
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int h = mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float v = (tile[h * P + col] + mean) + (acc2[m][n][r] + b2v[r]) + ad[r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, voff, (mt_h[m] * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0);
                    if (a.ynext) tile[h * P + col] = v;    // this thread's own element: the block output stays in LDS
                }
            }
        }
        if (a.ynext == nullptr) return;   // uniform

        // ---- the next block's S4 input: LN1_next down the columns of the output tile + its step-embedding projection
        __syncthreads();
        column_stats<H, P, PARTS>(tile, red, colmean, colalpha, a.n1_s[0], tid);
        {
            __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ynext + (size_t)b * H * L), 0, H * L4, 0x00020000);
            __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc((void*)(a.e_next + (size_t)b * a.e_stride + step_row_off(a.e_step, a.e_tstride)), 0, H * 4, 0x00020000);
            const float n1m = a.n1_m[0];
    #pragma unroll
            for (int m = 0; m < MT; ++m) {
                float ev[16];
    #pragma unroll
                for (int r = 0; r < 16; ++r)
                    ev[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rE, 16 * lhi, (mt_h[m] * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
    #pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = (wn * NT + n) * 32 + l31;
                    const int pos = l0 + col;
                    const int voff = pos < L ? (4 * lhi * L + pos) * 4 : OOB;
                    const float al = colalpha[col];
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int h = mt_h[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const float y = fmaf(al, tile[h * P + col] + n1m, ev[r]);      // (s/sd)(x - mu + m) + fc_t(e)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rY, voff, (mt_h[m] * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0);
                    }
                }
            }
        }

    }
}

// DWS_TAIL_TRACE=1 (tools only): stamp the phases of every wave of a launch (s_memtime, shader-clock ticks) and print the
// mean ticks per phase on stderr -- the per-phase budget of a tile.  s_memtime is per XCD: only differences
// inside one workgroup are formed.
template <typename F>
static void tail_trace_launch(int H, int nwg, int waves, S4TailArgs a, hipStream_t s, F launch) {
    static const char* names[16] = {"", "staging", "GEMM-o", "barrier", "GLU+res", "barrier", "LN2 stats", "GEMM-1 q0", "GELU q0",
                                    "GEMM-2 q0", "GEMM-1 q1", "GELU q1", "GEMM-2 q1", "acc->tile", "out stores", "next LN1"};
    unsigned long long* d = nullptr;
    const size_t n = (size_t)nwg * waves * 16;
    if (hipMalloc(&d, n * 8) != hipSuccess) return;
    hipMemsetAsync(d, 0, n * 8, s);
    a.trace = d;
    launch(a);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> h(n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    double ph[16] = {0}, span = 0, wgspan = 0;
    for (int g = 0; g < nwg; ++g) {
        unsigned long long g0 = ~0ull, g1 = 0;
        for (int w = 0; w < waves; ++w) {
            const unsigned long long* t = &h[((size_t)g * waves + w) * 16];
            for (int i = 1; i < 16; ++i) ph[i] += (double)(t[i] - t[i - 1]);
            span += (double)(t[15] - t[0]);
            if (t[0] < g0) g0 = t[0];
            if (t[15] > g1) g1 = t[15];
        }
        wgspan += (double)(g1 - g0);
    }
    const double nw = (double)nwg * waves;
    fprintf(stderr, "[tail trace] H=%d L=%d wgs=%d waves/wg=%d  mean ticks per wave:", H, a.L, nwg, waves);
    for (int i = 1; i < 16; ++i) fprintf(stderr, " %s %.0f |", names[i], ph[i] / nw);
    fprintf(stderr, " wave life %.0f, workgroup span %.0f\n", span / nw, wgspan / nwg);
}

// the split instances (H >= 256): the 16-byte tile I/O form only; SP = SplitBf16x3 | SplitF16x2
template <int H, int WM, int WN, int NT, int OCC, typename SP>
static int launch_tail_split_t(const S4TailArgs& a, hipStream_t s) {
    using T = TailCfg<H, WM, WN, NT, 2>;
    ProfileScope ps(SP::NT == 3 ? "s4_tail_mfma_tile6" : "s4_tail_mfma_tile_f16x3", s);
    const int ntl = ceil_div(a.L, T::P);
    const size_t lds = (size_t)T::LDS_FLOATS * 4;
    static bool attr_set_dev[DWS_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[current_device_slot()];
    if (!attr_set) {
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_mfma_kernel<H, WM, WN, NT, 2, true, OCC, false, SP>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((s4_tail_mfma_kernel<H, WM, WN, NT, 2, true, OCC, false, SP>), dim3(a.B * ntl), dim3(T::THREADS), lds, s, a);
    return DWS_OK;
}

template <int H, int WM, int WN, int NT, int OCC, bool KGU>
static int launch_tail_t(const S4TailArgs& a, hipStream_t s) {
    using T = TailCfg<H, WM, WN, NT, 2>;
    ProfileScope ps("s4_tail_mfma", s);
    const int ntl = ceil_div(a.L, T::P);
    const size_t lds = (size_t)T::LDS_FLOATS * 4;
    const bool no_vec = getenv("DWS_TAIL_NO_VEC") != nullptr;   // (read per launch: tests switch it inside one process)
    static bool attr_set_dev[DWS_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[current_device_slot()];
    if (!attr_set) {
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_mfma_kernel<H, WM, WN, NT, 2, true, OCC, KGU>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_mfma_kernel<H, WM, WN, NT, 2, false, OCC, KGU>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    static const bool trace = getenv("DWS_TAIL_TRACE") != nullptr;
    if (trace && (a.L & 3) == 0 && !no_vec && a.ynext) {
        tail_trace_launch(H, a.B * ntl, T::THREADS / 64, a, s, [&](const S4TailArgs& at) {
            hipLaunchKernelGGL((s4_tail_mfma_kernel<H, WM, WN, NT, 2, true, OCC, KGU>), dim3(at.B * ntl), dim3(T::THREADS), lds, s, at);
        });
        return DWS_OK;
    }
    if ((a.L & 3) == 0 && !no_vec)
        hipLaunchKernelGGL((s4_tail_mfma_kernel<H, WM, WN, NT, 2, true, OCC, KGU>), dim3(a.B * ntl), dim3(T::THREADS), lds, s, a);
    else
        hipLaunchKernelGGL((s4_tail_mfma_kernel<H, WM, WN, NT, 2, false, OCC, KGU>), dim3(a.B * ntl), dim3(T::THREADS), lds, s, a);
    return DWS_OK;
}

bool s4_tail_mfma_supported(int H, int ff) {
    return ff == 2 && (H == 32 || H == 64 || H == 128 || H == 256 || H == 512);
}

static int launch_s4_tail_mfma_impl(int H, const S4TailArgs& a, hipStream_t s, bool* ran_split);
int launch_s4_tail_mfma(int H, const S4TailArgs& a, hipStream_t s, bool* ran_split) {
    bool dummy = false;
    return launch_s4_tail_mfma_impl(H, a, s, ran_split ? ran_split : &dummy);
}
static int launch_s4_tail_mfma_impl(int H, const S4TailArgs& a, hipStream_t s, bool* ran_split) {
    // Tile shapes <H, waves along rows, waves along positions, 32-position tiles per wave, workgroups per CU the register
    // budget is set for, k-group loop unrolled>, each the best of a same-box sweep (`profiles/r02_tail_shapes.txt`): up to
    // H = 128 small tiles win (NT = 1, 32-64 positions: a workgroup is MFMA-active less than a third of its life, and
    // four of them per CU hide each other's staging, LayerNorm and epilogue phases better than two); at H = 256 the
    // A-fragment reuse of NT = 2 is worth more.  DWS_TAIL_CFG=1 selects the round-1 shapes.
    const int alt = getenv("DWS_TAIL_CFG") ? atoi(getenv("DWS_TAIL_CFG")) : 0;
    // H <= 64: the register-chained kernel (sashimi_chain.hip: a wave owns 32 positions, no LDS round trip between the
    // GEMMs, no barrier); DWS_TAIL_NO_CHAIN=1 keeps the LDS-tile kernel below (A/B runs, tests)
    if (alt == 0 && a.Ao_c6 && s4_tail_wide6_supported(H, 2) && getenv("DWS_TAIL_NO_CHAIN") == nullptr) {
        *ran_split = true;
        return launch_s4_tail_wide6(H, a, s);
    }       // precision = bf16x6, H = 128: one wave per SIMD, streamed weights
    if (alt == 0 && a.Ao_c6 && s4_tail_chain6_supported(H, 2) && getenv("DWS_TAIL_NO_CHAIN") == nullptr) {
        *ran_split = true;
        return launch_s4_tail_chain6(H, a, s);
    }      // precision = bf16x6: the same chain on the bf16 matrix cores, 3-term split
    if (alt == 0 && a.Ao_c && s4_tail_chain_supported(H, 2) && getenv("DWS_TAIL_NO_CHAIN") == nullptr)
        return launch_s4_tail_chain(H, a, s);
    if (alt == 1) {
        switch (H) {
            case 32: return launch_tail_t<32, 1, 4, 1, 2, true>(a, s);
            case 64: return launch_tail_t<64, 2, 2, 2, 2, true>(a, s);
        }
    }
    // precision = bf16x6 / f16x3 at H >= 256: the LDS-tile kernel with its GEMMs on the 16-bit matrix cores
    if (alt == 0 && a.split_on && (a.L & 3) == 0 && getenv("DWS_TAIL_NO_VEC") == nullptr && getenv("DWS_TAIL_NO_SPLIT_TILE") == nullptr) {
        *ran_split = (H == 256 || H == 512) && (a.split_c6 == WN_SPLIT_BF16X6 || (a.split_c6 == WN_SPLIT_F16X3 && a.wscale_c6));
        if (a.split_c6 == WN_SPLIT_F16X3 && a.wscale_c6) {
            if (H == 256) return launch_tail_split_t<256, 8, 1, 2, 1, SplitF16x2>(a, s);
            if (H == 512) return launch_tail_split_t<512, 16, 1, 1, 1, SplitF16x2>(a, s);
        } else if (a.split_c6 == WN_SPLIT_BF16X6) {
            if (H == 256) return launch_tail_split_t<256, 8, 1, 2, 1, SplitBf16x3>(a, s);
            if (H == 512) return launch_tail_split_t<512, 16, 1, 1, 1, SplitBf16x3>(a, s);
        }
    }
    switch (H) {
        case 32: return launch_tail_t<32, 1, 2, 1, 4, true>(a, s);    // 124.7 us (C4) against 127.9 for <32,1,4,1,2>; two position
                    // tiles per wave (NT = 2, each A fragment feeding two MFMAs; round 3, same box): <32,1,2,2,4> 177, <32,1,1,2,4> 161,
                    // <32,1,4,2,2> 170 us against 126 -- at H = 32 the tile's VALU / barrier phases, not the fragment reuse, are the time
        case 64: return launch_tail_t<64, 2, 2, 1, 4, true>(a, s);    // 141.9 / 89.2 us (C3 / C4) against 157.2 / 109.2 for <64,2,2,2,2>
        case 128:   // 32-position tiles on short stages only: 72.5 us against 77.0 at L = 1000 x 32 clips (C4), 128.6 against
                    // 126.9 at 4000 x 16 (C3), 17.96 against 17.91 ms per config-5 sampling step at 16000 x 16.  The choice
                    // looks at L alone, so a clip's result does not depend on how many neighbours share its batch
                    // (the shapes sum the LayerNorm partials in different orders).
            if (a.L < 2048) return launch_tail_t<128, 4, 1, 1, 4, false>(a, s);
            return launch_tail_t<128, 4, 1, 2, 2, true>(a, s);
        case 256: return launch_tail_t<256, 8, 1, 2, 1, true>(a, s);  // (32-position tiles, two workgroups per CU: 121-149 us against 115)
        case 512: return launch_tail_t<512, 16, 1, 1, 1, true>(a, s); // 32 positions x 16 waves: the tiles fill 139 KB of LDS
    }
    return set_error(DWS_ERR_UNSUPPORTED, "s4_tail_mfma: H=%d not instantiated", H);
}

// rs[o] = sum_k W[o][k]
__global__ void row_sum_kernel(const float* __restrict__ W, float* __restrict__ rs, int K) {
    const int o = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += W[(size_t)o * K + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) rs[o] = s;
}

int launch_row_sum(const float* W, float* rs, int O, int K, hipStream_t s) {
    hipLaunchKernelGGL(row_sum_kernel, dim3(O), dim3(64), 0, s, W, rs, K);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// Pooling 1x1 convs on MFMA.  M = 128*MT output rows, 4 waves along M, P = 64.
//   MODE 0 (DownPool): B operand row k = h*p + j is x[b, h, (l0+col)*p + j]; out[b, o, l]
//   MODE 1 (UpPool):   B operand row k is x[b, k, l]; out[b, o/p, l*p + o%p] (+ addend), p == 4:
//                      a lane's 4 consecutive accumulator rows are the 4 j of one channel -> one float4 store
// ---------------------------------------------------------------------------
// LN: the instance with the LayerNorm epilogue.  (At MT = 4 the accumulators alone are 128 registers and the epilogue's
// statistics spill ~330 bytes; still 9 us cheaper than the plain instance + a LayerNorm launch: 76.8 vs 69.3 + 16.5 us.)
// (Round 6, measured and not kept: the same GEMM on the bf16 matrix cores under precision = bf16x6 -- gemm_slab_split is a drop-in
// for the slab call below.  Same-box A/B: null, C3 4.93 -> 4.96 ms, C4 3.794 -> 3.784 ms per step; the M = 256 DownPool instance
// got 10 us SLOWER (256 VGPRs, spills).  The kernel is bound by its register-staged gather and its epilogue, not by the fp32
// MFMA time: profiles/r06_ab_pool_split_dropped.txt.)
template <int MT, int MODE, bool LN>
__global__ __launch_bounds__(256, 2) void pw_mfma_kernel(PwMfmaArgs a) {
    constexpr int P = 64, NT = 2, KC = 64;
    __shared__ __attribute__((aligned(16))) float lds[2 * KC * P];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L, K = a.K, p = a.p;   // L = number of GEMM columns per batch element
    const int ntl = (L + P - 1) / P;
    const int tix = xcd_remap(blockIdx.x, gridDim.x);
    const int b = tix / ntl, l0 = (tix % ntl) * P;
    const int nchunk = K / KC;
    const float* __restrict__ xb = a.in + (size_t)b * K * L;  // DownPool: [K/p][L*p] has the same element count

    float stg[KC * P / 256];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int i = 0; i < KC * P / 256; ++i) {
            const int e = tid + 256 * i;
            if (MODE == 0) {
                // element e of the contiguous run of (KC/p) channels x (P*p) samples
                const int hh = e / (P * p), w = e % (P * p);
                const int h = c * (KC / p) + hh;
                const size_t Lin = (size_t)L * p;
                const size_t src = (size_t)l0 * p + w;
                const bool ok = src < Lin;
                stg[i] = xb[(size_t)h * Lin + (ok ? src : 0)] * (ok ? 1.f : 0.f);
            } else {
                const int row = e / P, col = e % P;
                const int pos = l0 + col;
                const bool ok = pos < L;
                stg[i] = xb[(size_t)(c * KC + row) * L + (ok ? pos : 0)] * (ok ? 1.f : 0.f);
            }
        }
    };
    auto stage_store = [&](int buf) {
        float* d = lds + buf * KC * P;
#pragma unroll
        for (int i = 0; i < KC * P / 256; ++i) {
            const int e = tid + 256 * i;
            if (MODE == 0) {
                const int hh = e / (P * p), w = e % (P * p);
                d[(hh * p + (w % p)) * P + (w / p)] = stg[i];
            } else {
                d[e] = stg[i];
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    int mt[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) mt[m] = blockIdx.y * (4 * MT) + wave * MT + m;   // grid.y walks 128*MT-row blocks of M
    const float4* A = reinterpret_cast<const float4*>(a.A);

    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) stage_load(c + 1);
        gemm_slab<MT, NT, P, true>(acc, A, K / 8, c * (KC / 8), KC / 8, mt, lds + (c & 1) * KC * P, 0, lane);
        if (c + 1 < nchunk) stage_store((c + 1) & 1);
        __syncthreads();
    }

    // ---- optional LayerNorm of the output columns (the next block's LN1 + step embedding), from the accumulators:
    // a lane holds 16 MT rows of NT columns; MODE 1 interleaves four output positions (j = row & 3) per GEMM column, each
    // with its own statistics over the M/4 channels
    constexpr int NJ = MODE == 0 ? 1 : 4;
    const bool ln = LN && a.ln_y != nullptr;      // (uniform; the launcher only sets it when this workgroup owns all M rows)
    float lnscale[NT][NJ], lnshift[NT][NJ];
    if (ln) {
        float* red = lds;                   // [4 waves][64 columns][NJ]: the staging buffers are free after the last barrier
        const float inv_n = 1.f / (float)(a.M / NJ);
        float part[NT][NJ];
        auto reduce = [&](float (&out)[NT][NJ]) {      // sum `part` over the lane halves and the four waves
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    part[n][j] += __shfl_xor(part[n][j], 32);
                    if (lhi == 0) red[(wave * 64 + n * 32 + l31) * NJ + j] = part[n][j];
                }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) t += red[(w * 64 + n * 32 + l31) * NJ + j];
                    out[n][j] = t;
                }
            __syncthreads();
        };
        // the values the LayerNorm sees: accumulator + bias (+ U-Net skip)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int pos = l0 + n * 32 + l31;
                const int posc = pos < L ? pos : 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int o0 = mt[m] * 32 + 8 * qd + 4 * lhi;      // rows o0 .. o0+3
                    float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (MODE == 1 && a.addend)      // the four rows are the four positions of one channel: one 16-byte load
                        ad = *reinterpret_cast<const float4*>(a.addend + (size_t)b * (a.M / 4) * L * 4 + ((size_t)(o0 / 4) * L + posc) * 4);
                    acc[m][n][qd * 4 + 0] = (acc[m][n][qd * 4 + 0] + a.bias[o0 + 0]) + ad.x;   // (the plain epilogue's order)
                    acc[m][n][qd * 4 + 1] = (acc[m][n][qd * 4 + 1] + a.bias[o0 + 1]) + ad.y;
                    acc[m][n][qd * 4 + 2] = (acc[m][n][qd * 4 + 2] + a.bias[o0 + 2]) + ad.z;
                    acc[m][n][qd * 4 + 3] = (acc[m][n][qd * 4 + 3] + a.bias[o0 + 3]) + ad.w;
                }
            }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < NJ; ++j) part[n][j] = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[n][MODE == 0 ? 0 : (r & 3)] += acc[m][n][r];
        float mean[NT][NJ], var[NT][NJ];
        reduce(mean);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < NJ; ++j) { mean[n][j] *= inv_n; part[n][j] = 0.f; }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[m][n][r] - mean[n][MODE == 0 ? 0 : (r & 3)];
                    part[n][MODE == 0 ? 0 : (r & 3)] = fmaf(d, d, part[n][MODE == 0 ? 0 : (r & 3)]);
                }
        reduce(var);
        const float s_p = a.ln_s[0], m_p = a.ln_m[0];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                lnscale[n][j] = s_p / sqrtf(var[n][j] * inv_n);      // population std, no eps (`sashimi.py:17-20`)
                lnshift[n][j] = m_p - mean[n][j];
            }
    }
    const float* __restrict__ lne = ln ? a.ln_e + (size_t)b * a.ln_e_stride + step_row_off(a.ln_step, a.ln_e_tstride) : nullptr;

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int pos = l0 + n * 32 + l31;
            const bool ok = pos < L;
            if (MODE == 0) {
                float* __restrict__ ob = a.out + (size_t)b * a.M * L;
                float* __restrict__ yb = ln ? a.ln_y + (size_t)b * a.M * L : nullptr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = mt[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float v = ln ? acc[m][n][r] : acc[m][n][r] + a.bias[o];
                    if (ok) ob[(size_t)o * L + pos] = v;
                    if (ln && ok) yb[(size_t)o * L + pos] = fmaf(lnscale[n][0], v + lnshift[n][0], lne[o]);
                }
            } else {
                const int Ho = a.M / 4;
                float* __restrict__ ob = a.out + (size_t)b * Ho * L * 4;
                float* __restrict__ yb = ln ? a.ln_y + (size_t)b * Ho * L * 4 : nullptr;
                const float* __restrict__ adb = a.addend ? a.addend + (size_t)b * Ho * L * 4 : nullptr;
                const int posc = ok ? pos : 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int o0 = mt[m] * 32 + 8 * qd + 4 * lhi;   // rows o0..o0+3 = channel o0/4, j = 0..3
                    const size_t idx = ((size_t)(o0 / 4) * L + posc) * 4;
                    float4 v;
                    if (ln) {     // bias and skip are in the accumulators already
                        v = make_float4(acc[m][n][qd * 4 + 0], acc[m][n][qd * 4 + 1], acc[m][n][qd * 4 + 2], acc[m][n][qd * 4 + 3]);
                    } else {
                        v = make_float4(acc[m][n][qd * 4 + 0] + a.bias[o0], acc[m][n][qd * 4 + 1] + a.bias[o0 + 1],
                                        acc[m][n][qd * 4 + 2] + a.bias[o0 + 2], acc[m][n][qd * 4 + 3] + a.bias[o0 + 3]);
                        if (adb) {
                            const float4 ad = *reinterpret_cast<const float4*>(adb + idx);
                            v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                        }
                    }
                    if (ok) *reinterpret_cast<float4*>(ob + idx) = v;
                    if (ln && ok) {
                        const float e = lne[o0 / 4];
                        *reinterpret_cast<float4*>(yb + idx) =
                            make_float4(fmaf(lnscale[n][0], v.x + lnshift[n][0], e), fmaf(lnscale[n][NJ > 1 ? 1 : 0], v.y + lnshift[n][NJ > 1 ? 1 : 0], e),
                                        fmaf(lnscale[n][NJ > 2 ? 2 : 0], v.z + lnshift[n][NJ > 2 ? 2 : 0], e),
                                        fmaf(lnscale[n][NJ > 3 ? 3 : 0], v.w + lnshift[n][NJ > 3 ? 3 : 0], e));
                    }
                }
            }
        }
}

// the LayerNorm epilogue reduces over ALL output rows inside one workgroup (128 MT rows, MT <= 4)
bool pw_mfma_ln_supported(int M) { return M == 128 || M == 256 || M == 512; }

bool pw_mfma_supported(int mode, int K, int M, int p) {
    if (K % 64 != 0 || (M != 128 && M != 256 && M % 512 != 0)) return false;
    if (mode == 0) return p >= 1 && 64 % p == 0;
    return p == 4;
}

template <int MODE>
static int launch_pw_mode(const PwMfmaArgs& a, hipStream_t s) {
    DWS_CHECK(a.ln_y == nullptr || pw_mfma_ln_supported(a.M), DWS_ERR_INVALID,
              "pw_mfma: the LayerNorm epilogue needs all %d rows in one workgroup", a.M);
    const int grid = a.B * ceil_div(a.L, 64);
#define DWS_PW(MTV)                                                                                                 \
    do {                                                                                                            \
        const dim3 g(grid, (MTV) == 4 ? a.M / 512 : 1);                                                              \
        if (a.ln_y) hipLaunchKernelGGL((pw_mfma_kernel<MTV, MODE, true>), g, dim3(256), 0, s, a);                    \
        else hipLaunchKernelGGL((pw_mfma_kernel<MTV, MODE, false>), g, dim3(256), 0, s, a);                          \
    } while (0)
    if (a.M == 128) DWS_PW(1);
    else if (a.M == 256) DWS_PW(2);
    else DWS_PW(4);
#undef DWS_PW
    return DWS_OK;
}

int launch_pw_mfma(int mode, const PwMfmaArgs& a, hipStream_t s) {
    ProfileScope ps(mode == 0 ? "pw_mfma_down" : "pw_mfma_up", s);
    return mode == 0 ? launch_pw_mode<0>(a, s) : launch_pw_mode<1>(a, s);
}

}  // namespace dws
