// Launch interface of sashimi_mfma.hip (internal to libdws.so).
#pragma once
#include "dws_common.h"

namespace dws {

struct S4TailArgs {
    const float* g;       // [B,H,L] GELU(conv + D u)
    const float* x;       // [B,H,L] block input (residual)
    const float* Ao;      // packed output_linear weight [2H x H]
    const float* bo;      // [2H]
    const float* mel;     // nullable conditioner term [Bm,H,L]
    int mel_bstride;
    const float* ln_m;    // norm2.m, norm2.s (device scalars)
    const float* ln_s;
    const float* A1;      // packed ff[0] weight [ff*H x H]
    const float* b1;      // [ff*H]
    const float* rs1;     // [ff*H] row sums of the folded ff[0] weight
    const float* A2;      // packed ff[2] weight [H x ff*H]
    const float* b2;      // [H]
    const float* addend;  // nullable U-Net skip [B,H,L]
    float* out;           // [B,H,L]
    // optional: also emit the NEXT block's S4 input  y = LN1_next(out) + fc_t_next(e)[b, h]  (`sashimi.py:148-152`),
    // so the next block of the same stage needs no separate LayerNorm pass
    float* ynext;         // nullable [B,H,L]
    const float* n1_m;    // next block's norm1.m, norm1.s (device scalars)
    const float* n1_s;
    const float* e_next;  // next block's step-embedding projection: e_next[b * e_stride + h]
    int e_stride;
    const int* e_step;    // nullable: step-table mode (sampler) -- e_next is row 0 of a [T][e_tstride] table, the kernel
    int e_tstride;        // adds row *e_step (device-resident step counter; e_stride is 0 then)
    int B, L;
    // H <= 64: the same three weights with their columns in chain order (sashimi_chain.hip), A-fragment packed
    const float* Ao_c;
    const float* A1_c;
    const float* A2_c;
    // precision = "bf16x6" (sashimi_chain6.hip): the three weights as 3-term bf16 fragments of v_mfma_f32_32x32x16_bf16 in
    // the 16-wide chain order; non-null selects the split kernel for H <= 64
    const void* Ao_c6;
    const void* A1_c6;
    const void* A2_c6;
    int split_on;                // precision = bf16x6 / f16x3 requested (H >= 256 runs the LDS-tile kernel's split instances on Ao / A1 / A2)
    int split_c6;                // WN_SPLIT_BF16X6 | WN_SPLIT_F16X3 (wavenet.h): which split
    const float* wscale_c6;      // f16x3: the power of two each of Wo, W1, W2 was packed with [3]
    unsigned long long* trace;   // nullable (tools only, DWS_TAIL_TRACE=1): s_memtime stamps [workgroup][wave][16] of the
                                 // LDS-tile kernel's phases
};

struct PwMfmaArgs {
    const float* in;
    const float* A;       // packed weight [M x K]
    const float* bias;    // [M]
    const float* addend;  // UpPool only, nullable
    float* out;
    int B, K, M, L, p;    // L = GEMM columns per batch element (DownPool: output length; UpPool: input length)
    // optional (all M rows in one workgroup): also emit the S4 input of the block that runs next,
    //   y = LN1_next(out) + fc_t_next(e)[b, channel]   (`sashimi.py:148-152`), from the output tile still in registers
    float* ln_y;          // nullable, same layout as out
    const float* ln_m;    // next block's norm1.m, norm1.s (device scalars)
    const float* ln_s;
    const float* ln_e;    // its step-embedding projection: ln_e[b * ln_e_stride + channel] (+ row *ln_step of a step table)
    int ln_e_stride, ln_e_tstride;
    const int* ln_step;
};

bool s4_tail_mfma_supported(int H, int ff);
// *ran_split (optional) reports whether a split-precision instance was launched or an exact-f32 one (the models count them
// behind the tap "split_launches": which arithmetic a forward really ran)
int launch_s4_tail_mfma(int H, const S4TailArgs& a, hipStream_t s, bool* ran_split = nullptr);
bool s4_tail_chain_supported(int H, int ff);
int launch_s4_tail_chain(int H, const S4TailArgs& a, hipStream_t s);
int launch_chain_permute_cols(const float* w, float* out, int M, int K, hipStream_t s);
bool s4_tail_chain6_supported(int H, int ff);
int launch_s4_tail_chain6(int H, const S4TailArgs& a, hipStream_t s);
int launch_chain16_permute_cols(const float* w, float* out, int M, int K, hipStream_t s);
// H = 128: the chain with one wave per SIMD and the weights streamed through an LDS ring; Ao_c6 then points at ONE blob
// [Wo | W1 | W2] of k-block-major fragments (pack_a_bx6_kmajor)
bool s4_tail_wide6_supported(int H, int ff);
int launch_s4_tail_wide6(int H, const S4TailArgs& a, hipStream_t s);
int launch_pack_a_bx6_kmajor(const float* w, void* out, int M, int K, int split, const float* scale, hipStream_t s);
int launch_row_sum(const float* W, float* rs, int O, int K, hipStream_t s);
bool pw_mfma_supported(int mode, int K, int M, int p);
bool pw_mfma_ln_supported(int M);
int launch_pw_mfma(int mode, const PwMfmaArgs& a, hipStream_t s);

}  // namespace dws
