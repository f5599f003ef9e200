// fp32-equivalent GEMMs on the bf16 matrix cores: the 3-term split ("bf16x6").
//
//   x = x0 + x1 + x2   with  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)        (round to nearest even)
//
// The two differences are exact in fp32 (x - x0 has <= 16 significant bits, the second remainder <= 8), so the three
// bf16 terms carry all 24 significand bits of x: the split is EXACT (no subnormals / infinities assumed).  A product
// x*w is then nine bf16 products; the six of order 2^0, 2^-9, 2^-18
//
//   x0 w0  |  x0 w1, x1 w0  |  x0 w2, x1 w1, x2 w0
//
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (each bf16 product is exact in the fp32 accumulator format);
// the three dropped ones (x1 w2, x2 w1, x2 w2) are <= 2^-27 + 2^-27 + 2^-36 of |x w| -- a quarter of an fp32 ulp.
// What remains is the rounding of the fp32 accumulation itself, i.e. the error class of an fp32 GEMM.  Six bf16 MFMAs
// cost 6/16 of the one exact-f32 MFMA they replace (MI355X: bf16 2.5 PFLOP/s, f32 157 TFLOP/s) and, unlike it, do
// not occupy the VALU pipe.  tests/test_bf16x6_gpu.py measures both paths against a float64 evaluation.
#pragma once
#include <hip/hip_runtime.h>

namespace dws {

typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx_bf16x4 __attribute__((ext_vector_type(4)));
typedef float bx_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
    p0 = (__bf16)x;
    const float r1 = x - (float)p0;
    p1 = (__bf16)r1;
    const float r2 = r1 - (float)p1;
    p2 = (__bf16)r2;
}

// four values -> one 8-byte half item per part (element e of the item = value e)
__device__ __forceinline__ void split3x4(const float (&x)[4], bx_bf16x4& p0, bx_bf16x4& p1, bx_bf16x4& p2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 a, b, c;
        split3(x[e], a, b, c);
        p0[e] = a; p1[e] = b; p2[e] = c;
    }
}

// order of the six products: smallest terms first (they meet an accumulator that is still small in the first k-block;
// later it does not matter), the leading product last
__device__ constexpr int BX6_IA[6] = {2, 1, 0, 1, 0, 0};
__device__ constexpr int BX6_IB[6] = {0, 1, 2, 0, 1, 0};

// acc += (a0 + a1 + a2) . (b0 + b1 + b2) without the three products below 2^-26
__device__ __forceinline__ void mfma6(bx_f32x16& acc, const bx_bf16x8 (&a)[3], const bx_bf16x8 (&b)[3]) {
#pragma unroll
    for (int t = 0; t < 6; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BX6_IA[t]], b[BX6_IB[t]], acc, 0, 0, 0);
}

// A / B fragment of a rank-2 correction k-block: k = 0 carries v0, k = 1 carries v1 (lanes of the lower k half only),
// everything else 0.  Used for biases and the step-embedding rows (B side: indicator values, exact in bf16).
__device__ __forceinline__ void frag_rank2(float v0, float v1, bool lower_half, bx_bf16x8 (&out)[3]) {
    __bf16 a0, a1, a2, b0, b1, b2;
    split3(lower_half ? v0 : 0.f, a0, a1, a2);
    split3(lower_half ? v1 : 0.f, b0, b1, b2);
    const __bf16 z = (__bf16)0.f;
    out[0] = bx_bf16x8{a0, b0, z, z, z, z, z, z};
    out[1] = bx_bf16x8{a1, b1, z, z, z, z, z, z};
    out[2] = bx_bf16x8{a2, b2, z, z, z, z, z, z};
}

// ---- split policies: what the templated GEMM code of wavenet_bx6.hip needs to know about a split ------------------------
// SplitBf16x3: the exact 3-term bf16 split above (six products): fp32-FAITHFUL, every operand carries its 24 bits.
struct SplitBf16x3 {
    typedef bx_bf16x8 v8;
    typedef bx_bf16x4 v4;
    static constexpr int NT = 3, NP = 6;           // terms per operand, partial products per term pair
    static constexpr bool SCALED = false;          // bf16 has fp32's exponent range: no operand scaling
    static constexpr float SX = 1.f, SG = 1.f;
    __device__ static constexpr int ia(int t) { return BX6_IA[t]; }
    __device__ static constexpr int ib(int t) { return BX6_IB[t]; }
    __device__ static __forceinline__ bx_f32x16 mfma(const v8& a, const v8& b, const bx_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void split4(const float (&x)[4], v4 (&p)[3]) { split3x4(x, p[0], p[1], p[2]); }
    __device__ static __forceinline__ void split1(float x, v8 (&out)[3], int i) {
        __bf16 a, b, c;
        split3(x, a, b, c);
        out[0][i] = a; out[1][i] = b; out[2][i] = c;
    }
    __device__ static __forceinline__ void rank2(float v0, float v1, bool lower_half, v8 (&out)[3]) { frag_rank2(v0, v1, lower_half, out); }
    __device__ static __forceinline__ void bits(float x, unsigned short (&b)[3]) {   // the terms as stored by the packers
        __bf16 p0, p1, p2;
        split3(x, p0, p1, p2);
        b[0] = __builtin_bit_cast(unsigned short, p0); b[1] = __builtin_bit_cast(unsigned short, p1); b[2] = __builtin_bit_cast(unsigned short, p2);
    }
    __device__ static __forceinline__ v8 bvals(float v0, float v1) {   // B fragment of a correction k-block: exact small values
        const __bf16 z = (__bf16)0.f;
        return v8{(__bf16)v0, (__bf16)v1, z, z, z, z, z, z};
    }
};

// SplitF16x2 ("f16x3"): x = h + l with h = fp16(x), l = fp16(x - h): 22 significand bits, THREE products (h h, h l, l h; the
// dropped l l is 2^-22 of |x w|).  Not fp32-faithful element by element -- 2 bits short, and fp16's exponent range needs
// power-of-two operand scaling so that the low terms stay normal (activations x 2^4, gate x 2^12, each weight matrix by its
// own power of two; all undone exactly on the fp32 accumulators) -- but in a K >= 256 contraction its representation error
// is a fraction of the fp32 ACCUMULATION error that both splits and the f32 path share, at half the matrix work of bf16x6.
// Accepted by the same float64 criterion (tests/test_bf16x6_gpu.py); activations beyond 2^11 overflow fp16 (the Winograd-transformed
// differences of two of them reach 2^12 x 2^4 = fp16's largest binade); measured with scales 2^2, 2^4 and 2^6: the network-level error is
// the same to three digits (the fp32 accumulation dominates), so the smallest scale that keeps O(1) low terms normal is used.
typedef _Float16 hx_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hx_f16x4 __attribute__((ext_vector_type(4)));
struct SplitF16x2 {
    typedef hx_f16x8 v8;
    typedef hx_f16x4 v4;
    static constexpr int NT = 2, NP = 3;
    static constexpr bool SCALED = true;
#ifndef F16X3_SX
#define F16X3_SX 16.f
#endif
    static constexpr float SX = F16X3_SX, SG = 4096.f;
    __device__ static constexpr int ia(int t) { return t == 0 ? 1 : 0; }     // (l, h), (h, l), (h, h)
    __device__ static constexpr int ib(int t) { return t == 1 ? 1 : 0; }
    __device__ static __forceinline__ bx_f32x16 mfma(const v8& a, const v8& b, const bx_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void split4(const float (&x)[4], v4 (&p)[2]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const _Float16 h = (_Float16)x[e];
            p[0][e] = h;
            p[1][e] = (_Float16)(x[e] - (float)h);
        }
    }
    __device__ static __forceinline__ void split1(float x, v8 (&out)[2], int i) {
        const _Float16 h = (_Float16)x;
        out[0][i] = h;
        out[1][i] = (_Float16)(x - (float)h);
    }
    __device__ static __forceinline__ void bits(float x, unsigned short (&b)[2]) {
        const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
        b[0] = __builtin_bit_cast(unsigned short, h); b[1] = __builtin_bit_cast(unsigned short, l);
    }
    __device__ static __forceinline__ void rank2(float v0, float v1, bool lower_half, v8 (&out)[2]) {
        const _Float16 z = (_Float16)0.f;
        out[0] = v8{z, z, z, z, z, z, z, z};
        out[1] = out[0];
        split1(lower_half ? v0 : 0.f, out, 0);
        split1(lower_half ? v1 : 0.f, out, 1);
    }
    __device__ static __forceinline__ v8 bvals(float v0, float v1) {
        const _Float16 z = (_Float16)0.f;
        return v8{(_Float16)v0, (_Float16)v1, z, z, z, z, z, z};
    }
};

}  // namespace dws
