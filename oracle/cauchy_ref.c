/* Oracle (test infrastructure only): plain-C fp64 restatement of the symmetric
 * Cauchy multiply and its backward, term by term as the reference kernel
 * evaluates them (extensions/cauchy/cauchy_cuda.cu:331 forward, :420-447
 * backward).  Compiled by oracle/Makefile into oracle/libcauchy_ref.so and used
 * by tests/test_cauchy_oracle.py to cross-check the torch restatement in
 * oracle/cauchy.py against an independent implementation.
 *
 * Complex arrays are interleaved (re, im) doubles.
 *   v, w : [B, N] (half state)   z : [L]   out, dout : [B, L]   dv, dw : [B, N] */
#include <complex.h>
#include <stdint.h>

void cauchy_sym_fwd_ref(const double* v_, const double* z_, const double* w_, double* out_,
                        int64_t B, int64_t N, int64_t L) {
    const double complex* v = (const double complex*)v_;
    const double complex* z = (const double complex*)z_;
    const double complex* w = (const double complex*)w_;
    double complex* out = (double complex*)out_;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t l = 0; l < L; ++l) {
            double complex acc = 0;
            for (int64_t n = 0; n < N; ++n) {
                const double complex vv = v[b * N + n], ww = w[b * N + n];
                acc += vv / (z[l] - ww) + conj(vv) / (z[l] - conj(ww));
            }
            out[b * L + l] = acc;
        }
}

void cauchy_sym_bwd_ref(const double* v_, const double* z_, const double* w_, const double* dout_,
                        double* dv_, double* dw_, int64_t B, int64_t N, int64_t L) {
    const double complex* v = (const double complex*)v_;
    const double complex* z = (const double complex*)z_;
    const double complex* w = (const double complex*)w_;
    const double complex* dout = (const double complex*)dout_;
    double complex* dv = (double complex*)dv_;
    double complex* dw = (double complex*)dw_;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t n = 0; n < N; ++n) {
            const double complex wc = conj(w[b * N + n]);
            double complex sdv = 0, sdw = 0;
            for (int64_t l = 0; l < L; ++l) {
                const double complex d1 = conj(z[l]) - wc, d2 = z[l] - wc;
                const double complex t1 = dout[b * L + l] / d1, t2 = conj(dout[b * L + l]) / d2;
                sdv += t1 + t2;
                sdw += t1 / d1 + t2 / d2;
            }
            dv[b * N + n] = sdv;
            dw[b * N + n] = sdw * conj(v[b * N + n]);
        }
}
