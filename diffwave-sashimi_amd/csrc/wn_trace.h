// Phase trace of the fused WaveNet layer kernels (tools only): every wave stamps s_memtime at the phase boundaries
// into trace[tile][wave][32]; the launcher prints per-phase means.  Shared by wavenet_wino.hip and wavenet_bx6.hip.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wavenet.h"

namespace dws {

// DWS_WINO_TRACE=1 (tools only): stamp the phases of every wave of the first traced launch and print a summary
template <typename F>
static void wino_trace_launch(int nwg, int waves, WnLayerArgs a, hipStream_t s, F launch, const char* tag = "wino") {
    unsigned long long* d = nullptr;
    const size_t n = (size_t)nwg * waves * 32;
    if (hipMalloc(&d, n * 8) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, n * 8, s);
    a.trace = d;
    launch(a);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    double ph[8] = {0};
    static double chunk[8][32];
    for (auto& c : chunk) for (double& v : c) v = 0;
    double g2[8] = {0}, ep[8] = {0};
    double wgspan = 0;   // (s_memtime is per XCD: only differences inside one workgroup mean anything)
    for (int g = 0; g < nwg; ++g) {
        unsigned long long g0 = ~0ull, g1 = 0;
        for (int w = 0; w < waves; ++w) {
            const unsigned long long* t = &h[((size_t)g * waves + w) * 32];
            for (int i = 1; i < 8; ++i) ph[i] += (double)(t[i] - t[i - 1]);
            if (w < 8) { g2[w] += (double)(t[5] - t[4]); ep[w] += (double)(t[6] - t[5]); }
            if (w < 8)
                for (int i = 8; i < 32; ++i)
                    if (t[i]) { chunk[w][i] += (double)(t[i] - t[1]); }
            if (t[0] < g0) g0 = t[0];
            if (t[7] > g1) g1 = t[7];
        }
        wgspan += (double)(g1 - g0);
    }
    const double nw = (double)nwg * waves;
    fprintf(stderr, "[%s trace] d=%d wgs=%d per-wave mean cycles: prologue %.0f gemm1 %.0f "
            "extra+gate %.0f barrier %.0f gemm2 %.0f epilogue %.0f drain %.0f; mean workgroup span %.0f\n", tag, a.dilation, nwg, ph[1] / nw, ph[2] / nw, ph[3] / nw, ph[4] / nw, ph[5] / nw,
            ph[6] / nw, ph[7] / nw, wgspan / nwg);
    if (std::getenv("DWS_WINO_TRACE_CHUNKS")) {
        fprintf(stderr, "  per-wave gemm2 / epilogue ticks:");
        for (int w = 0; w < waves && w < 8; ++w) fprintf(stderr, " %.0f/%.0f", g2[w] / nwg, ep[w] / nwg);
        fprintf(stderr, "\n");
    }
    if (std::getenv("DWS_WINO_TRACE_CHUNKS"))
        for (int w = 0; w < waves && w < 8; ++w) {
            fprintf(stderr, "  wave %d arrive/release since gemm1 start:", w);
            for (int i = 8; i < 24; ++i) fprintf(stderr, " %.0f%s", chunk[w][i] / nwg, (i & 1) ? " |" : "");
            fprintf(stderr, "\n");
        }
}

}  // namespace dws
