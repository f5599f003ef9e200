// Mel-spectrogram front-end of the vocoding path (`dataloaders/stft.py:100-244`,
// `dataloaders/mel2samp.py:76-82`): reflect-pad by n_fft/2, windowed DFT magnitude at hop
// `hop`, mel filterbank, log(clamp(., clip)).  One workgroup per frame: the windowed frame,
// a cos/sin table and the magnitudes live in LDS; a thread owns DFT bins k = tid, tid+256, ...
// (direct O(n_fft^2 / 2) transform: 63 frames of 1024 samples per second of audio is 33 MFLOP,
// not worth an FFT), then 80 mel rows are reduced by one wave each.
#include "dws_common.h"

namespace dws {

__global__ __launch_bounds__(256) void mel_frame_kernel(const float* __restrict__ audio, const float* __restrict__ window,
                                                        const float* __restrict__ basis, float* __restrict__ out, int T,
                                                        int n_fft, int hop, int n_mels, int n_frames, float clip) {
    extern __shared__ float lds[];
    float* frame = lds;                 // [n_fft]
    float* ct = frame + n_fft;          // [n_fft] cos(2 pi j / n_fft)
    float* st = ct + n_fft;             // [n_fft] sin(2 pi j / n_fft)
    float* mag = st + n_fft;            // [n_fft/2 + 1]
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int half = n_fft / 2, nb = half + 1;
    const float* a = audio + (size_t)b * T;
    for (int i = tid; i < n_fft; i += 256) {
        int j = f * hop + i - half;      // index into the un-padded signal; reflect (no edge repeat) outside
        if (j < 0) j = -j;
        if (j >= T) j = 2 * (T - 1) - j;
        frame[i] = a[j] * window[i];
        float s, c;
        sincospif(2.f * (float)i / (float)n_fft, &s, &c);
        ct[i] = c;
        st[i] = s;
    }
    __syncthreads();
    const int mask = n_fft - 1;          // n_fft is a power of two
    for (int k = tid; k < nb; k += 256) {
        float re = 0.f, im = 0.f;
        int ph = 0;
        for (int i = 0; i < n_fft; ++i) {
            const float v = frame[i];
            re = fmaf(v, ct[ph], re);
            im = fmaf(v, st[ph], im);
            ph = (ph + k) & mask;
        }
        mag[k] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int m = wave; m < n_mels; m += 4) {
        const float* w = basis + (size_t)m * nb;
        float s = 0.f;
        for (int k = lane; k < nb; k += 64) s = fmaf(w[k], mag[k], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) out[((size_t)b * n_mels + m) * n_frames + f] = logf(fmaxf(s, clip));
    }
}

}  // namespace dws

extern "C" int dws_mel_spectrogram(const float* audio, int64_t B, int64_t T, const float* window, const float* mel_basis,
                                   int32_t n_fft, int32_t hop, int32_t n_mels, float clip, float* out, void* stream) {
    using namespace dws;
    DWS_CHECK(audio && window && mel_basis && out, DWS_ERR_INVALID, "dws_mel_spectrogram: null argument");
    DWS_CHECK(B > 0 && hop > 0 && n_mels > 0, DWS_ERR_INVALID, "dws_mel_spectrogram: bad shape");
    DWS_CHECK(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0, DWS_ERR_UNSUPPORTED,
              "dws_mel_spectrogram: filter_length %d (powers of two 64..4096 are built)", n_fft);
    DWS_CHECK(T > n_fft / 2, DWS_ERR_INVALID, "dws_mel_spectrogram: reflect padding needs T > filter_length/2 (`stft.py:127-131`)");
    const int n_frames = (int)(T / hop) + 1;
    const size_t lds = (size_t)(3 * n_fft + n_fft / 2 + 1) * 4;
    DWS_HIP(hipFuncSetAttribute((const void*)mel_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mel_frame_kernel, dim3(n_frames, (unsigned)B), dim3(256), lds, (hipStream_t)stream, audio, window,
                       mel_basis, out, (int)T, n_fft, hop, n_mels, n_frames, clip);
    DWS_HIP(hipGetLastError());
    return DWS_OK;
}
