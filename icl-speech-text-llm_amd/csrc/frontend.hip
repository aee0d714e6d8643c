// frontend.hip — audio front-ends of the ICL path on gfx950:
//   K1  Whisper log-mel   (WhisperFeatureExtractor semantics; data/model_processors.py:641-645)
//   K4  BEATs Kaldi fbank (torchaudio.compliance.kaldi.fbank semantics as used by BEATs.preprocess)
// Both are tiny (≈1 GFLOP per 30 s clip) and are computed in f64 (direct DFT from an LDS twiddle
// table: n_fft = 400 is not a power of two and MI355X has full-rate f64 vector FMA), so the result
// is closer to the exact value than the reference's f32 FFT; outputs are f32.
// Layout: one block = FR consecutive frames of one audio; frames are staged (windowed) in LDS, folded into
// s[n] = x[n] + x[N-n], d[n] = x[n] - x[N-n], and every thread owns a PAIR of frequency bins (k, N/2 - k) for all
// FR frames (the frame sample is an LDS broadcast, the twiddle a per-lane LDS read shared by both bins): a quarter of
// the plain DFT's FMAs and LDS reads.  Loads of the waveform are coalesced; nothing is re-read from HBM.
#include "common.h"

namespace {

constexpr int FR = 8;            // frames per block
constexpr int WLEN = 400;        // window length (both front-ends)
constexpr int HOP = 160;
constexpr int WH_NFFT = 400, WH_BINS = 201, WH_FRAMES = 3000, WH_SAMPLES = 480000;
constexpr int FB_NFFT = 512, FB_BINS = 256, FB_MEL = 128;  // Nyquist bin has a zero mel column

__device__ __forceinline__ int float_to_ordered(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// power[f][bin] = |sum_n x[f][n] * exp(-2 pi i bin n / NFFT)|^2, one bin per thread, x real with x[n] = 0 for n >= WLEN.
// Real-input symmetry: cos(2 pi k (N-n)/N) = cos(2 pi k n/N) and sin(...(N-n)) = -sin(...n), so with
//   s[n] = x[n] + x[N-n],  d[n] = x[n] - x[N-n]   (n = 1 .. N/2-1, folded IN PLACE into the frame rows beforehand)
//   Re X[k] = x[0] + (-1)^k x[N/2] + sum_n s[n] cos(2 pi k n/N),   Im X[k] = -sum_n d[n] sin(2 pi k n/N)
// the loop runs over N/2-1 sample pairs with one twiddle read and 2 FMAs per frame each: half the FMAs and half the LDS
// twiddle traffic of the plain sum.  FSTR = row pitch of the frame array (>= NFFT so that n and N-n both have a slot).
template <int NFFT, int FSTR>
__device__ __forceinline__ void fold_frames(double* frames) {
  constexpr int HALF = NFFT / 2;
  for (int i = threadIdx.x; i < FR * (HALF - 1); i += blockDim.x) {
    const int f = i / (HALF - 1), n = 1 + i - f * (HALF - 1);
    if (NFFT - n < WLEN) {          // the partner sample exists (always for NFFT = WLEN; n >= NFFT - WLEN + 1 when zero-padded)
      double* row = frames + f * FSTR;
      const double a = row[n], b = row[NFFT - n];
      row[n] = a + b;
      row[NFFT - n] = a - b;
    }
  }
}
// Bin symmetry on top of it: theta_n(N/2 - k) = pi n - theta_n(k), so cos -> (-1)^n cos and sin -> -(-1)^n sin.  With the
// sums split by the parity of n,  E = sum_even s[n] cos, O = sum_odd s[n] cos, Ei = sum_even d[n] sin, Oi = sum_odd d[n] sin:
//   X[k]       = (x0 + (-1)^k xh + E + O) - i (Ei + Oi)
//   X[N/2 - k] = (x0 + (-1)^(N/2-k) xh + E - O) + i (Ei - Oi)
// one thread produces BOTH bins from one pass over the sample pairs: a quarter of the plain sum's FMAs and LDS reads.
template <int NFFT, int NB, int FSTR>
__device__ __forceinline__ void dft_power(const double* frames, const double* tw, double* power) {
  constexpr int HALF = NFFT / 2, QUART = NFFT / 4;
  static_assert(NFFT % 4 == 0, "bin pairing needs NFFT % 4 == 0");
  const int bin = threadIdx.x;                 // 0 .. NFFT/4; partner bin = NFFT/2 - bin
  if (bin <= QUART) {
    double e[FR], o[FR], ei[FR], oi[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) e[f] = o[f] = ei[f] = oi[f] = 0.0;
    int idx = bin;
    for (int n = 1; n < HALF; n += 2) {        // n odd, then n + 1 even
      {
        const double c = tw[2 * idx], s = tw[2 * idx + 1];
        const int nd = (NFFT - n < WLEN) ? NFFT - n : n;
#pragma unroll
        for (int f = 0; f < FR; ++f) {
          o[f] += frames[f * FSTR + n] * c;
          oi[f] += frames[f * FSTR + nd] * s;
        }
        idx += bin;
        if (idx >= NFFT) idx -= NFFT;
      }
      if (n + 1 < HALF) {
        const double c = tw[2 * idx], s = tw[2 * idx + 1];
        const int m = n + 1, nd = (NFFT - m < WLEN) ? NFFT - m : m;
#pragma unroll
        for (int f = 0; f < FR; ++f) {
          e[f] += frames[f * FSTR + m] * c;
          ei[f] += frames[f * FSTR + nd] * s;
        }
        idx += bin;
        if (idx >= NFFT) idx -= NFFT;
      }
    }
    const int pb = HALF - bin;                 // partner bin
    const double sg = (bin & 1) ? -1.0 : 1.0, sp = (pb & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const double x0 = frames[f * FSTR], xh = HALF < WLEN ? frames[f * FSTR + HALF] : 0.0;
      const double re = x0 + sg * xh + e[f] + o[f], im = ei[f] + oi[f];
      power[f * NB + bin] = re * re + im * im;
      if (pb < NB && pb != bin) {
        const double re2 = x0 + sp * xh + e[f] - o[f], im2 = ei[f] - oi[f];
        power[f * NB + pb] = re2 * re2 + im2 * im2;
      }
    }
  }
}

// [lo, hi) of the non-zero taps of every (triangular, contiguous) mel filter, found once per block: the projection then
// walks 5-30 bins per filter instead of all of them
__device__ __forceinline__ void mel_ranges(const double* mel, int n_mel, int row_ld, int n_bins, int* lo, int* hi) {
  for (int m = threadIdx.x; m < n_mel; m += blockDim.x) {
    const double* mf = mel + (int64_t)m * row_ld;
    int a = n_bins, b = 0;
    for (int k = 0; k < n_bins; ++k)
      if (mf[k] != 0.0) {
        a = min(a, k);
        b = k + 1;
      }
    lo[m] = min(a, b);
    hi[m] = b;
  }
}

template <int NFFT>
__device__ __forceinline__ void fill_twiddles(double* tw) {
  for (int k = threadIdx.x; k < NFFT; k += blockDim.x) {
    double s, c;
    sincospi(2.0 * (double)k / (double)NFFT, &s, &c);
    tw[2 * k] = c;
    tw[2 * k + 1] = s;
  }
}

// ---- K1 stage 1: STFT power -> mel -> log10, per-audio running max -----------------------------------
__global__ __launch_bounds__(256) void whisper_logmel_kernel(const float* wav, int64_t wav_ld, const int* wav_lens,
                                                              const double* mel, int n_mel, float* raw,
                                                              int* gmax) {
  __shared__ double frames[FR * WLEN];
  __shared__ double tw[2 * WH_NFFT];
  __shared__ double power[FR * WH_BINS];
  __shared__ int mlo[128], mhi[128];
  const int a = blockIdx.y, f0 = blockIdx.x * FR;
  const float* w = wav + (int64_t)a * wav_ld;
  const int L = min(wav_lens[a], WH_SAMPLES);
  fill_twiddles<WH_NFFT>(tw);
  mel_ranges(mel, n_mel, WH_BINS, WH_BINS, mlo, mhi);
  __syncthreads();
  // centre frames over the zero-padded 480000-sample signal with reflect padding of n_fft/2
  for (int i = threadIdx.x; i < FR * WLEN; i += blockDim.x) {
    const int f = i / WLEN, n = i - f * WLEN;
    int j = (f0 + f) * HOP + n - WH_NFFT / 2;
    if (j < 0) j = -j;
    if (j >= WH_SAMPLES) j = 2 * (WH_SAMPLES - 1) - j;
    const double x = (j < L) ? (double)w[j] : 0.0;
    frames[i] = x * (0.5 - 0.5 * tw[2 * n]);  // periodic Hann: cos(2 pi n / 400) is the twiddle
  }
  __syncthreads();
  fold_frames<WH_NFFT, WLEN>(frames);
  __syncthreads();
  dft_power<WH_NFFT, WH_BINS, WLEN>(frames, tw, power);
  __syncthreads();
  float local_max = -INFINITY;
  for (int o = threadIdx.x; o < n_mel * FR; o += blockDim.x) {
    const int m = o / FR, f = o - m * FR;
    if (f0 + f >= WH_FRAMES) continue;
    const double* mf = mel + (int64_t)m * WH_BINS;
    const double* pw = power + f * WH_BINS;
    double acc = 0.0;
    for (int b = mlo[m]; b < mhi[m]; ++b) acc += mf[b] * pw[b];
    const float lv = (float)log10(fmax(acc, 1e-10));
    raw[((int64_t)a * n_mel + m) * WH_FRAMES + f0 + f] = lv;
    local_max = fmaxf(local_max, lv);
  }
  local_max = wave_reduce_max(local_max);
  if ((threadIdx.x & 63) == 0 && local_max > -INFINITY) atomicMax(gmax + a, float_to_ordered(local_max));
}

// ---- K1 stage 2: clamp to max-8, (x+4)/4, write [mel][t] f32 and the time-major conv operand ---------
__global__ __launch_bounds__(256) void whisper_logmel_finish_kernel(const float* raw, const int* gmax, int n_mel,
                                                                     float* spec, unsigned short* xt,
                                                                     int64_t xt_ld) {
  __shared__ float tile[64][129];
  const int a = blockIdx.y, t0 = blockIdx.x * 64;
  const float floor_v = gmax ? ordered_to_float(gmax[a]) - 8.0f : -INFINITY;
  for (int i = threadIdx.x; i < n_mel * 64; i += blockDim.x) {
    const int m = i >> 6, t = i & 63;
    float v = 0.f;
    if (t0 + t < WH_FRAMES) {
      const int64_t off = ((int64_t)a * n_mel + m) * WH_FRAMES + t0 + t;
      v = raw[off];
      if (gmax) {
        v = (fmaxf(v, floor_v) + 4.0f) / 4.0f;
        if (spec) spec[off] = v;
      }
    }
    tile[t][m] = v;
  }
  __syncthreads();
  if (xt) {
    const int cols = (int)xt_ld;
    for (int i = threadIdx.x; i < 64 * cols; i += blockDim.x) {
      const int t = i / cols, c = i - t * cols;
      if (t0 + t >= WH_FRAMES) continue;
      const float v = c < n_mel ? tile[t][c] : 0.f;
      xt[((int64_t)a * (WH_FRAMES + 2) + 1 + t0 + t) * xt_ld + c] = f32_to_bf16_bits(v);
    }
    if (blockIdx.x == 0) {
      for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        xt[((int64_t)a * (WH_FRAMES + 2)) * xt_ld + c] = 0;
        xt[((int64_t)a * (WH_FRAMES + 2) + WH_FRAMES + 1) * xt_ld + c] = 0;
      }
    }
  }
}

// ---- K4: Kaldi fbank ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kaldi_fbank_kernel(const float* wav, int64_t wav_ld, const int* wav_lens,
                                                           const double* mel, int max_frames, float mean,
                                                           float stdv, float* out) {
  __shared__ double frames[FR * WLEN];
  __shared__ double tw[2 * FB_NFFT];
  __shared__ double power[FR * FB_BINS];
  __shared__ double fmean[FR];
  __shared__ int mlo[FB_MEL], mhi[FB_MEL];
  const int a = blockIdx.y, f0 = blockIdx.x * FR;
  const float* w = wav + (int64_t)a * wav_ld;
  const int L = wav_lens[a];
  const int n_frames = L >= WLEN ? 1 + (L - WLEN) / HOP : 0;
  if (f0 >= min(n_frames, max_frames)) return;
  fill_twiddles<FB_NFFT>(tw);
  mel_ranges(mel, FB_MEL, FB_BINS + 1, FB_BINS, mlo, mhi);
  // raw frames (x * 2^15), snip_edges framing
  for (int i = threadIdx.x; i < FR * WLEN; i += blockDim.x) {
    const int f = i / WLEN, n = i - f * WLEN;
    const int fr = min(f0 + f, n_frames - 1);
    frames[i] = (double)w[(int64_t)fr * HOP + n] * 32768.0;
  }
  __syncthreads();
  {  // DC offset: 32 threads per frame
    const int f = threadIdx.x >> 5, l = threadIdx.x & 31;
    double s = 0.0;
    for (int n = l; n < WLEN; n += 32) s += frames[f * WLEN + n];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l == 0) fmean[f] = s / (double)WLEN;
  }
  __syncthreads();
  // pre-emphasis needs the un-emphasised neighbour: compute into registers, then write back
  {
    constexpr int NV = (FR * WLEN + 255) / 256;
    double vals[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i = threadIdx.x + c * 256;
      vals[c] = 0.0;
      if (i < FR * WLEN) {
        const int f = i / WLEN, n = i - f * WLEN;
        const double cur = frames[i] - fmean[f];
        const double prev = frames[n > 0 ? i - 1 : i] - fmean[f];
        const double win = pow(0.5 - 0.5 * cospi(2.0 * (double)n / (double)(WLEN - 1)), 0.85);  // povey
        vals[c] = (cur - 0.97 * prev) * win;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i = threadIdx.x + c * 256;
      if (i < FR * WLEN) frames[i] = vals[c];
    }
  }
  __syncthreads();
  fold_frames<FB_NFFT, WLEN>(frames);
  __syncthreads();
  dft_power<FB_NFFT, FB_BINS, WLEN>(frames, tw, power);
  __syncthreads();
  for (int o = threadIdx.x; o < FB_MEL * FR; o += blockDim.x) {
    const int f = o / FB_MEL, m = o - f * FB_MEL;
    if (f0 + f >= n_frames || f0 + f >= max_frames) continue;
    const double* mf = mel + (int64_t)m * (FB_BINS + 1);
    const double* pw = power + f * FB_BINS;
    double acc = 0.0;
    for (int b = mlo[m]; b < mhi[m]; ++b) acc += mf[b] * pw[b];
    const double lv = log(fmax(acc, 1.1920928955078125e-07));
    out[((int64_t)a * max_frames + f0 + f) * FB_MEL + m] = (float)((lv - (double)mean) / (2.0 * (double)stdv));
  }
}

}  // namespace

extern "C" int icl_logmel_whisper(const float* wav, int64_t wav_ld, const int32_t* wav_lens,
                                  const double* mel_filters, int32_t n_mel, int32_t n_audio, float* spec,
                                  void* xt, int64_t xt_ld, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ICL_CHECK_ARG(wav && wav_lens && mel_filters && workspace, "icl_logmel_whisper: NULL pointer");
  ICL_CHECK_ARG(n_mel > 0 && n_mel <= 128 && n_audio > 0 && n_audio <= 65535, "icl_logmel_whisper: n_mel=%d n_audio=%d out of range", n_mel, n_audio);
  ICL_CHECK_ARG(spec || xt, "icl_logmel_whisper: at least one of spec / xt must be given");
  if (xt) ICL_CHECK_ARG(xt_ld >= n_mel && xt_ld <= 128 && xt_ld % 8 == 0, "icl_logmel_whisper: xt_ld=%lld must be in [n_mel,128] and a multiple of 8", (long long)xt_ld);
  float* raw = (float*)workspace;
  int* gmax = (int*)(raw + (int64_t)n_audio * n_mel * WH_FRAMES);
  hipError_t e = hipMemsetD32Async((hipDeviceptr_t)gmax, (int)0x80000000, n_audio, stream);
  if (e != hipSuccess) {
    icl_set_error("icl_logmel_whisper: memset failed: %s", hipGetErrorString(e));
    return ICL_ELAUNCH;
  }
  hipLaunchKernelGGL(whisper_logmel_kernel, dim3((WH_FRAMES + FR - 1) / FR, n_audio), dim3(256), 0, stream, wav,
                     wav_ld, wav_lens, mel_filters, n_mel, raw, gmax);
  ICL_CHECK_LAUNCH("icl_logmel_whisper(stft)");
  hipLaunchKernelGGL(whisper_logmel_finish_kernel, dim3((WH_FRAMES + 63) / 64, n_audio), dim3(256), 0, stream,
                     (const float*)raw, (const int*)gmax, n_mel, spec, (unsigned short*)xt, xt_ld);
  ICL_CHECK_LAUNCH("icl_logmel_whisper(finish)");
  return ICL_OK;
}

extern "C" int icl_spec_to_xt(const float* spec, int32_t n_mel, int32_t n_audio, void* xt, int64_t xt_ld,
                              void* stream) {
  ICL_CHECK_ARG(spec && xt, "icl_spec_to_xt: NULL pointer");
  ICL_CHECK_ARG(n_mel > 0 && n_mel <= 128 && n_audio > 0 && n_audio <= 65535, "icl_spec_to_xt: bad sizes");
  ICL_CHECK_ARG(xt_ld >= n_mel && xt_ld <= 128 && xt_ld % 8 == 0, "icl_spec_to_xt: bad xt_ld");
  hipLaunchKernelGGL(whisper_logmel_finish_kernel, dim3((WH_FRAMES + 63) / 64, n_audio), dim3(256), 0,
                     (hipStream_t)stream, spec, (const int*)nullptr, n_mel, (float*)nullptr, (unsigned short*)xt, xt_ld);
  ICL_CHECK_LAUNCH("icl_spec_to_xt");
  return ICL_OK;
}

extern "C" int icl_fbank_kaldi(const float* wav, int64_t wav_ld, const int32_t* wav_lens, const double* mel_banks,
                               int32_t n_audio, int32_t max_frames, float mean, float stdv, float* fbank,
                               void* stream) {
  ICL_CHECK_ARG(wav && wav_lens && mel_banks && fbank, "icl_fbank_kaldi: NULL pointer");
  ICL_CHECK_ARG(n_audio > 0 && n_audio <= 65535 && max_frames > 0, "icl_fbank_kaldi: bad sizes");
  ICL_CHECK_ARG(stdv > 0.f, "icl_fbank_kaldi: std must be > 0");
  hipLaunchKernelGGL(kaldi_fbank_kernel, dim3((max_frames + FR - 1) / FR, n_audio), dim3(256), 0,
                     (hipStream_t)stream, wav, wav_ld, wav_lens, mel_banks, max_frames, mean, stdv, fbank);
  ICL_CHECK_LAUNCH("icl_fbank_kaldi");
  return ICL_OK;
}
