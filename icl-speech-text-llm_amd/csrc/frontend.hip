// frontend.hip — audio front-ends of the ICL path on gfx950:
//   K1  Whisper log-mel   (WhisperFeatureExtractor semantics; data/model_processors.py:641-645)
//   K4  BEATs Kaldi fbank (torchaudio.compliance.kaldi.fbank semantics as used by BEATs.preprocess)
// Both are tiny (≈1 GFLOP per 30 s clip) and are computed in f64 (direct DFT from an LDS twiddle table: n_fft = 400 is not a
// power of two), so the result is closer to the exact value than the reference's f32 FFT; outputs are f32.
// Layout: one block walks groups of 16 consecutive frames of one audio.  The samples the 16 frames cover are staged once in
// LDS (f32, exact); the transform uses the real-input fold s[n] = x[n] + x[N-n], d[n] = x[n] - x[N-n] and the bin pairing
// (k, N/2 - k) — a quarter of the plain DFT's FMAs — and runs as four small GEMMs on the f64 matrix pipe
// (dft_power_mfma below; round 1 ran the same sums on the vector pipe, bound by one LDS read per FMA pair: 3.7 / 4.6 ms for
// 128 clips against 1.5 / 2.8 ms now).  Loads of the waveform are coalesced; nothing is re-read from HBM.
#include "common.h"

namespace {

constexpr int WLEN = 400;        // window length (both front-ends)
constexpr int HOP = 160;
constexpr int WH_NFFT = 400, WH_BINS = 201, WH_FRAMES = 3000, WH_SAMPLES = 480000;
constexpr int FB_NFFT = 512, FB_BINS = 256, FB_MEL = 128;  // Nyquist bin has a zero mel column
constexpr int WH_GROUPS = 4, FB_GROUPS = 4;     // groups of 16 frames per block (Whisper: 3000 / 16 = 188 groups = 47 blocks x 4 per clip)

__device__ __forceinline__ int float_to_ordered(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// [lo, hi) of the non-zero taps of every (triangular, contiguous) mel filter, found once per block: the projection then
// walks 5-30 bins per filter instead of all of them.  The whole block scans the filter matrix with independent, coalesced
// loads and LDS atomics on the few non-zeros (a thread walking one row alone is a chain of L2 round trips: 200-256 of them
// per block used to cost more than the transform).
__device__ __forceinline__ void mel_ranges(const double* mel, int n_mel, int row_ld, int n_bins, int* lo, int* hi) {
  for (int m = threadIdx.x; m < n_mel; m += blockDim.x) {
    lo[m] = n_bins;
    hi[m] = 0;
  }
  __syncthreads();
  const int total = n_mel * n_bins;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int m = i / n_bins, k = i - m * n_bins;
    if (mel[(int64_t)m * row_ld + k] != 0.0) {
      atomicMin(&lo[m], k);
      atomicMax(&hi[m], k + 1);
    }
  }
  __syncthreads();
  for (int m = threadIdx.x; m < n_mel; m += blockDim.x) lo[m] = min(lo[m], hi[m]);
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

// The non-zero taps of all filters packed back to back in LDS (a bin feeds at most two triangular filters, so about 2 x bins
// doubles), offset table moff[]: the projection's inner loop then reads LDS instead of walking a global row, one dependent L2
// load per tap.  Filter banks whose taps do not fit MEL_TAPS stay in global memory (moff[0] = -1).
constexpr int MEL_TAPS = 640;
__device__ __forceinline__ void mel_pack(const double* mel, int n_mel, int row_ld, const int* lo, const int* hi, int* moff,
                                         double* taps) {
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int m = 0; m < n_mel; ++m) {
      moff[m] = tot;
      tot += hi[m] - lo[m];
    }
    if (tot > MEL_TAPS) moff[0] = -1;
  }
  __syncthreads();
  if (moff[0] < 0) return;
  for (int m = threadIdx.x; m < n_mel; m += blockDim.x) {
    const double* mf = mel + (int64_t)m * row_ld;
    for (int b = lo[m]; b < hi[m]; ++b) taps[moff[m] + b - lo[m]] = mf[b];
  }
}
__device__ __forceinline__ double mel_dot(const double* mel, int row_ld, int m, const int* lo, const int* hi, const int* moff,
                                          const double* taps, const double* pw) {
  double acc = 0.0;
  if (moff[0] >= 0) {
    const double* tp = taps + moff[m] - lo[m];
    for (int b = lo[m]; b < hi[m]; ++b) acc += tp[b] * pw[b];
  } else {
    const double* mf = mel + (int64_t)m * row_ld;
    for (int b = lo[m]; b < hi[m]; ++b) acc += mf[b] * pw[b];
  }
  return acc;
}

// ---- DFT power of MF = 16 frames on the f64 matrix pipe ----------------------------------------------------------------
// The same folded sums as dft_power() written as four small GEMMs, frames x samples x bins, on v_mfma_f64_16x16x4_f64:
//   E = S_even C_even,  O = S_odd C_odd,  Ei = D_even Sn_even,  Oi = D_odd Sn_odd      (bins k = 0 .. N/4)
//   X[k] = (E + O) - i (Ei + Oi),   X[N/2 - k] = (E - O) + i (Ei - Oi)
// with s[n] = x[n] + x[N-n], d[n] = x[n] - x[N-n] for 0 < n < N/2, s[0] = x[0], s[N/2] = x[N/2], d[0] = d[N/2] = 0 (so the
// DC and Nyquist samples ride in the even sum).  Round 1's vector form read LDS once per FMA pair and was bound by that
// (11-15 % of the f64 FMA rate); here one 8-byte operand per lane feeds 1024 FMAs and the kernels are bound by the matrix
// pipe itself: v_mfma_f64_16x16x4_f64 measures ~128 cycles per instruction in these loops (16 FLOP/clk/SIMD, half the f64
// VECTOR rate), so a register-blocked vector kernel could in principle go further.  Operand maps (guide, "MFMA layouts"):
// A[frame = lane & 15][sample slot = lane >> 4], B[slot = lane >> 4][bin = lane & 15], C/D col = lane & 15 (bin),
// row = (lane >> 4) + 4 r (frame).  A lane builds its A operands itself from the staged samples through `xw(frame, n)`
// (windowed sample in f64, 0 outside the window), so no folded copy of the frames lives in LDS; twiddles start from the
// cos/sin table and advance by rotation.  Wave w owns bin tiles w*NT .. w*NT + NT-1.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int MF = 16;
template <int NFFT, int NB, int NT, bool POWER_ALIASES_FRAMES = false, class XW>
__device__ __forceinline__ void dft_power_mfma(const double* tw, double* power, XW xw) {
  constexpr int HALF = NFFT / 2, QUART = NFFT / 4;
  constexpr int KE = (QUART + 1 + 3) / 4;       // K-steps: even n = 0, 2 .. N/2 (N/4 + 1 samples); odd n = 1, 3 .. N/2 - 1 fit in them
  static_assert(NFFT % 8 == 0, "even / odd split of the folded samples");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = lane & 15, q = lane >> 4;
  f64x4 E[NT], O[NT], Ei[NT], Oi[NT];
  // twiddles of this lane's bin for its current even / odd sample, advanced by ROTATION (angle 2 pi 8k / N per K-step, four
  // f64 FMAs each) instead of a table read per step: lanes of consecutive bins hit the table with stride n — bank conflicts
  // that made the LDS, not the matrix pipe, the limit.  33 rotations add ~4e-15 of relative error (outputs are f32).
  double ce[NT], sne[NT], co[NT], sno[NT], cd[NT], sd[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    E[t] = O[t] = Ei[t] = Oi[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int k = min((wave * NT + t) * 16 + f, QUART);      // padding bins repeat bin N/4; their results are not stored
    const int ie = (k * (2 * q)) % NFFT, io = (k * (2 * q + 1)) % NFFT, inc = (k * 8) % NFFT;
    ce[t] = tw[2 * ie];
    sne[t] = tw[2 * ie + 1];
    co[t] = tw[2 * io];
    sno[t] = tw[2 * io + 1];
    cd[t] = tw[2 * inc];
    sd[t] = tw[2 * inc + 1];
  }
  for (int j = 0; j < KE; ++j) {
    // branch-free (selects on clamped indices): q differs between lanes, a branch here is a divergent one around LDS reads
    const int ne = 2 * (4 * j + q), no = ne + 1;
    const int nec = min(ne, HALF), noc = min(no, HALF - 1);
    const double ae = xw(f, nec), be = xw(f, nec == 0 ? 0 : NFFT - nec);
    const double ao = xw(f, noc), bo = xw(f, NFFT - noc);
    const bool mid = ne > 0 && ne < HALF, edge = ne == 0 || ne == HALF, odd = no < HALF;
    const double se = mid ? ae + be : (edge ? ae : 0.0), de = mid ? ae - be : 0.0;
    const double so = odd ? ao + bo : 0.0, dd = odd ? ao - bo : 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      E[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(se, ce[t], E[t], 0, 0, 0);
      Ei[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(de, sne[t], Ei[t], 0, 0, 0);
      O[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(so, co[t], O[t], 0, 0, 0);
      Oi[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(dd, sno[t], Oi[t], 0, 0, 0);
      const double c1 = ce[t] * cd[t] - sne[t] * sd[t], s1 = sne[t] * cd[t] + ce[t] * sd[t];
      const double c2 = co[t] * cd[t] - sno[t] * sd[t], s2 = sno[t] * cd[t] + co[t] * sd[t];
      ce[t] = c1;
      sne[t] = s1;
      co[t] = c2;
      sno[t] = s2;
    }
  }
  if (POWER_ALIASES_FRAMES) __syncthreads();    // every wave is done reading the frames `power` is about to overwrite
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = (wave * NT + t) * 16 + f;
    if (k > QUART) continue;
    const int pb = HALF - k;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int fr = q + 4 * r;
      const double re = E[t][r] + O[t][r], im = Ei[t][r] + Oi[t][r];
      power[fr * NB + k] = re * re + im * im;
      if (pb < NB && pb != k) {
        const double re2 = E[t][r] - O[t][r], im2 = Ei[t][r] - Oi[t][r];
        power[fr * NB + pb] = re2 * re2 + im2 * im2;
      }
    }
  }
}

// (launch bounds with a minimum of 2 waves per SIMD on both transform kernels: with the whole 512-register file on offer hipcc
// puts the MFMA accumulators in AGPRs but carries them through the loop in VGPRs — 96 copies in and 96 out per K-step)
// ---- K1 stage 1: STFT power -> mel -> log10, per-audio running max -----------------------------------
__global__ __launch_bounds__(256, 2) void whisper_logmel_kernel(const float* wav, int64_t wav_ld, const int* wav_lens,
                                                              const double* mel, int n_mel, float* raw,
                                                              int* gmax) {
  constexpr int SPAN = (MF - 1) * HOP + WLEN;        // samples the block's MF frames cover (hop 160, window 400)
  __shared__ float span[SPAN];
  __shared__ double tw[2 * WH_NFFT];
  __shared__ double power[MF * WH_BINS];
  __shared__ double taps[MEL_TAPS];
  __shared__ int mlo[128], mhi[128], moff[128];
  const int a = blockIdx.y;
  const float* w = wav + (int64_t)a * wav_ld;
  const int L = min(wav_lens[a], WH_SAMPLES);
  // the twiddle table and the mel tap ranges are per-block set-up (400 f64 sincospi, a scan of the 80 x 201 filter matrix):
  // one block walks WH_GROUPS groups of MF frames so that it is paid once per WH_GROUPS * MF frames
  fill_twiddles<WH_NFFT>(tw);
  mel_ranges(mel, n_mel, WH_BINS, WH_BINS, mlo, mhi);
  __syncthreads();
  mel_pack(mel, n_mel, WH_BINS, mlo, mhi, moff, taps);
  float local_max = -INFINITY;
  for (int g = 0; g < WH_GROUPS; ++g) {
    const int f0 = (blockIdx.x * WH_GROUPS + g) * MF;
    if (f0 >= WH_FRAMES) break;
    __syncthreads();                       // set-up visible (g = 0) / the previous group's span and power are consumed
    // centre frames over the zero-padded 480000-sample signal with reflect padding of n_fft/2: sample i of the span is signal
    // index f0 * hop - 200 + i (f32 staging is exact; the arithmetic is f64)
    for (int i = threadIdx.x; i < SPAN; i += blockDim.x) {
      int j = f0 * HOP + i - WH_NFFT / 2;
      if (j < 0) j = -j;
      if (j >= WH_SAMPLES) j = 2 * (WH_SAMPLES - 1) - j;
      span[i] = (j >= 0 && j < L) ? w[j] : 0.f;
    }
    __syncthreads();
    // periodic Hann: cos(2 pi n / 400) is the twiddle
    dft_power_mfma<WH_NFFT, WH_BINS, 2, false>(tw, power, [&](int f, int n) { return (double)span[f * HOP + n] * (0.5 - 0.5 * tw[2 * n]); });
    __syncthreads();
    for (int o = threadIdx.x; o < n_mel * MF; o += blockDim.x) {
      const int m = o / MF, f = o - m * MF;
      if (f0 + f >= WH_FRAMES) continue;
      const double acc = mel_dot(mel, WH_BINS, m, mlo, mhi, moff, taps, power + f * WH_BINS);
      const float lv = (float)log10(fmax(acc, 1e-10));
      raw[((int64_t)a * n_mel + m) * WH_FRAMES + f0 + f] = lv;
      local_max = fmaxf(local_max, lv);
    }
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
__global__ __launch_bounds__(192, 2) void kaldi_fbank_kernel(const float* wav, int64_t wav_ld, const int* wav_lens,
                                                              const double* mel, int max_frames, float mean,
                                                              float stdv, float* out) {
  // 3 waves x 3 bin tiles of 16 = the 129 bin pairs of the 512-point transform; MF = 16 frames per group
  constexpr int XP = WLEN + 1;               // row pitch of the windowed frames (odd: the 16 frames of an A operand spread over the banks)
  __shared__ double big[MF * XP];            // windowed frames during the transform, then (behind a barrier) the power spectrum
  __shared__ double tw[2 * FB_NFFT];
  __shared__ double povey[WLEN];
  __shared__ double fmean[MF];
  __shared__ double taps[MEL_TAPS];
  __shared__ int mlo[FB_MEL], mhi[FB_MEL], moff[FB_MEL];
  static_assert(MF * XP >= MF * FB_BINS, "the power spectrum reuses the frame buffer");
  double* xwin = big;
  double* power = big;
  const int a = blockIdx.y;
  const float* w = wav + (int64_t)a * wav_ld;
  const int L = wav_lens[a];
  const int n_frames = L >= WLEN ? 1 + (L - WLEN) / HOP : 0;
  const int f_end = min(n_frames, max_frames);
  if (blockIdx.x * FB_GROUPS * MF >= f_end) return;
  // per-block set-up, paid once per FB_GROUPS * MF frames: twiddles, povey window, mel tap ranges + packed taps
  fill_twiddles<FB_NFFT>(tw);
  for (int n = threadIdx.x; n < WLEN; n += blockDim.x)
    povey[n] = pow(0.5 - 0.5 * cospi(2.0 * (double)n / (double)(WLEN - 1)), 0.85);
  mel_ranges(mel, FB_MEL, FB_BINS + 1, FB_BINS, mlo, mhi);
  __syncthreads();
  mel_pack(mel, FB_MEL, FB_BINS + 1, mlo, mhi, moff, taps);
  for (int g = 0; g < FB_GROUPS; ++g) {
    const int f0 = (blockIdx.x * FB_GROUPS + g) * MF;
    if (f0 >= f_end) break;                  // block-uniform
    __syncthreads();                         // set-up visible / the previous group's power spectrum is consumed
    // snip_edges framing: frame f starts at sample f * hop; frames past the last one repeat it (their rows are not stored)
    {  // DC offset of x * 2^15: 8 threads per frame, fixed order
      const int f = threadIdx.x >> 3, l = threadIdx.x & 7;
      if (f < MF) {
        const float* fw = w + (int64_t)min(f0 + f, n_frames - 1) * HOP;
        double sacc = 0.0;
        for (int n = l; n < WLEN; n += 8) sacc += (double)fw[n] * 32768.0;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
        if (l == 0) fmean[f] = sacc / (double)WLEN;
      }
    }
    __syncthreads();
    // windowed frames, once per group for all three waves: remove the DC offset, pre-emphasise against the un-emphasised
    // neighbour (sample 0 against itself), povey window
    for (int i = threadIdx.x; i < MF * WLEN; i += blockDim.x) {
      const int f = i / WLEN, n = i - f * WLEN;
      const float* fw = w + (int64_t)min(f0 + f, n_frames - 1) * HOP;
      const double mu = fmean[f];
      const double cur = (double)fw[n] * 32768.0 - mu;
      const double prev = (double)fw[max(n - 1, 0)] * 32768.0 - mu;
      xwin[f * XP + n] = (cur - 0.97 * prev) * povey[n];
    }
    __syncthreads();
    // (the 512-point transform's samples 400 .. 511 are zero padding)
    dft_power_mfma<FB_NFFT, FB_BINS, 3, true>(tw, power, [&](int f, int n) -> double {
      const double v = xwin[f * XP + min(n, WLEN - 1)];
      return n < WLEN ? v : 0.0;
    });
    __syncthreads();
    for (int o = threadIdx.x; o < FB_MEL * MF; o += blockDim.x) {
      const int f = o / FB_MEL, m = o - f * FB_MEL;
      if (f0 + f >= f_end) continue;
      const double acc = mel_dot(mel, FB_BINS + 1, m, mlo, mhi, moff, taps, power + f * FB_BINS);
      const double lv = log(fmax(acc, 1.1920928955078125e-07));
      out[((int64_t)a * max_frames + f0 + f) * FB_MEL + m] = (float)((lv - (double)mean) / (2.0 * (double)stdv));
    }
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
  hipLaunchKernelGGL(whisper_logmel_kernel, dim3((WH_FRAMES + MF * WH_GROUPS - 1) / (MF * WH_GROUPS), n_audio), dim3(256), 0, stream, wav,
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
  hipLaunchKernelGGL(kaldi_fbank_kernel, dim3((max_frames + MF * FB_GROUPS - 1) / (MF * FB_GROUPS), n_audio), dim3(192), 0,
                     (hipStream_t)stream, wav, wav_ld, wav_lens, mel_banks, max_frames, mean, stdv, fbank);
  ICL_CHECK_LAUNCH("icl_fbank_kaldi");
  return ICL_OK;
}
