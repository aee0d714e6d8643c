// beats.hip — data-movement kernels of the BEATs front of the path (K4/K5): patch im2col for the
// 16x16/stride-16 Conv2d, and the zero-padded per-group operand of the grouped positional Conv1d
// (k=128, groups=16), so that both convolutions run on the MFMA GEMM with an affine row view
// (no gather in the GEMM).  HBM-bound byte movers; coalesced 4-16 B per lane.
#include "common.h"

namespace {

// The audio a packed row belongs to: the largest a with cu[a] + pad * a <= row, by bisection (every block of these kernels
// starts with it; walking the prefix sums one dependent global load at a time cost ~50 us per block at 256 clips —
// 6.6 ms of the 7 the pos-conv pack took).
__device__ __forceinline__ int audio_of(const int* cu, int n_audio, int row, int pad) {
  int lo = 0, hi = n_audio - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cu[mid] + pad * mid <= row) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// fbank f32 [n_audio][max_frames][128] -> patches bf16 [cu[a] + t'*8 + f'][i*16 + j] = fbank[a][16t'+i][16f'+j]
__global__ __launch_bounds__(256) void beats_patchify_kernel(const float* fbank, int max_frames, const int* cu,
                                                              int n_audio, unsigned short* out) {
  const int row = blockIdx.x;  // packed patch row
  const int a = audio_of(cu, n_audio, row, 0);
  const int local = row - cu[a];
  const int tp = local >> 3, fp = local & 7;
  const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  const float v = fbank[((int64_t)a * max_frames + tp * 16 + i) * 128 + fp * 16 + j];
  out[(int64_t)row * 256 + threadIdx.x] = f32_to_bf16_bits(v);
}

// x f32 [M][768] packed by cu; rows >= valid[a] of audio a are zeroed IN PLACE (x[padding_mask] = 0),
// and xg (bf16) receives, per audio, the image [16 groups][T_a + 128][48] with 64 zero rows in front.
__global__ __launch_bounds__(256) void beats_posconv_pack_kernel(float* x, const int* cu, const int* valid,
                                                                  int n_audio, int C, int G, unsigned short* xg) {
  // one block per padded row index of one audio: blockIdx.x enumerates (audio, r) over sum(T_a + 128)
  if ((int)blockIdx.x >= cu[n_audio] + 128 * n_audio) return;
  const int a = audio_of(cu, n_audio, blockIdx.x, 128);
  const int r = blockIdx.x - (cu[a] + 128 * a);
  const int T = cu[a + 1] - cu[a];
  const int t = r - 64;
  const bool live = (t >= 0) && (t < T) && (t < valid[a]);
  const int cpg = C / G;
  const int64_t base = ((int64_t)cu[a] + 128 * (int64_t)a) * C;  // start of this audio's image
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = 0.f;
    if (t >= 0 && t < T) {
      float* px = x + ((int64_t)cu[a] + t) * C + c;
      if (live) v = *px; else *px = 0.f;
    }
    const int g = c / cpg, ci = c - g * cpg;
    xg[base + ((int64_t)g * (T + 128) + r) * cpg + ci] = f32_to_bf16_bits(v);
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* src, int64_t lds_, const int* idx, float* out,
                                                           int64_t ldo, int N) {
  const int64_t r = blockIdx.x;
  const float* s = src + (int64_t)idx[r] * lds_;
  float* d = out + r * ldo;
  for (int c = threadIdx.x * 4; c < N; c += blockDim.x * 4) *(f32x4*)(d + c) = *(const f32x4*)(s + c);
}

}  // namespace

extern "C" int icl_beats_patchify(const float* fbank, int32_t max_frames, const int32_t* cu_rows, int32_t n_audio,
                                  int32_t total_rows, void* out, void* stream) {
  ICL_CHECK_ARG(fbank && cu_rows && out && n_audio > 0 && total_rows > 0, "icl_beats_patchify: bad arguments");
  hipLaunchKernelGGL(beats_patchify_kernel, dim3(total_rows), dim3(256), 0, (hipStream_t)stream, fbank, max_frames,
                     cu_rows, n_audio, (unsigned short*)out);
  ICL_CHECK_LAUNCH("icl_beats_patchify");
  return ICL_OK;
}

extern "C" int icl_beats_posconv_pack(float* x, const int32_t* cu_rows, const int32_t* valid_rows, int32_t n_audio,
                                      int32_t total_rows, int32_t channels, int32_t groups, void* xg, void* stream) {
  ICL_CHECK_ARG(x && cu_rows && valid_rows && xg && n_audio > 0 && total_rows > 0, "icl_beats_posconv_pack: bad arguments");
  ICL_CHECK_ARG(groups > 0 && channels % groups == 0 && (channels / groups) % 8 == 0,
                "icl_beats_posconv_pack: channels/groups must be a multiple of 8 (got %d/%d)", channels, groups);
  hipLaunchKernelGGL(beats_posconv_pack_kernel, dim3(total_rows + 128 * n_audio), dim3(256), 0, (hipStream_t)stream, x,
                     cu_rows, valid_rows, n_audio, channels, groups, (unsigned short*)xg);
  ICL_CHECK_LAUNCH("icl_beats_posconv_pack");
  return ICL_OK;
}

extern "C" int icl_gather_rows_f32(const float* src, int64_t ld_src, const int32_t* idx, float* out, int64_t ld_out,
                                   int32_t rows, int32_t N, void* stream) {
  ICL_CHECK_ARG(src && idx && out && rows > 0 && N > 0 && N % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0,
                "icl_gather_rows_f32: bad arguments");
  ICL_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)out & 15) == 0, "icl_gather_rows_f32: misaligned");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx, out, ld_out, N);
  ICL_CHECK_LAUNCH("icl_gather_rows_f32");
  return ICL_OK;
}
