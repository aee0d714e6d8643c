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
// 32 lanes per patch row, 8 consecutive j per lane (two 16-B f32 loads -> ONE 16-B bf16 store), 8 patch rows per block; the
// block finds the audio of its first row by bisection and walks forward from there (rows of a block are consecutive).
// Round 3: the first form moved one element per thread (a 2-byte store each, one block and one bisection per 512-byte row):
// 4.3 ms per 256 clips = 138 GB/s.
__global__ __launch_bounds__(256) void beats_patchify_kernel(const float* fbank, int max_frames, const int* cu,
                                                              int n_audio, int total_rows, unsigned short* out) {
  const int row0 = blockIdx.x * 8;
  const int row = row0 + (threadIdx.x >> 5);
  if (row >= total_rows) return;
  int a = audio_of(cu, n_audio, row0, 0);
  while (a + 1 < n_audio && cu[a + 1] <= row) ++a;      // at most a few steps: rows of a block are consecutive
  const int local = row - cu[a];
  const int tp = local >> 3, fp = local & 7;
  const int l = threadIdx.x & 31, i = l >> 1, j0 = (l & 1) * 8;
  const float* src = fbank + ((int64_t)a * max_frames + tp * 16 + i) * 128 + fp * 16 + j0;
  const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
  *(u32x4*)(out + (int64_t)row * 256 + l * 8) =
      u32x4{pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
}

// x f32 [M][C] packed by cu; rows >= valid[a] of audio a are zeroed IN PLACE (x[padding_mask] = 0),
// and xg (bf16) receives, per audio, the image [G groups][T_a + 128][C/G] with 64 zero rows in front and behind.
// One lane per 8 consecutive channels of one padded row (C/G is a multiple of 8, so a piece never crosses a group): two 16-B
// f32 loads -> one 16-B bf16 store; a block takes 256 / (C/8) consecutive padded rows.  Round 3: the first form moved one
// element per thread-iteration with 2-byte stores, one block per row: 6.6 ms per 256 clips = 270 GB/s.
__global__ __launch_bounds__(256) void beats_posconv_pack_kernel(float* x, const int* cu, const int* valid,
                                                                  int n_audio, int C, int G, unsigned short* xg) {
  const int lanes_per_row = C >> 3, rows_per_block = 256 / lanes_per_row;
  const int sub = threadIdx.x / lanes_per_row;
  if (sub >= rows_per_block) return;
  const int total = cu[n_audio] + 128 * n_audio;
  const int prow0 = blockIdx.x * rows_per_block, prow = prow0 + sub;
  if (prow >= total) return;
  int a = audio_of(cu, n_audio, prow0, 128);
  while (a + 1 < n_audio && cu[a + 1] + 128 * (a + 1) <= prow) ++a;
  const int r = prow - (cu[a] + 128 * a);
  const int T = cu[a + 1] - cu[a];
  const int t = r - 64;
  const bool in_range = (t >= 0) && (t < T);
  const bool live = in_range && (t < valid[a]);
  const int cpg = C / G;
  const int c = (threadIdx.x - sub * lanes_per_row) * 8;
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
  if (in_range) {
    float* px = x + ((int64_t)cu[a] + t) * C + c;
    if (live) {
      v0 = *(const f32x4*)px;
      v1 = *(const f32x4*)(px + 4);
    } else {
      *(f32x4*)px = v0;
      *(f32x4*)(px + 4) = v0;
    }
  }
  const int g = c / cpg, ci = c - g * cpg;
  const int64_t base = ((int64_t)cu[a] + 128 * (int64_t)a) * C;  // start of this audio's image
  *(u32x4*)(xg + base + ((int64_t)g * (T + 128) + r) * cpg + ci) =
      u32x4{pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
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
  ICL_CHECK_ARG(((uintptr_t)fbank & 15) == 0 && ((uintptr_t)out & 15) == 0, "icl_beats_patchify: misaligned pointer");
  hipLaunchKernelGGL(beats_patchify_kernel, dim3((total_rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, fbank, max_frames,
                     cu_rows, n_audio, total_rows, (unsigned short*)out);
  ICL_CHECK_LAUNCH("icl_beats_patchify");
  return ICL_OK;
}

extern "C" int icl_beats_posconv_pack(float* x, const int32_t* cu_rows, const int32_t* valid_rows, int32_t n_audio,
                                      int32_t total_rows, int32_t channels, int32_t groups, void* xg, void* stream) {
  ICL_CHECK_ARG(x && cu_rows && valid_rows && xg && n_audio > 0 && total_rows > 0, "icl_beats_posconv_pack: bad arguments");
  ICL_CHECK_ARG(groups > 0 && channels % groups == 0 && (channels / groups) % 8 == 0,
                "icl_beats_posconv_pack: channels/groups must be a multiple of 8 (got %d/%d)", channels, groups);
  ICL_CHECK_ARG(channels % 8 == 0 && channels <= 2048 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)xg & 15) == 0,
                "icl_beats_posconv_pack: channels must be a multiple of 8 (<= 2048), pointers 16-byte aligned");
  const int rows_per_block = 256 / (channels / 8);
  hipLaunchKernelGGL(beats_posconv_pack_kernel, dim3((total_rows + 128 * n_audio + rows_per_block - 1) / rows_per_block), dim3(256),
                     0, (hipStream_t)stream, x, cu_rows, valid_rows, n_audio, channels, groups, (unsigned short*)xg);
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
