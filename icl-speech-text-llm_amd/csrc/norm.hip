// norm.hip — LayerNorm / RMSNorm for gfx950 (HBM-bound; one 256-thread block per row).
// The row lives in registers between the statistics passes (no re-read): each thread owns up to
// MAXV 4-element vectors (16 B f32 / 8 B bf16 loads, cdna_hip_programming.md Guideline 13).
// Statistics are two-pass in f32 (mean, then sum of squared deviations), matching torch's
// LayerNorm numerics more closely than E[x^2]-mean^2.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXV = 8;  // N <= NT*4*MAXV = 8192

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // protect red[] reuse between consecutive reductions
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ f32x4 load4(const void* base, int64_t off, int dtype) {
  if (dtype == ICL_F32) return *(const f32x4*)((const float*)base + off);
  const u32x2 raw = *(const u32x2*)((const unsigned short*)base + off);
  return f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u),
               __uint_as_float(raw[1] << 16), __uint_as_float(raw[1] & 0xffff0000u)};
}
__device__ __forceinline__ void store4(void* base, int64_t off, int dtype, f32x4 v) {
  if (dtype == ICL_F32) {
    *(f32x4*)((float*)base + off) = v;
  } else {
    u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2*)((unsigned short*)base + off) = pk;
  }
}

template <bool RMS>
__global__ __launch_bounds__(NT) void norm_kernel(const void* x, int64_t ldx, const void* res,
                                                   float alpha, const float* gamma,
                                                   const float* beta, void* y, int64_t ldy,
                                                   void* y2, int64_t ldy2, int N, float eps,
                                                   int in_dtype, int out_dtype) {
  __shared__ float red[4];
  const int64_t m = blockIdx.x;
  const int nvec = N >> 2;
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = threadIdx.x + i * NT;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nvec) {
      v[i] = load4(x, m * ldx + c * 4, in_dtype);
      if (res) {
        const f32x4 r = load4(res, m * ldx + c * 4, in_dtype);
        v[i] = v[i] + alpha * r;
      }
      s += RMS ? (v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3])
               : (v[i][0] + v[i][1] + v[i][2] + v[i][3]);
    }
  }
  float mean = 0.f, rstd;
  if (RMS) {
    const float ss = block_sum(s, red);
    rstd = rsqrtf(ss / (float)N + eps);
  } else {
    mean = block_sum(s, red) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = threadIdx.x + i * NT;
      if (c < nvec) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = v[i][r] - mean;
          q += d * d;
        }
      }
    }
    rstd = rsqrtf(block_sum(q, red) / (float)N + eps);
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = threadIdx.x + i * NT;
    if (c < nvec) {
      const f32x4 g = *(const f32x4*)(gamma + c * 4);
      f32x4 o;
      if (RMS) {
        o = v[i] * rstd * g;
      } else {
        const f32x4 b = *(const f32x4*)(beta + c * 4);
        o = (v[i] - mean) * rstd * g + b;
      }
      store4(y, m * ldy + c * 4, out_dtype, o);
      if (y2) store4(y2, m * ldy2 + c * 4, ICL_BF16, o);
    }
  }
}

int check_common(const char* name, const void* x, int64_t ldx, const float* gamma, void* y,
                 int64_t ldy, int M, int N, int in_dtype, int out_dtype) {
  ICL_CHECK_ARG(x && gamma && y, "%s: NULL pointer", name);
  ICL_CHECK_ARG(M > 0 && N > 0, "%s: M,N must be > 0", name);
  ICL_CHECK_ARG(N % 4 == 0 && N <= NT * 4 * MAXV, "%s: N=%d must be a multiple of 4 and <= %d", name, N,
                NT * 4 * MAXV);
  ICL_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= N && ldy >= N, "%s: bad leading dimensions", name);
  ICL_CHECK_ARG((in_dtype == ICL_F32 || in_dtype == ICL_BF16) && (out_dtype == ICL_F32 || out_dtype == ICL_BF16),
                "%s: bad dtype", name);
  ICL_CHECK_ARG(((uintptr_t)x % (in_dtype == ICL_F32 ? 16 : 8)) == 0 &&
                    ((uintptr_t)y % (out_dtype == ICL_F32 ? 16 : 8)) == 0 && ((uintptr_t)gamma % 16) == 0,
                "%s: misaligned pointer", name);
  return ICL_OK;
}

}  // namespace

extern "C" int icl_layernorm(const void* x, int64_t ldx, const void* res, float alpha,
                             const float* gamma, const float* beta, void* y, int64_t ldy, void* y2,
                             int64_t ldy2, int32_t M, int32_t N, float eps, int32_t in_dtype,
                             int32_t out_dtype, void* stream) {
  int rc = check_common("icl_layernorm", x, ldx, gamma, y, ldy, M, N, in_dtype, out_dtype);
  if (rc) return rc;
  ICL_CHECK_ARG(beta && ((uintptr_t)beta % 16) == 0, "icl_layernorm: beta NULL or misaligned");
  if (y2) ICL_CHECK_ARG(ldy2 % 4 == 0 && ldy2 >= N && ((uintptr_t)y2 % 8) == 0, "icl_layernorm: bad y2");
  if (res) ICL_CHECK_ARG(((uintptr_t)res % (in_dtype == ICL_F32 ? 16 : 8)) == 0, "icl_layernorm: res misaligned");
  hipLaunchKernelGGL(norm_kernel<false>, dim3(M), dim3(NT), 0, (hipStream_t)stream, x, ldx, res, alpha,
                     gamma, beta, y, ldy, y2, ldy2, N, eps, in_dtype, out_dtype);
  ICL_CHECK_LAUNCH("icl_layernorm");
  return ICL_OK;
}

extern "C" int icl_rmsnorm(const void* x, int64_t ldx, const float* gamma, void* y, int64_t ldy,
                           int32_t M, int32_t N, float eps, int32_t in_dtype, int32_t out_dtype,
                           void* stream) {
  int rc = check_common("icl_rmsnorm", x, ldx, gamma, y, ldy, M, N, in_dtype, out_dtype);
  if (rc) return rc;
  hipLaunchKernelGGL(norm_kernel<true>, dim3(M), dim3(NT), 0, (hipStream_t)stream, x, ldx,
                     (const void*)nullptr, 0.f, gamma, (const float*)nullptr, y, ldy, (void*)nullptr,
                     (int64_t)0, N, eps, in_dtype, out_dtype);
  ICL_CHECK_LAUNCH("icl_rmsnorm");
  return ICL_OK;
}
