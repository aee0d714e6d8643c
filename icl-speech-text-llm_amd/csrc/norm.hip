// norm.hip — LayerNorm / RMSNorm for gfx950 (HBM-bound; one 64-lane wave per row, 4 rows per block).
// The row lives in registers between the statistics passes (no re-read): each lane owns up to
// MAXV 4-element vectors (16 B f32 / 8 B bf16 loads, cdna_hip_programming.md Guideline 13).
// Statistics are two-pass in f32 (mean, then sum of squared deviations), matching torch's
// LayerNorm numerics more closely than E[x^2]-mean^2.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXN = 8192;  // 64 lanes x 4 x 32 chunks

#ifndef ICL_NORM_NT
#define ICL_NORM_NT 0     // A/B: non-temporal loads of the f32 input stream (read once per norm)
#endif
__device__ __forceinline__ f32x4 load4(const void* base, int64_t off, int dtype) {
  if (dtype == ICL_F32) return ICL_NORM_NT ? __builtin_nontemporal_load((const f32x4*)((const float*)base + off)) : *(const f32x4*)((const float*)base + off);
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

// One WAVE per row (4 rows per block): the row lives in registers, both statistics are shuffle-only
// reductions — no LDS, no barrier.  MAXV = float4 chunks per lane (N <= 64*4*MAXV).
template <bool RMS, int MAXV, bool HOIST>
__global__ __launch_bounds__(NT) void norm_kernel(const void* x, int64_t ldx, const void* res,
                                                   float alpha, const float* gamma,
                                                   const float* beta, void* y, int64_t ldy,
                                                   void* y2, int64_t ldy2, int M, int N, float eps,
                                                   int in_dtype, int out_dtype) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int nvec = N >> 2;
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
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
  // HOIST (launches of few rows): gamma / beta are loaded HERE, under the reductions.  Left inside the store loop below, each load
  // waits behind the previous chunk's store (y may alias them as far as the compiler knows), i.e. 16 serialised L2 round trips
  // per row — 12-13 us for the single-row launches of a small decode batch, where nothing else hides them.  Same values, same
  // arithmetic, so the choice may depend on M; at large M the extra registers cost occupancy and bandwidth (measured, round 2).
  f32x4 gv[HOIST ? MAXV : 1], bv[(HOIST && !RMS) ? MAXV : 1];
  if (HOIST) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        gv[i] = *(const f32x4*)(gamma + c * 4);
        if (!RMS) bv[i] = *(const f32x4*)(beta + c * 4);
      }
    }
  }
  float mean = 0.f, rstd;
  if (RMS) {
    rstd = rsqrtf(wave_reduce_sum(s) / (float)N + eps);
  } else {
    mean = wave_reduce_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = v[i][r] - mean;
          q += d * d;
        }
      }
    }
    rstd = rsqrtf(wave_reduce_sum(q) / (float)N + eps);
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      const f32x4 g = HOIST ? gv[i] : *(const f32x4*)(gamma + c * 4);
      f32x4 o;
      if (RMS) {
        o = v[i] * rstd * g;
      } else {
        const f32x4 b = HOIST ? bv[i] : *(const f32x4*)(beta + c * 4);
        o = (v[i] - mean) * rstd * g + b;
      }
      store4(y, m * ldy + c * 4, out_dtype, o);
      if (y2) store4(y2, m * ldy2 + c * 4, ICL_BF16, o);
    }
  }
}

// Streaming variant for the large activation matrices (N % 8 == 0): each lane owns 8 CONTIGUOUS elements per chunk (two 16-B
// f32 loads / one 16-B bf16 load, ONE 16-B bf16 store instead of two 8-B ones), a wave walks rows w, w + W, ... of a
// persistent grid and issues the loads of its next row before reducing the current one, so every wave keeps two rows of
// loads in flight.  Same two-pass f32 statistics as norm_kernel.
__device__ __forceinline__ void load8(const void* base, int64_t off, int dtype, float (&v)[8]) {
  if (dtype == ICL_F32) {
    const f32x4 a = ICL_NORM_NT ? __builtin_nontemporal_load((const f32x4*)((const float*)base + off)) : *(const f32x4*)((const float*)base + off);
    const f32x4 b = ICL_NORM_NT ? __builtin_nontemporal_load((const f32x4*)((const float*)base + off + 4)) : *(const f32x4*)((const float*)base + off + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = a[r]; v[4 + r] = b[r]; }
  } else {
    const u32x4 raw = *(const u32x4*)((const unsigned short*)base + off);
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[2 * r] = __uint_as_float(raw[r] << 16); v[2 * r + 1] = __uint_as_float(raw[r] & 0xffff0000u); }
  }
}
__device__ __forceinline__ void store8(void* base, int64_t off, int dtype, const float (&v)[8]) {
  if (dtype == ICL_F32) {
    *(f32x4*)((float*)base + off) = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)((float*)base + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    u32x4 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    *(u32x4*)((unsigned short*)base + off) = pk;
  }
}

template <bool RMS, int MAXC>
__global__ __launch_bounds__(NT) void norm8_kernel(const void* x, int64_t ldx, const void* res, float alpha,
                                                    const float* gamma, const float* beta, void* y, int64_t ldy,
                                                    void* y2, int64_t ldy2, int M, int N, float eps, int in_dtype,
                                                    int out_dtype) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  const int nch = N >> 3;
  float cur[MAXC][8], nxt[MAXC][8];
  auto load_row = [&](int64_t m, float (&dst)[MAXC][8]) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        load8(x, m * ldx + c * 8, in_dtype, dst[i]);
        if (res) {
          float r[8];
          load8(res, m * ldx + c * 8, in_dtype, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) dst[i][e] += alpha * r[e];
        }
      }
    }
  };
  int64_t m = wave;
  if (m < M) load_row(m, cur);
  for (; m < M; m += n_waves) {
    const bool more = m + n_waves < M;
    if (more) load_row(m + n_waves, nxt);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s += RMS ? cur[i][e] * cur[i][e] : cur[i][e];
      }
    float mean = 0.f, rstd;
    if (RMS) {
      rstd = rsqrtf(wave_reduce_sum(s) / (float)N + eps);
    } else {
      mean = wave_reduce_sum(s) / (float)N;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
        if (lane + i * 64 < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = cur[i][e] - mean;
            q += d * d;
          }
        }
      rstd = rsqrtf(wave_reduce_sum(q) / (float)N + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float g[8], o[8];
        load8(gamma, c * 8, ICL_F32, g);
        if (RMS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = cur[i][e] * rstd * g[e];
        } else {
          float b[8];
          load8(beta, c * 8, ICL_F32, b);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (cur[i][e] - mean) * rstd * g[e] + b[e];
        }
        store8(y, m * ldy + c * 8, out_dtype, o);
        if (y2) store8(y2, m * ldy2 + c * 8, ICL_BF16, o);
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) cur[i][e] = nxt[i][e];
    }
  }
}

template <bool RMS>
void launch_norm(hipStream_t st, const void* x, int64_t ldx, const void* res, float alpha, const float* gamma,
                 const float* beta, void* y, int64_t ldy, void* y2, int64_t ldy2, int M, int N, float eps,
                 int in_dtype, int out_dtype) {
  const dim3 block(NT);
  // streaming variant: 8-element chunks, persistent waves with a one-row prefetch (alignment of every operand permitting)
  const bool a32 = ((uintptr_t)x % (in_dtype == ICL_F32 ? 16 : 16)) == 0 && (!res || ((uintptr_t)res % 16) == 0) &&
                   ((uintptr_t)y % 16) == 0 && (!y2 || ((uintptr_t)y2 % 16) == 0) && ((uintptr_t)gamma % 16) == 0 &&
                   (RMS || ((uintptr_t)beta % 16) == 0);
  // measured at micro-batch 128 (tools/bench_norm128.py): Whisper LN (N = 1280) 399 -> 317 us, 13B RMS (N = 5120) 258 -> 221 us;
  // N <= 1024 (BEATs), N = 4096 and the residual / dual-output forms are as fast or faster on the one-row-per-wave kernel.
  // The choice depends on N and the call form only, never on M: a row must round the same in any batch.
  const int pl8 = ((N >> 3) + 63) / 64;
#ifndef ICL_NORM8_ALL
#define ICL_NORM8_ALL 0      // A/B: route every eligible shape to the streaming variant
#endif
  if (N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && !res && !y2 && a32 && N <= 5120 && (ICL_NORM8_ALL || pl8 == 3 || pl8 > 8)) {
    static int n_cu = 0;
    if (n_cu <= 0) {
      n_cu = icl_device_cu_count();
      if (n_cu <= 0) n_cu = 256;
    }
    const int per_lane8 = ((N >> 3) + 63) / 64;
    const dim3 pgrid(min((M + 3) / 4, n_cu * (per_lane8 <= 3 ? 8 : 4)));
#define ICL_NORM8_CASE(C)                                                                                              \
  hipLaunchKernelGGL((norm8_kernel<RMS, C>), pgrid, block, 0, st, x, ldx, res, alpha, gamma, beta, y, ldy, y2, ldy2, M, N, \
                     eps, in_dtype, out_dtype)
    if (per_lane8 <= 2) ICL_NORM8_CASE(2);        // N <= 1024 (BEATs, Q-Former)
    else if (per_lane8 <= 3) ICL_NORM8_CASE(3);   // N <= 1536 (Whisper)
    else if (per_lane8 <= 8) ICL_NORM8_CASE(8);   // N <= 4096 (Llama-7B)
    else ICL_NORM8_CASE(10);                      // N <= 5120 (Llama-13B)
#undef ICL_NORM8_CASE
    return;
  }
  const dim3 grid((M + 3) / 4);
  const int per_lane = ((N >> 2) + 63) / 64;   // float4 chunks per lane; exact-fit instantiations keep VGPRs (and so occupancy) tight
  const bool few_rows = M <= 2048;             // at most two workgroups per CU: latency, not bandwidth, is what the launch costs
#define ICL_NORM_CASE(V)                                                                                                  \
  do {                                                                                                                    \
    constexpr bool CAN = V * (RMS ? 2 : 3) * 4 <= 200;      /* registers: row + gamma (+ beta) */                          \
    if (CAN && few_rows)                                                                                                  \
      hipLaunchKernelGGL((norm_kernel<RMS, V, CAN>), grid, block, 0, st, x, ldx, res, alpha, gamma, beta, y, ldy, y2, ldy2, M, \
                         N, eps, in_dtype, out_dtype);                                                                    \
    else                                                                                                                  \
      hipLaunchKernelGGL((norm_kernel<RMS, V, false>), grid, block, 0, st, x, ldx, res, alpha, gamma, beta, y, ldy, y2, ldy2, \
                         M, N, eps, in_dtype, out_dtype);                                                                 \
  } while (0)
  if (per_lane <= 3) ICL_NORM_CASE(3);         // N <= 768  (BEATs, Q-Former)
  else if (per_lane <= 5) ICL_NORM_CASE(5);    // N <= 1280 (Whisper)
  else if (per_lane <= 8) ICL_NORM_CASE(8);    // N <= 2048
  else if (per_lane <= 16) ICL_NORM_CASE(16);  // N <= 4096 (Llama-7B)
  else if (per_lane <= 20) ICL_NORM_CASE(20);  // N <= 5120 (Llama-13B)
  else ICL_NORM_CASE(32);
#undef ICL_NORM_CASE
}

int check_common(const char* name, const void* x, int64_t ldx, const float* gamma, void* y,
                 int64_t ldy, int M, int N, int in_dtype, int out_dtype) {
  ICL_CHECK_ARG(x && gamma && y, "%s: NULL pointer", name);
  ICL_CHECK_ARG(M > 0 && N > 0, "%s: M,N must be > 0", name);
  ICL_CHECK_ARG(N % 4 == 0 && N <= MAXN, "%s: N=%d must be a multiple of 4 and <= %d", name, N,
                MAXN);
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
  launch_norm<false>((hipStream_t)stream, x, ldx, res, alpha, gamma, beta, y, ldy, y2, ldy2, M, N, eps, in_dtype, out_dtype);
  ICL_CHECK_LAUNCH("icl_layernorm");
  return ICL_OK;
}

extern "C" int icl_rmsnorm(const void* x, int64_t ldx, const float* gamma, void* y, int64_t ldy,
                           int32_t M, int32_t N, float eps, int32_t in_dtype, int32_t out_dtype,
                           void* stream) {
  int rc = check_common("icl_rmsnorm", x, ldx, gamma, y, ldy, M, N, in_dtype, out_dtype);
  if (rc) return rc;
  launch_norm<true>((hipStream_t)stream, x, ldx, nullptr, 0.f, gamma, nullptr, y, ldy, nullptr, 0, M, N, eps, in_dtype, out_dtype);
  ICL_CHECK_LAUNCH("icl_rmsnorm");
  return ICL_OK;
}
