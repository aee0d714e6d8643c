// gemm.hip — bf16 MFMA GEMM for gfx950:  C = epilogue(A[M,K] * W[N,K]^T), fp32 accumulate.
//
// Structure (cdna_hip_programming.md §5): LDS-staged, double-buffered K loop, BK = 64.
//   * global -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip); the LDS image
//     is lane-linear per wave-instruction (8 rows x 128 B), so the bank-conflict swizzle is
//     applied to the per-lane SOURCE address and again on the ds_read (rule 21):
//       chunk' = chunk ^ ((row >> 1) & 7)   (16-B chunks of a 128-B row)
//     which makes the ds_read_b128 fragment reads of 16 distinct rows conflict-free.
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand and the activation
//     fragment as the B operand, i.e. each MFMA produces a C^T sub-tile: a lane then owns 4
//     consecutive n for one m, so the epilogue stores 8 B (bf16) / 16 B (f32) per lane.
//   * XCD-aware block remap (bijective, T1) + grouped (GROUP_M) tile order so that the ~64
//     blocks resident on one XCD form an 8x8 super-tile sharing A and W panels in that L2.
// Tiles: 128x128 (2x2 waves, 4x4 MFMA tiles per wave) for prefill/encoder shapes,
//        64x64   (2x2 waves, 2x2 MFMA tiles per wave) for skinny / decode shapes (+ split-K).
#include "common.h"

namespace {

struct GemmParams {
  const __bf16* A;
  const __bf16* W;
  void* C;
  const float* bias;
  const void* R;
  float* ws;
  int64_t lda, ldw, ldc, ldr, sA, sC, sR;
  int M, N, K, epi, out_dtype, res_dtype, split_k, tiles_m, tiles_n;
};

constexpr int GROUP_M = 8;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int WAVES_M, int WAVES_N, int MI, int NI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  constexpr int BM = WAVES_M * MI * 16, BN = WAVES_N * NI * 16;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / 32, W_INSTR = BN / 32;  // glds wave-instructions per wave
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- block -> output tile: XCD remap (bijective) then grouped order -----------------------
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  int wgid;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  {
    const int per_group = GROUP_M * p.tiles_n;
    const int group = wgid / per_group;
    const int first_m = group * GROUP_M;
    const int gsize = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = wgid - group * per_group;
    tm = first_m + in_group % gsize;
    tn = in_group / gsize;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- batch / split-K ------------------------------------------------------------------------
  const int z = blockIdx.z;
  const __bf16* A = p.A;
  int kt0 = 0, kt1 = p.K >> 6;
  if (p.split_k > 1) {
    const int nk = p.K >> 6;
    kt0 = (int)(((int64_t)z * nk) / p.split_k);
    kt1 = (int)(((int64_t)(z + 1) * nk) / p.split_k);
  } else {
    A += (int64_t)z * p.sA;
  }

  // ---- per-lane source pointers for the LDS-DMA staging ------------------------------------
  // wave-instruction i of a tile covers rows 8i..8i+7; lane l -> row 8i + (l>>3), LDS slot l&7,
  // global chunk = slot ^ ((row>>1)&7).
  const __bf16* ga[A_INSTR];
  const __bf16* gw[W_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int gm = min(m0 + row, p.M - 1);
    ga[j] = A + (int64_t)gm * p.lda + chunk * 8;
  }
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int gn = min(n0 + row, p.N - 1);
    gw[j] = p.W + (int64_t)gn * p.ldw + chunk * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(ga[j] + (int64_t)kt * 64),
                                       (lptr_t)(base + (j * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(gw[j] + (int64_t)kt * 64),
                                       (lptr_t)(base + A_BYTES + (j * 4 + wave) * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets (bytes inside a tile image) -------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2], w_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = ((kk * 4 + fq) ^ (fr >> 1)) * 16;
    a_off[kk] = (wm * MI * 16 + fr) * 128 + sw;
    w_off[kk] = (wn * NI * 16 + fr) * 128 + sw;
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) {
    stage(0, kt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
      const char* a_s = smem + cur * BUF;
      const char* w_s = a_s + A_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[MI], wf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(a_s + a_off[kk] + i * 16 * 128);
#pragma unroll
        for (int j = 0; j < NI; ++j) wf[j] = *(const bf16x8*)(w_s + w_off[kk] + j * 16 * 128);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // acc[i][j][r] = C[m][n], m = m0 + wm*MI*16 + i*16 + fr, n = n0 + wn*NI*16 + j*16 + fq*4 + r
  if (p.split_k > 1) {
    float* ws = p.ws + (int64_t)z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * MI * 16 + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * NI * 16 + j * 16 + fq * 4;
        float* dst = ws + (int64_t)m * p.N + n;
        if (n + 3 < p.N && (p.N & 3) == 0) {
          *(f32x4*)dst = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) dst[r] = acc[i][j][r];
        }
      }
    }
    return;
  }

  const bool has_bias = p.epi & ICL_EPI_BIAS, has_gelu = p.epi & ICL_EPI_GELU;
  const bool has_res = p.epi & ICL_EPI_RESIDUAL, swiglu = p.epi & ICL_EPI_SWIGLU;
  char* Cb = (char*)p.C;
  const int64_t cz = (int64_t)z * p.sC, rz = (int64_t)z * p.sR;

  if (swiglu) {
    if constexpr (NI % 2 == 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * MI * 16 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          const int nt = n0 + wn * NI * 16 + j * 16;  // interleaved-row index of the gate block
          if (nt >= p.N) continue;
          const int oc = (nt >> 1) + fq * 4;          // output column
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float g = acc[i][j][r], u = acc[i][j + 1][r];
            if (has_bias) {
              g += p.bias[nt + fq * 4 + r];
              u += p.bias[nt + 16 + fq * 4 + r];
            }
            v[r] = silu_f(g) * u;
          }
          const int64_t off = cz + (int64_t)m * p.ldc + oc;
          if (p.out_dtype == ICL_BF16) {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *(u32x2*)(Cb + off * 2) = pk;
          } else {
            *(f32x4*)(Cb + off * 4) = f32x4{v[0], v[1], v[2], v[3]};
          }
        }
      }
    }
    return;
  }

  const bool vec_ok = ((p.ldc & 3) == 0) && (!has_res || (p.ldr & 3) == 0);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * MI * 16 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * NI * 16 + j * 16 + fq * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const bool full = (n + 3 < p.N);
      if (has_bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (full || n + r < p.N) v[r] += p.bias[n + r];
      }
      if (has_gelu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      }
      const int64_t coff = cz + (int64_t)m * p.ldc + n;
      if (full && vec_ok) {
        if (has_res) {
          const int64_t roff = rz + (int64_t)m * p.ldr + n;
          if (p.res_dtype == ICL_F32) {
            f32x4 rv = *(const f32x4*)((const char*)p.R + roff * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
          } else {
            const unsigned short* rp = (const unsigned short*)p.R + roff;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bf16_bits_to_f32(rp[r]);
          }
        }
        if (p.out_dtype == ICL_BF16) {
          u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *(u32x2*)(Cb + coff * 2) = pk;
        } else {
          *(f32x4*)(Cb + coff * 4) = f32x4{v[0], v[1], v[2], v[3]};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.N) continue;
          float x = v[r];
          if (has_res) {
            const int64_t roff = rz + (int64_t)m * p.ldr + n + r;
            x += (p.res_dtype == ICL_F32) ? ((const float*)p.R)[roff]
                                          : bf16_bits_to_f32(((const unsigned short*)p.R)[roff]);
          }
          if (p.out_dtype == ICL_BF16)
            ((unsigned short*)Cb)[coff + r] = f32_to_bf16_bits(x);
          else
            ((float*)Cb)[coff + r] = x;
        }
      }
    }
  }
}

// split-K reduction + epilogue: one thread per output element (4 consecutive n when aligned).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p) {
  const bool swiglu = p.epi & ICL_EPI_SWIGLU;
  const int Nout = swiglu ? p.N / 2 : p.N;
  const int64_t total = (int64_t)p.M * Nout;
  const int64_t slab = (int64_t)p.M * p.N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / Nout), c = (int)(idx % Nout);
    float v;
    if (swiglu) {
      const int ng = (c >> 4) * 32 + (c & 15), nu = ng + 16;
      float g = 0.f, u = 0.f;
      for (int s = 0; s < p.split_k; ++s) {
        g += p.ws[s * slab + (int64_t)m * p.N + ng];
        u += p.ws[s * slab + (int64_t)m * p.N + nu];
      }
      if (p.epi & ICL_EPI_BIAS) {
        g += p.bias[ng];
        u += p.bias[nu];
      }
      v = silu_f(g) * u;
    } else {
      v = 0.f;
      for (int s = 0; s < p.split_k; ++s) v += p.ws[s * slab + (int64_t)m * p.N + c];
      if (p.epi & ICL_EPI_BIAS) v += p.bias[c];
      if (p.epi & ICL_EPI_GELU) v = gelu_erf(v);
      if (p.epi & ICL_EPI_RESIDUAL) {
        const int64_t roff = (int64_t)m * p.ldr + c;
        v += (p.res_dtype == ICL_F32) ? ((const float*)p.R)[roff]
                                      : bf16_bits_to_f32(((const unsigned short*)p.R)[roff]);
      }
    }
    const int64_t coff = (int64_t)m * p.ldc + c;
    if (p.out_dtype == ICL_BF16)
      ((unsigned short*)p.C)[coff] = f32_to_bf16_bits(v);
    else
      ((float*)p.C)[coff] = v;
  }
}

template <int WAVES_M, int WAVES_N, int MI, int NI>
int launch_tile(GemmParams& p, int batch, hipStream_t stream) {
  constexpr int BM = WAVES_M * MI * 16, BN = WAVES_N * NI * 16;
  constexpr int SMEM = (BM + BN) * 128 * 2;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  auto kern = gemm_bf16_kernel<WAVES_M, WAVES_N, MI, NI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      icl_set_error("icl_gemm_bf16: hipFuncSetAttribute(%d) failed: %s", SMEM, hipGetErrorString(e));
      return ICL_ELAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.split_k > 1 ? p.split_k : batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, stream, p);
  ICL_CHECK_LAUNCH("icl_gemm_bf16");
  return ICL_OK;
}

}  // namespace

extern "C" int icl_gemm_bf16(const icl_gemm_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ICL_CHECK_ARG(a != nullptr, "icl_gemm_bf16: args is NULL");
  ICL_CHECK_ARG(a->A && a->W && a->C, "icl_gemm_bf16: A/W/C must be non-NULL");
  ICL_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "icl_gemm_bf16: M,N,K must be > 0 (got %d,%d,%d)",
                a->M, a->N, a->K);
  ICL_CHECK_ARG(a->K % 64 == 0, "icl_gemm_bf16: K=%d must be a multiple of 64", a->K);
  ICL_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0,
                "icl_gemm_bf16: lda=%lld / ldw=%lld must be multiples of 8", (long long)a->lda,
                (long long)a->ldw);
  ICL_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0,
                "icl_gemm_bf16: A and W must be 16-byte aligned");
  ICL_CHECK_ARG(a->ldw >= a->K, "icl_gemm_bf16: ldw=%lld < K=%d", (long long)a->ldw, a->K);
  ICL_CHECK_ARG(a->batch >= 1 && a->batch <= 65535, "icl_gemm_bf16: batch=%d out of range", a->batch);
  ICL_CHECK_ARG(a->split_k >= 1 && a->split_k <= 64, "icl_gemm_bf16: split_k=%d out of range", a->split_k);
  ICL_CHECK_ARG(a->out_dtype == ICL_BF16 || a->out_dtype == ICL_F32, "icl_gemm_bf16: bad out_dtype %d",
                a->out_dtype);
  ICL_CHECK_ARG(a->res_dtype == ICL_BF16 || a->res_dtype == ICL_F32, "icl_gemm_bf16: bad res_dtype %d",
                a->res_dtype);
  ICL_CHECK_ARG((a->epilogue & ~15) == 0, "icl_gemm_bf16: unknown epilogue bits 0x%x", a->epilogue);
  const bool swiglu = a->epilogue & ICL_EPI_SWIGLU;
  if (swiglu) {
    ICL_CHECK_ARG(a->N % 32 == 0, "icl_gemm_bf16: SWIGLU needs N %% 32 == 0 (N=%d)", a->N);
    ICL_CHECK_ARG((a->epilogue & (ICL_EPI_GELU | ICL_EPI_RESIDUAL)) == 0,
                  "icl_gemm_bf16: SWIGLU cannot be combined with GELU/RESIDUAL");
    ICL_CHECK_ARG(a->ldc % 4 == 0, "icl_gemm_bf16: SWIGLU needs ldc %% 4 == 0");
  }
  if (a->epilogue & ICL_EPI_BIAS) ICL_CHECK_ARG(a->bias, "icl_gemm_bf16: EPI_BIAS without bias");
  if (a->epilogue & ICL_EPI_RESIDUAL) ICL_CHECK_ARG(a->R, "icl_gemm_bf16: EPI_RESIDUAL without R");
  const int nout = swiglu ? a->N / 2 : a->N;
  ICL_CHECK_ARG(a->ldc >= nout, "icl_gemm_bf16: ldc=%lld < %d output columns", (long long)a->ldc, nout);
  if (a->split_k > 1) {
    ICL_CHECK_ARG(a->batch == 1, "icl_gemm_bf16: split_k > 1 requires batch == 1");
    ICL_CHECK_ARG(a->workspace, "icl_gemm_bf16: split_k > 1 requires a workspace");
    ICL_CHECK_ARG(a->split_k <= a->K / 64, "icl_gemm_bf16: split_k=%d > K/64=%d", a->split_k, a->K / 64);
  }
  // vector stores need an aligned C (and R) base
  const int cal = a->out_dtype == ICL_BF16 ? 8 : 16;
  ICL_CHECK_ARG(((uintptr_t)a->C % cal) == 0, "icl_gemm_bf16: C must be %d-byte aligned", cal);
  if (a->R) {
    const int ral = a->res_dtype == ICL_BF16 ? 8 : 16;
    ICL_CHECK_ARG(((uintptr_t)a->R % ral) == 0, "icl_gemm_bf16: R must be %d-byte aligned", ral);
  }

  GemmParams p;
  p.A = (const __bf16*)a->A;
  p.W = (const __bf16*)a->W;
  p.C = a->C;
  p.bias = a->bias;
  p.R = a->R;
  p.ws = a->workspace;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->ldr;
  p.sA = a->strideA; p.sC = a->strideC; p.sR = a->strideR;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.epi = a->epilogue; p.out_dtype = a->out_dtype; p.res_dtype = a->res_dtype;
  p.split_k = a->split_k;
  p.tiles_m = p.tiles_n = 0;

  int tile = a->tile;
  if (tile == 0) {
    // 128x128 unless the problem is skinny or would leave most CUs idle
    const int64_t t128 = (int64_t)((a->M + 127) / 128) * ((a->N + 127) / 128) * a->batch;
    tile = (a->M <= 64 || t128 < 256) ? 2 : 1;
  }
  int rc;
  if (tile == 1)
    rc = launch_tile<2, 2, 4, 4>(p, a->batch, stream);
  else if (tile == 2)
    rc = launch_tile<2, 2, 2, 2>(p, a->batch, stream);
  else {
    icl_set_error("icl_gemm_bf16: unsupported tile id %d", tile);
    return ICL_EINVAL;
  }
  if (rc != ICL_OK) return rc;
  if (a->split_k > 1) {
    const int64_t total = (int64_t)a->M * nout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    ICL_CHECK_LAUNCH("icl_gemm_bf16(split-K reduce)");
  }
  return ICL_OK;
}
