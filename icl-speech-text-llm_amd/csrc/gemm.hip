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
#include <algorithm>
#include <type_traits>

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


// ---- epilogue helpers shared by all tile shapes ---------------------------------------------------------
// v[0..3] = C[m][n..n+3] (4 consecutive n owned by one lane).
__device__ __forceinline__ void epi_store4(const GemmParams& p, int z, int m, int n, f32x4 acc) {
  if (m >= p.M || n >= p.N) return;
  const bool has_bias = p.epi & ICL_EPI_BIAS, has_gelu = p.epi & ICL_EPI_GELU, has_res = p.epi & ICL_EPI_RESIDUAL;
  char* Cb = (char*)p.C;
  const int64_t cz = (int64_t)z * p.sC, rz = (int64_t)z * p.sR;
  const bool vec_ok = ((p.ldc & 3) == 0) && (!has_res || (p.ldr & 3) == 0);
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const bool full = (n + 3 < p.N);
  if (has_bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (full || n + r < p.N) v[r] += p.bias[n + r];
  }
  if (has_gelu) {
    const f32x4 g = gelu_erf4(f32x4{v[0], v[1], v[2], v[3]});
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = g[r];
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
// gate block at interleaved rows nt + fq4 + r, up block 16 rows later; output column nt/2 + fq4 + r
__device__ __forceinline__ void epi_store_swiglu(const GemmParams& p, int z, int m, int nt, int fq4, f32x4 g4, f32x4 u4) {
  if (m >= p.M || nt >= p.N) return;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float g = g4[r], u = u4[r];
    if (p.epi & ICL_EPI_BIAS) {
      g += p.bias[nt + fq4 + r];
      u += p.bias[nt + 16 + fq4 + r];
    }
    v[r] = silu_f(g) * u;
  }
  const int64_t off = (int64_t)z * p.sC + (int64_t)m * p.ldc + (nt >> 1) + fq4;
  if (p.out_dtype == ICL_BF16) {
    u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2*)((char*)p.C + off * 2) = pk;
  } else {
    *(f32x4*)((char*)p.C + off * 4) = f32x4{v[0], v[1], v[2], v[3]};
  }
}
__device__ __forceinline__ void epi_store_partial(const GemmParams& p, int z, int m, int n, f32x4 acc) {
  if (m >= p.M || n >= p.N) return;
  float* dst = p.ws + (int64_t)z * p.M * p.N + (int64_t)m * p.N + n;
  if (n + 3 < p.N && (p.N & 3) == 0) {
    *(f32x4*)dst = acc;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < p.N) dst[r] = acc[r];
  }
}

// ---- interior-tile fast path: no bounds checks, bias as one 16-B load per n-fragment (prefetched before the K loop),
// ---- residual fragments loaded as ONE batch (independent loads in flight together), then add + convert + store.
__device__ __forceinline__ f32x4 load_res4(const GemmParams& p, int64_t roff) {
  if (p.res_dtype == ICL_F32) return *(const f32x4*)((const char*)p.R + roff * 4);
  const u32x2 raw = *(const u32x2*)((const char*)p.R + roff * 2);
  return f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u),
               __uint_as_float(raw[1] << 16), __uint_as_float(raw[1] & 0xffff0000u)};
}
__device__ __forceinline__ void store_out4(const GemmParams& p, int64_t coff, f32x4 v) {
  if (p.out_dtype == ICL_BF16) {
    u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2*)((char*)p.C + coff * 2) = pk;
  } else {
    *(f32x4*)((char*)p.C + coff * 4) = v;
  }
}
// Tile-INDEPENDENT vector-path predicate: whether a row's arithmetic order (accumulator init = bias + residual) may
// depend only on the problem, never on which tile of the grid the row falls in -> results are batch-invariant.
__device__ __forceinline__ bool vec_path_ok(const GemmParams& p) {
  const bool has_res = p.epi & ICL_EPI_RESIDUAL;
  return ((p.ldc & 3) == 0) && ((p.N & 3) == 0) && (!has_res || (p.ldr & 3) == 0) &&
         (!(p.epi & ICL_EPI_BIAS) || (((uintptr_t)p.bias & 15) == 0));
}
__device__ __forceinline__ bool tile_is_interior(const GemmParams& p, int m0, int n0, int BM, int BN) {
  return (m0 + BM <= p.M) && (n0 + BN <= p.N) && vec_path_ok(p);
}
__device__ __forceinline__ void block_to_tile(const GemmParams& p, int bid, int& tm, int& tn) {
  const int nwg = p.tiles_m * p.tiles_n;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);   // bijective XCD remap
  const int per_group = GROUP_M * p.tiles_n;
  const int group = wgid / per_group;
  const int first_m = group * GROUP_M;
  const int gsize = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = wgid - group * per_group;
  tm = first_m + in_group % gsize;
  tn = in_group / gsize;
}

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

  int tm, tn;
  block_to_tile(p, blockIdx.x, tm, tn);
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

  // The epilogue operands are folded into the accumulator INIT: acc = bias (+ f32 residual when no activation follows),
  // loaded while the first K-tile is in flight, so no dependent global load is left after the K loop.  The fold
  // decision is tile-independent (vec_path_ok); edge tiles only add bounds guards to the same loads.
  const bool vecp = p.split_k == 1 && vec_path_ok(p);
  const bool interior = vecp && (m0 + BM <= p.M) && (n0 + BN <= p.N);
  const bool fold_bias = vecp && (p.epi & ICL_EPI_BIAS);
  // the residual is added LAST, (bias + sum) + r, in every kernel and every tile (interior, edge, any tile shape): the order
  // is part of the batch-invariance contract; the 256x256 kernel reads it as whole rows in its LDS-staged epilogue
  constexpr bool fold_res = false;

  // order matters: pure loads first (no use -> no wait), then the LDS-DMA of K-tile 0, then the first use (one wait
  // that covers everything); a use placed between loads would make hipcc drain vmcnt(0) per load.
  f32x4 bias_f[NI];
  {
    const int mb = m0 + wm * MI * 16 + fr, nb = n0 + wn * NI * 16 + fq * 4;
    if (fold_res) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          if (interior || (mb + i * 16 < p.M && nb + j * 16 < p.N))
            acc[i][j] = *(const f32x4*)((const float*)p.R + (int64_t)z * p.sR + (int64_t)(mb + i * 16) * p.ldr + nb + j * 16);
    }
    if (fold_bias) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bias_f[j] = (interior || nb + j * 16 < p.N) ? *(const f32x4*)(p.bias + nb + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (kt0 < kt1) {
    __builtin_amdgcn_sched_barrier(0);
    stage(0, kt0);
    __builtin_amdgcn_sched_barrier(0);
    if (fold_bias) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = acc[i][j] + bias_f[j];
    }
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

  // ---- epilogue: acc[i][j][r] = C[m][n], m = m0 + wm*MI*16 + i*16 + fr, n = n0 + wn*NI*16 + j*16 + fq*4 + r ----
  if (interior) {
    const int mb = m0 + wm * MI * 16 + fr, nb = n0 + wn * NI * 16 + fq * 4;
    if (p.epi & ICL_EPI_SWIGLU) {
      if constexpr (NI % 2 == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; j += 2) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(acc[i][j][r]) * acc[i][j + 1][r];
            store_out4(p, (int64_t)z * p.sC + (int64_t)(mb + i * 16) * p.ldc + ((n0 + wn * NI * 16 + j * 16) >> 1) + fq * 4, v);
          }
      }
      return;
    }
    const bool late_res = (p.epi & ICL_EPI_RESIDUAL) && !fold_res;   // activation, then residual (Whisper conv2 + pos)
    f32x4 rv[MI][NI];
    if (late_res) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          rv[i][j] = load_res4(p, (int64_t)z * p.sR + (int64_t)(mb + i * 16) * p.ldr + nb + j * 16);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        f32x4 v = acc[i][j];
        if (p.epi & ICL_EPI_GELU) {
            v = gelu_erf4(v);
        }
        if (late_res) v = v + rv[i][j];
        store_out4(p, (int64_t)z * p.sC + (int64_t)(mb + i * 16) * p.ldc + nb + j * 16, v);
      }
    return;
  }
  GemmParams q = p;   // edge tiles: same arithmetic, bounds-checked stores; operands already folded are not re-applied
  if (fold_bias) q.epi &= ~ICL_EPI_BIAS;
  if (fold_res) q.epi &= ~ICL_EPI_RESIDUAL;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * MI * 16 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nt = n0 + wn * NI * 16 + j * 16;
      if (p.split_k > 1) {
        epi_store_partial(p, z, m, nt + fq * 4, acc[i][j]);
      } else if (p.epi & ICL_EPI_SWIGLU) {
        if constexpr (NI % 2 == 0) {
          if ((j & 1) == 0) epi_store_swiglu(q, z, m, nt, fq * 4, acc[i][j], acc[i][j + 1]);
        }
      } else {
        epi_store4(q, z, m, nt + fq * 4, acc[i][j]);
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


// =================================================================================================================
// 256x256x64 tile, 8 waves (2 x 4), one block per CU, 128 KiB LDS: a rolling LDS-DMA pipeline (guide §5 "8-phase").
//
//  * LDS = 2 buffers (K-tile parity) x 4 regions {A0, A1, B0, B1}; a region = 128 rows x 64 k (16 KiB) = one staging
//    granule = 2 global_load_lds_dwordx4 per thread.  Wave (wr, wc) owns C rows {wr*64..+63} of BOTH A regions and
//    C columns {wc*32..+31} of BOTH B regions, so each of the 4 phases of a K-tile (one 64x32 quadrant x K=64 =
//    16 MFMAs per wave) touches ONE A region and ONE B region for every wave:
//        P0: read A0,B0 -> q(0,0) | P1: read B1 -> q(0,1) | P2: read A1 -> q(1,1) | P3: (B0 frags kept) -> q(1,0)
//  * every phase stages exactly one granule, 5-6 phases ahead of its first read and >= 2 phases after the last read of
//    the region it overwrites:   P0: B1(t+1)  P1: A1(t+1)  P2: A0(t+2)  P3: B0(t+2)
//    so 4 granules (8 LDS-DMA per thread) stay in flight ACROSS barriers: each phase ends with a counted
//    `s_waitcnt vmcnt(8)` (never 0 in the loop) + ONE raw s_barrier, then its MFMA cluster under s_setprio(1).
//  * past the last K-tile the stages re-load the last tile into regions nobody reads any more, which keeps the
//    vmcnt arithmetic uniform (<= 6 wasted granules per block).
// =================================================================================================================
constexpr int T256_REGION = 128 * 128;          // bytes
constexpr int T256_BUF = 4 * T256_REGION;       // A0 A1 B0 B1
constexpr int T256_SMEM = 256 * (256 * 2 + 16);   // 135168: the two K-tile buffers (131072) / the C staging of the epilogue
                                                  // (whole bf16 tile, or one 128-row half in f32: 133120)

// Fused RoPE + KV-cache append for the QKV projection (icl_gemm_rope_kv_bf16): the row phase of the staged epilogue.
// head_dim = 128, so a 256-column tile holds two whole heads of q, of k or of v; the staged tile is bf16, i.e. the
// rotation sees exactly the values the unfused path would have read back from HBM (same rounding points, rope_rot8).
struct RopeFuse {
  const float* cosT;
  const float* sinT;
  const int* pos;
  const int* seq_ids;
  unsigned short* kc;
  unsigned short* vc;
  int k_off, v_off, H, max_len;
  int kv_rows_to_c;   // 0: k / v go to the cache only (the prefill attention reads them there)
};

__device__ __forceinline__ void rope_rows(const GemmParams& p, const RopeFuse& rf, const char* smem, int pitch, int m0,
                                          int n0, int tid) {
  const int sect = n0 >= rf.v_off ? 2 : (n0 >= rf.k_off ? 1 : 0);
  const int head0 = (n0 - (sect == 2 ? rf.v_off : sect == 1 ? rf.k_off : 0)) >> 7;
  unsigned short* C = (unsigned short*)p.C;
  constexpr int U = 4;
  if (sect == 2) {   // v: whole rows to the QKV buffer and to the cache
    for (int base = tid; base < 256 * 32; base += U * 512) {
      int64_t crow[U];
      if (rf.vc) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int m = min(m0 + ((base + u * 512) >> 5), p.M - 1);
          crow[u] = (int64_t)rf.seq_ids[m] * rf.H * rf.max_len + rf.pos[m];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = base + u * 512, row = c >> 5, cc = c & 31, m = m0 + row;
        if (m >= p.M) continue;
        const u32x4 v = *(const u32x4*)(smem + row * pitch + cc * 16);
        if (rf.kv_rows_to_c) *(u32x4*)(C + (int64_t)m * p.ldc + n0 + cc * 8) = v;
        if (rf.vc) *(u32x4*)(rf.vc + (crow[u] + (int64_t)(head0 + (cc >> 4)) * rf.max_len) * 128 + (cc & 15) * 8) = v;
      }
    }
    return;
  }
  const bool to_cache = sect == 1 && rf.kc;
  for (int base = tid; base < 256 * 16; base += U * 512) {   // items: (row, head-in-tile, 8-column piece of the low half)
    int ps[U], sq[U];
    u32x4 lo[U], hi[U];
    f32x4 c0[U], c1[U], s0[U], s1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = min(m0 + ((base + u * 512) >> 4), p.M - 1);
      ps[u] = rf.pos[m];
      sq[u] = to_cache ? rf.seq_ids[m] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = base + u * 512, row = it >> 4, hh = (it >> 3) & 1, j = it & 7;
      const char* src = smem + row * pitch + hh * 256 + j * 16;
      lo[u] = *(const u32x4*)src;
      hi[u] = *(const u32x4*)(src + 128);
      const float* cp = rf.cosT + (int64_t)ps[u] * 64 + j * 8;
      const float* sp = rf.sinT + (int64_t)ps[u] * 64 + j * 8;
      c0[u] = *(const f32x4*)cp;
      c1[u] = *(const f32x4*)(cp + 4);
      s0[u] = *(const f32x4*)sp;
      s1[u] = *(const f32x4*)(sp + 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = base + u * 512, row = it >> 4, hh = (it >> 3) & 1, j = it & 7, m = m0 + row;
      if (m >= p.M) continue;
      u32x4 olo, ohi;
      rope_rot8(lo[u], hi[u], c0[u], c1[u], s0[u], s1[u], olo, ohi);
      if (sect == 0 || rf.kv_rows_to_c) {
        unsigned short* dst = C + (int64_t)m * p.ldc + n0 + hh * 128 + j * 8;
        *(u32x4*)dst = olo;
        *(u32x4*)(dst + 64) = ohi;
      }
      if (to_cache) {
        unsigned short* cd = rf.kc + (((int64_t)sq[u] * rf.H + head0 + hh) * rf.max_len + ps[u]) * 128 + j * 8;
        *(u32x4*)cd = olo;
        *(u32x4*)(cd + 64) = ohi;
      }
    }
  }
}

template <bool ROPE>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmParams p, RopeFuse rf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int tm, tn;
  block_to_tile(p, blockIdx.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int z = blockIdx.z;
  const __bf16* A = p.A + (int64_t)z * p.sA;
  const int nk = p.K >> 6;

  // ---- staging sources: region h, round r -> rows 8*(r*8 + wave) + (lane>>3) of the region -------------------
  const __bf16* gsrc[2][2][2];  // [A|B][region][round]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = (r * 8 + wave) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      gsrc[0][h][r] = A + (int64_t)min(m0 + h * 128 + row, p.M - 1) * p.lda + chunk * 8;
      gsrc[1][h][r] = p.W + (int64_t)min(n0 + h * 128 + row, p.N - 1) * p.ldw + chunk * 8;
    }
  auto stage = [&](int bo, int which, int h, int kt) {   // bo: byte offset of the K-tile buffer (0 | T256_BUF); which: 0 = A, 1 = B
    const int64_t koff = (int64_t)min(kt, nk - 1) * 64;
    char* base = smem + bo + (which * 2 + h) * T256_REGION + wave * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[which][h][0] + koff), (lptr_t)(base), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[which][h][1] + koff), (lptr_t)(base + 8 * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets -------------------------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int a_base = (wr * 64 + fr) * 128, b_base = (wc * 32 + fr) * 128;
  const int sw0 = ((0 + fq) ^ (fr >> 1)) * 16, sw1 = ((4 + fq) ^ (fr >> 1)) * 16;

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], b0f[2][2], b1f[2][2];
  const bool vecp = vec_path_ok(p);
  const bool interior = vecp && (m0 + 256 <= p.M) && (n0 + 256 <= p.N);
  const bool fold_bias = vecp && (p.epi & ICL_EPI_BIAS);
  // the residual is added LAST, (bias + sum) + r, in every kernel and every tile (interior, edge, any tile shape): the order
  // is part of the batch-invariance contract; the 256x256 kernel reads it as whole rows in its LDS-staged epilogue
  constexpr bool fold_res = false;

  auto read_a = [&](int bo, int h) {
    const char* r = smem + bo + h * T256_REGION + a_base;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *(const bf16x8*)(r + i * 2048 + sw0);
      af[i][1] = *(const bf16x8*)(r + i * 2048 + sw1);
    }
  };
  auto read_b = [&](int bo, int h, bf16x8 (&bf)[2][2]) {
    const char* r = smem + bo + (2 + h) * T256_REGION + b_base;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf[j][0] = *(const bf16x8*)(r + j * 2048 + sw0);
      bf[j][1] = *(const bf16x8*)(r + j * 2048 + sw1);
    }
  };
  // MFMA cluster of one phase, closed by the phase's SECOND barrier.  The two wave groups (wr = 0 / 1) run one barrier
  // apart (see the stagger below), so between two consecutive barriers one group issues its 16 MFMAs while the other
  // issues its LDS reads + LDS-DMA + waits: matrix pipe and LDS/VMEM overlap on every SIMD (2 waves/SIMD, one per group).
  auto mma = [&](f32x4 (&c)[4][2], bf16x8 (&bf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][kk], af[i][kk], c[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // the phase's counted wait (N = LDS-DMA instructions that may stay in flight: 8 = four granules in steady state) + barrier
  auto phase_sync = [&](auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // MODE 0: steady state (every phase stages one granule 5-6 phases ahead, vmcnt(8)); bo / bo ^ T256_BUF = this / the other
  // K-tile buffer.  MODE 1 / 2: K-tiles nk-2 / nk-1: stage-ahead targets past the end of K are NOT issued (no dummy re-loads,
  // no drain before the block retires); each wait lets exactly the instructions issued in the last four phases stay in
  // flight, which shrinks 8 -> 6 -> 4 -> 2 -> 0 as the staging runs dry, so the guarantee "the granule staged four phases
  // ago has landed" is the one of the steady state.
  auto tile = [&](int bo, int t, auto mode) {
    constexpr int MODE = decltype(mode)::value;
    const int bx = bo ^ T256_BUF;
    // P0
    read_a(bo, 0);
    read_b(bo, 0, b0f);
    if constexpr (MODE != 2) stage(bx, 1, 1, t + 1);
    phase_sync(std::integral_constant<int, MODE == 2 ? 2 : 8>{});
    mma(acc[0][0], b0f);
    // P1
    read_b(bo, 1, b1f);
    if constexpr (MODE != 2) stage(bx, 0, 1, t + 1);
    phase_sync(std::integral_constant<int, MODE == 2 ? 0 : 8>{});
    mma(acc[0][1], b1f);
    // P2
    read_a(bo, 1);
    if constexpr (MODE == 0) stage(bo, 0, 0, t + 2);
    phase_sync(std::integral_constant<int, MODE == 0 ? 8 : (MODE == 1 ? 6 : 0)>{});
    mma(acc[1][1], b1f);
    // P3
    if constexpr (MODE == 0) stage(bo, 1, 0, t + 2);
    phase_sync(std::integral_constant<int, MODE == 0 ? 8 : (MODE == 1 ? 4 : 0)>{});
    mma(acc[1][0], b0f);
  };
  using M0_ = std::integral_constant<int, 0>;
  using M1_ = std::integral_constant<int, 1>;
  using M2_ = std::integral_constant<int, 2>;

  // accumulator init = (f32 residual) + bias: pure loads issued BEFORE the prologue's LDS-DMA, first use after it;
  // tile-independent decision (vec_path_ok), edge tiles only add bounds guards
  f32x4 bias_f[2][2];
  if (fold_res) {
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int m = m0 + qa * 128 + wr * 64 + i * 16 + fr, n = n0 + qb * 128 + wc * 32 + j * 16 + fq * 4;
            if (interior || (m < p.M && n < p.N))
              acc[qa][qb][i][j] = *(const f32x4*)((const float*)p.R + (int64_t)z * p.sR + (int64_t)m * p.ldr + n);
          }
  }
  if (fold_bias) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + qb * 128 + wc * 32 + j * 16 + fq * 4;
        bias_f[qb][j] = (interior || n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  // prologue: A0(0) B0(0) B1(0) A1(0) A0(1) B0(1), then the uniform wait
  __builtin_amdgcn_sched_barrier(0);
  stage(0, 0, 0, 0);
  stage(0, 1, 0, 0);
  stage(0, 1, 1, 0);
  stage(0, 0, 1, 0);
  stage(T256_BUF, 0, 0, 1);
  stage(T256_BUF, 1, 0, 1);
  __builtin_amdgcn_sched_barrier(0);
  if (fold_bias) {   // first use of the pre-loaded operands: ONE wait covers loads and prologue
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[qa][qb][i][j] = acc[qa][qb][i][j] + bias_f[qb][j];
  }
  phase_sync(std::integral_constant<int, 8>{});
  if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger: group 1 runs one barrier behind group 0 (hazard analysis in DESIGN.md §4)
  // K-tiles 0 .. nk-3 in steady state (compile-time buffers), then the two tail K-tiles on a run-time buffer offset (the host
  // sends K < 128 to the other tiles: nk >= 2 here)
  const int n_steady = nk - 2;
  int t = 0;
  for (; t + 1 < n_steady; t += 2) {
    tile(0, t, M0_{});
    tile(T256_BUF, t + 1, M0_{});
  }
  if (t < n_steady) {
    tile(0, t, M0_{});
    ++t;
  }
  const int bo = (t & 1) ? T256_BUF : 0;
  tile(bo, t, M1_{});
  tile(bo ^ T256_BUF, t + 1, M2_{});
  if (wr == 0) __builtin_amdgcn_s_barrier();   // re-balance the barrier count of the two groups

  // ---- interior tiles: the C tile leaves through LDS ---------------------------------------------------------------
  // An MFMA fragment gives a lane 4 consecutive columns of ONE row, so direct stores are 8-B (bf16) pieces in 32-B row
  // segments: 32 partial-line stores per thread, measured at 6.4 us per tile (21 % of a K = 1280 tile, 6 % at K = 4096; the
  // same kernel without its stores runs 1.39 PF/s at K = 1280).  The K-tile buffers are dead after the main loop, so each
  // 128-row half of the tile is written to LDS in its output type (row pitch + 16 B: conflict-free for both the fragment
  // writes and the row reads) and read back as whole rows, 16 B per lane, full cache lines per wave-instruction.
  const int es_out = p.out_dtype == ICL_BF16 ? 2 : 4;
  const bool rows16 = (((uintptr_t)p.C | (uintptr_t)(p.ldc * es_out) | (uintptr_t)(p.sC * es_out)) & 15) == 0;   // whole rows in 16-B pieces
  const bool has_res = p.epi & ICL_EPI_RESIDUAL;
  const bool res_rows = has_res && p.res_dtype == ICL_F32 && p.out_dtype == ICL_F32 && !(p.epi & ICL_EPI_SWIGLU) &&
                        (((uintptr_t)p.R | (uintptr_t)(p.ldr * 4) | (uintptr_t)(p.sR * 4)) & 15) == 0;
  if (ROPE || (interior && rows16 && (!has_res || res_rows))) {   // ROPE: the host has checked the layout; row-masked M edge
    const bool swiglu = p.epi & ICL_EPI_SWIGLU;
    const bool obf = p.out_dtype == ICL_BF16;
    const int out_cols = swiglu ? 128 : 256;
    const int es = obf ? 2 : 4;
    const int pitch = out_cols * es + 16;                        // bytes per staged row
    const int chunks_per_row = out_cols * es / 16;               // 16-B pieces per row: 16 | 32 | 64
    const int64_t c_col0 = swiglu ? (n0 >> 1) : n0;
    const bool one_round = ROPE || obf;                          // a bf16 tile fits whole: two barriers instead of four (two rounds
                                                                 // under GELU, to drain stores behind the second half's VALU work: no gain)
#pragma unroll
    for (int qa = 0; qa < 2; ++qa) {
      if (qa == 0 || !one_round) __syncthreads();                // K-tile reads (qa = 0) / the previous half's row reads are done
      const int row_off = one_round ? qa * 128 : 0;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row_off + wr * 64 + i * 16 + fr;
          if (swiglu) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(acc[qa][qb][i][0][r]) * acc[qa][qb][i][1][r];
            const int col = qb * 64 + wc * 16 + fq * 4;
            char* dst = smem + row * pitch + col * es;
            if (obf) *(u32x2*)dst = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            else *(f32x4*)dst = v;
          } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f32x4 v = acc[qa][qb][i][j];
              if (p.epi & ICL_EPI_GELU) {
            v = gelu_erf4(v);
              }
              const int col = qb * 128 + wc * 32 + j * 16 + fq * 4;
              char* dst = smem + row * pitch + col * es;
              if (obf) *(u32x2*)dst = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
              else *(f32x4*)dst = v;
            }
          }
        }
      if (one_round && qa == 0) continue;
      __syncthreads();
      if constexpr (ROPE) {   // (not a return: an early exit inside the qa loop keeps hipcc from unrolling it -> acc in scratch)
        rope_rows(p, rf, smem, pitch, m0, n0, tid);
        continue;
      }
      char* cbase = (char*)p.C + ((int64_t)z * p.sC + (int64_t)(m0 + (one_round ? 0 : qa * 128)) * p.ldc + c_col0) * es;
      const int n_chunks = (one_round ? 256 : 128) * chunks_per_row;   // a multiple of the 512 threads
      if (has_res) {   // f32 residual stream: whole-row 16-B loads, all of a thread's 16 issued before the first use
        const char* rbase = (const char*)p.R + ((int64_t)z * p.sR + (int64_t)(m0 + qa * 128) * p.ldr + n0) * 4;
        f32x4 rr[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int c = tid + it * 512, row = c >> 6, cc = c & 63;
          rr[it] = *(const f32x4*)(rbase + (int64_t)row * p.ldr * 4 + cc * 16);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int c = tid + it * 512, row = c >> 6, cc = c & 63;
          const f32x4 v = *(const f32x4*)(smem + row * pitch + cc * 16) + rr[it];
          *(f32x4*)(cbase + (int64_t)row * p.ldc * 4 + cc * 16) = v;
        }
      } else {
        for (int c = tid; c < n_chunks; c += 512) {
          const int row = c / chunks_per_row, cc = c - row * chunks_per_row;
          const u32x4 v = *(const u32x4*)(smem + row * pitch + cc * 16);
          *(u32x4*)(cbase + (int64_t)row * p.ldc * es + cc * 16) = v;
        }
      }
    }
    return;
  } else if (interior) {
    const bool late_res = (p.epi & ICL_EPI_RESIDUAL) && !fold_res;
#pragma unroll
    for (int qa = 0; qa < 2; ++qa) {
      const int mb = m0 + qa * 128 + wr * 64 + fr;
      if (p.epi & ICL_EPI_SWIGLU) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(acc[qa][qb][i][0][r]) * acc[qa][qb][i][1][r];
            store_out4(p, (int64_t)z * p.sC + (int64_t)(mb + i * 16) * p.ldc + ((n0 + qb * 128 + wc * 32) >> 1) + fq * 4, v);
          }
        continue;
      }
      f32x4 rv[2][4][2];
      if (late_res) {   // one batch of 16 independent loads per half tile
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              rv[qb][i][j] = load_res4(p, (int64_t)z * p.sR + (int64_t)(mb + i * 16) * p.ldr + n0 + qb * 128 + wc * 32 + j * 16 + fq * 4);
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4 v = acc[qa][qb][i][j];
            if (p.epi & ICL_EPI_GELU) {
            v = gelu_erf4(v);
            }
            if (late_res) v = v + rv[qb][i][j];
            store_out4(p, (int64_t)z * p.sC + (int64_t)(mb + i * 16) * p.ldc + n0 + qb * 128 + wc * 32 + j * 16 + fq * 4, v);
          }
    }
    return;
  }

  // ---- edge tiles: same arithmetic, bounds-checked stores; operands already folded are not re-applied ----------------
  GemmParams q = p;
  if (fold_bias) q.epi &= ~ICL_EPI_BIAS;
  if (fold_res) q.epi &= ~ICL_EPI_RESIDUAL;
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + qa * 128 + wr * 64 + i * 16 + fr;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int nt = n0 + qb * 128 + wc * 32;
        if (p.epi & ICL_EPI_SWIGLU) {
          epi_store_swiglu(q, z, m, nt, fq * 4, acc[qa][qb][i][0], acc[qa][qb][i][1]);
        } else {
          epi_store4(q, z, m, nt + fq * 4, acc[qa][qb][i][0]);
          epi_store4(q, z, m, nt + 16 + fq * 4, acc[qa][qb][i][1]);
        }
      }
    }
}

int launch_tile256(GemmParams& p, int batch, hipStream_t stream, const RopeFuse* rope = nullptr) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T256_SMEM);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm256_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T256_SMEM);
    if (e != hipSuccess) {
      icl_set_error("icl_gemm_bf16: hipFuncSetAttribute(%d) failed: %s", T256_SMEM, hipGetErrorString(e));
      return ICL_ELAUNCH;
    }
    attr_set = true;
  }
  const dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
  if (rope)
    hipLaunchKernelGGL(gemm256_bf16_kernel<true>, grid, dim3(512), T256_SMEM, stream, p, *rope);
  else
    hipLaunchKernelGGL(gemm256_bf16_kernel<false>, grid, dim3(512), T256_SMEM, stream, p, RopeFuse{});
  ICL_CHECK_LAUNCH("icl_gemm_bf16(256)");
  return ICL_OK;
}


// =================================================================================================================
// Skinny GEMM for decode (M <= 64): the weight matrix is streamed ONCE, straight from HBM into VGPRs (no LDS round
// trip, no barriers in the stream: guide §5 table row "GEMV / M <= 16 decode"), 16 B per lane, several KiB in flight per
// wave.  Block = 8 waves = one 16*NT-column slab of the output; the waves split K eight ways and combine their 16x16
// f32 partial tiles through LDS (in-block split-K: deterministic, no workspace, no second launch).  The activations
// (M x K, <= 0.7 MB) are re-read by every block from L2.  MB = 16-row blocks of M, NT = 16-column tiles per block
// (2 for the SwiGLU epilogue so a gate block and its up block meet in one lane).
// =================================================================================================================
template <int MB, int NT, int U, bool PACKED>   // PACKED: W is the decode-packed copy (tile 6): a wave-load is 1 KB contiguous
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GemmParams p) {
  __shared__ float red[8][NT][MB][256];   // [wave][n-tile][m-block][lane*4 + r]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int steps = p.K >> 5;                       // 32-wide k-steps
  const int s0 = (int)(((int64_t)wave * steps) >> 3), s1 = (int)(((int64_t)(wave + 1) * steps) >> 3);

  const __bf16* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    wp[t] = PACKED ? p.W + (int64_t)min((n0 >> 4) + t, ((p.N + 15) >> 4) - 1) * (p.K >> 5) * 512 + lane * 8
                   : p.W + (int64_t)min(n0 + t * 16 + fr, p.N - 1) * p.ldw + fq * 8;
  constexpr int WSTEP = PACKED ? 512 : 32;   // elements between consecutive 32-wide k-steps of one n-tile
  const __bf16* ap[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) ap[b] = p.A + (int64_t)min(b * 16 + fr, p.M - 1) * p.lda + fq * 8;

  f32x4 acc[NT][MB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int s = s0;
  for (; s + U <= s1; s += U) {
    bf16x8 wf[U][NT], af[U][MB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[u][t] = *(const bf16x8*)(wp[t] + (int64_t)(s + u) * WSTEP);
#pragma unroll
      for (int b = 0; b < MB; ++b) af[u][b] = *(const bf16x8*)(ap[b] + (int64_t)(s + u) * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < MB; ++b)
          acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][t], af[u][b], acc[t][b], 0, 0, 0);
  }
  for (; s < s1; ++s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 wf = *(const bf16x8*)(wp[t] + (int64_t)s * WSTEP);
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        const bf16x8 af = *(const bf16x8*)(ap[b] + (int64_t)s * 32);
        acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[t][b], 0, 0, 0);
      }
    }
  }
  // ---- in-block split-K combine: fixed order (wave 0..7) -> bitwise reproducible ---------------------------------
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int b = 0; b < MB; ++b) *(f32x4*)&red[wave][t][b][lane * 4] = acc[t][b];
  __syncthreads();
  // 8 waves share the NT*MB output fragments
  for (int f = wave; f < NT * MB; f += 8) {
    const int t = f / MB, b = f - t * MB;
    f32x4 v = *(const f32x4*)&red[0][t][b][lane * 4];
#pragma unroll
    for (int w = 1; w < 8; ++w) v = v + *(const f32x4*)&red[w][t][b][lane * 4];
    if (!(p.epi & ICL_EPI_SWIGLU)) epi_store4(p, 0, b * 16 + fr, n0 + t * 16 + fq * 4, v);
    else *(f32x4*)&red[0][t][b][lane * 4] = v;
  }
  if (p.epi & ICL_EPI_SWIGLU) {
    if constexpr (NT == 2) {
      __syncthreads();
      for (int b = wave; b < MB; b += 8) {
        const f32x4 g = *(const f32x4*)&red[0][0][b][lane * 4], u = *(const f32x4*)&red[0][1][b][lane * 4];
        epi_store_swiglu(p, 0, b * 16 + fr, n0, fq * 4, g, u);
      }
    }
  }
}

template <int MB, int NT, int U>
int launch_skinny(GemmParams& p, hipStream_t stream, bool packed) {
  const int blocks = (p.N + 16 * NT - 1) / (16 * NT);
  if (packed) hipLaunchKernelGGL((gemm_skinny_kernel<MB, NT, U, true>), dim3(blocks), dim3(512), 0, stream, p);
  else        hipLaunchKernelGGL((gemm_skinny_kernel<MB, NT, U, false>), dim3(blocks), dim3(512), 0, stream, p);
  ICL_CHECK_LAUNCH("icl_gemm_bf16(skinny)");
  return ICL_OK;
}

// =================================================================================================================
// Decode GEMM for 64 < M <= 128 (tile id 5).  At this size a decode GEMM sits on the ridge: 2*128 FLOP per weight
// byte, i.e. the 13.5 GB of Llama-7B weights cost about the same on the matrix pipe as on HBM, and what decides is how
// much a CU has to move per weight byte: through its vector-memory path (measured ceiling here ~55-68 GB/s per CU) and
// out of LDS (128 B/clk).  The 64x64 LDS tile moves 3 bytes per weight byte (W once, the activation slice twice as
// much again from L2) and reads 4 bytes of LDS; this kernel moves 2 and reads 4 — but of a tile twice as wide:
//   * one block = ALL (<= 128) rows x 128 columns.  Its 8 waves are 4 column groups (32 columns = two 16-wide n-tiles,
//     so a SwiGLU gate block and its up block meet in one lane) x 2 K-halves: wave (wc, wk) takes the 32-wide k-step
//     wk of every 64-wide K-tile.  Every weight byte is loaded by exactly one wave, and each wave reads only its half
//     of the staged activations (splitting the columns 8 ways instead would have every wave read all of them: LDS-bound
//     at 0.59 us per K-tile).  The two K-halves meet once, through LDS, after the loop (fixed order: even + odd);
//   * W goes HBM -> VGPR directly in MFMA operand order from the decode-packed copy (icl_pack_decode_weights): per
//     16-row n-tile a K-long stream of 1-KB pieces, one per 32-wide k-step, so a wave-load is 1 KB contiguous, lane l
//     at byte 16*l.  (Row-major W read in operand order is 16 rows x 64 B per wave-load = 64 separate L1 accesses; that
//     pattern capped the first version of this kernel and caps the skinny kernel.)  DEPTH K-tiles deep in registers;
//   * A (shared by all waves) is staged by LDS-DMA into a DEPTH+1 ring, one barrier per K-tile;
//   * loads past the end of the K range are clamped to its last tile, so the in-flight count (vmcnt) is the same in
//     every iteration and no tail code exists;
//   * split-K over grid.z with the same workspace slabs + reduce kernel as the other tiles; with split_k == 1 the bias
//     is folded into the accumulator init like everywhere else.
template <int DEPTH, int MT>   // DEPTH: K-tiles of W in registers (and of A in LDS, + 1 being read); MT: 16-row tiles (8 | 4)
__global__ __launch_bounds__(512) void gemm_m128_kernel(GemmParams p) {
  constexpr int NI = 2, BN = 4 * NI * 16, A_INSTR = MT / 4, A_STAGE = MT * 16 * 128, NSA = DEPTH + 1;
  constexpr int G = A_INSTR + NI;   // VMEM loads per K-tile per lane
  static_assert(MT == 16 || MT == 8 || MT == 4, "256-, 128- or 64-row blocks");
  static_assert(NSA * A_STAGE >= 4 * MT * NI * 1024, "the K-half exchange reuses the A ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, wk = wave >> 2;
  const int n0 = blockIdx.x * BN, z = blockIdx.z;
  const int nk = p.K >> 6;
  int kt0 = 0, kt1 = nk;
  if (p.split_k > 1) {
    kt0 = (int)(((int64_t)z * nk) / p.split_k);
    kt1 = (int)(((int64_t)(z + 1) * nk) / p.split_k);
  }
  const int nt = kt1 - kt0;
  const int fr = lane & 15, fq = lane >> 4;

  const __bf16* ga[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (j * 8 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    ga[j] = p.A + (int64_t)min(row, p.M - 1) * p.lda + (int64_t)kt0 * 64 + chunk * 8;
  }
  const int n_tiles16 = (p.N + 15) >> 4;
  const __bf16* gw[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
    gw[j] = p.W + ((int64_t)min((n0 >> 4) + wc * NI + j, n_tiles16 - 1) * (p.K >> 5) + (int64_t)kt0 * 2 + wk) * 512 + lane * 8;

  auto stage_a = [&](int t, int slot) {
    const int tc = min(t, nt - 1);
    char* base = smem + slot * A_STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(ga[j] + (int64_t)tc * 64), (lptr_t)(base + j * 8 * 1024), 16, 0, 0);
  };
  bf16x8 wf[DEPTH][NI];
  auto load_w = [&](bf16x8 (&w)[NI], int t) {
    const int tc = min(t, nt - 1);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      // raw loads: hipcc's waitcnt pass answers a loop-carried register load next to LDS-DMA with vmcnt(0) at the loop
      // header (the whole prefetch drained once per unrolled body); the counted wait in tile() covers these instead
      const __bf16* src = gw[j] + (int64_t)tc * 1024;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[j]) : "v"(src) : "memory");
    }
  };
  const int a_off = fr * 128 + (((wk * 4 + fq) ^ (fr >> 1)) * 16);

  f32x4 acc[MT][NI];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias folded into the accumulator init exactly like the LDS tiles do ((bias + sum), residual last); K-half 0 carries it
  const bool fold_bias = p.split_k == 1 && vec_path_ok(p) && (p.epi & ICL_EPI_BIAS);
  if (fold_bias && wk == 0) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + (wc * NI + j) * 16 + fq * 4;
      const f32x4 b4 = n < p.N ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = b4;
    }
  }

  int slot = 0;                              // ring slot of the K-tile being computed
  auto tile = [&](bf16x8 (&w)[NI], int t) {
    // the K-tile t operands are the oldest loads in flight; tiles t+1 .. t+DEPTH-1 (G loads each) may still be
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * G) : "memory");
    __builtin_amdgcn_s_barrier();            // A(t) of every wave has landed; every wave is done reading A(t-1)
    __builtin_amdgcn_sched_barrier(0);
    stage_a(t + DEPTH, slot == 0 ? NSA - 1 : slot - 1);   // into the ring slot of t-1
    const char* a_s = smem + slot * A_STAGE + a_off;
    slot = slot == NSA - 1 ? 0 : slot + 1;
    constexpr int MH = MT > 8 ? 8 : MT;      // activation fragments held at a time (256-row blocks take two passes: registers)
#pragma unroll
    for (int ih = 0; ih < MT / MH; ++ih) {
      bf16x8 af[MH];
#pragma unroll
      for (int i = 0; i < MH; ++i) af[i] = *(const bf16x8*)(a_s + (ih * MH + i) * 16 * 128);
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[ih * MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], af[i], acc[ih * MH + i][j], 0, 0, 0);
      if (MT > MH) __builtin_amdgcn_sched_barrier(0);      // keep the second pass's fragment reads behind the first pass's MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);
    load_w(w, t + DEPTH);                    // this register set is free again
    __builtin_amdgcn_sched_barrier(0);
  };

#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    stage_a(d, d);
    load_w(wf[d], d);
    __builtin_amdgcn_sched_barrier(0);
  }
  int t = 0;
  for (; t + DEPTH <= nt; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) tile(wf[d], t + d);
  }
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (t + d < nt) tile(wf[d], t + d);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped over-issue must not outlive the block's LDS
  // ... nor its registers: to the compiler a raw load's result is there at once, so a result nobody reads is a free
  // register while the load is still in flight.  Reading every set here keeps all of them allocated up to the wait.
#pragma unroll
  for (int s_ = 0; s_ < DEPTH; ++s_)
#pragma unroll
    for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(wf[s_][j]));

  // ---- the two K-halves meet: odd half -> LDS (the A ring is dead), even half adds it on top and stores ---------------
  __syncthreads();
  char* xbase = smem + wc * (MT * NI * 1024) + lane * 16;
  if (wk == 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) *(f32x4*)(xbase + (i * NI + j) * 1024) = acc[i][j];
  }
  __syncthreads();
  if (wk == 1) return;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = acc[i][j] + *(const f32x4*)(xbase + (i * NI + j) * 1024);

  // ---- epilogue: acc[i][j][r] = C[m][n], m = i*16 + fr, n = n0 + (wc*NI + j)*16 + fq*4 + r ---------------------------
  GemmParams q = p;
  if (fold_bias) q.epi &= ~ICL_EPI_BIAS;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = i * 16 + fr;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nt_ = n0 + (wc * NI + j) * 16;
      if (p.split_k > 1) {
        epi_store_partial(p, z, m, nt_ + fq * 4, acc[i][j]);
      } else if (p.epi & ICL_EPI_SWIGLU) {
        if ((j & 1) == 0) epi_store_swiglu(q, 0, m, nt_, fq * 4, acc[i][j], acc[i][(j + 1) % NI]);
      } else {
        epi_store4(q, 0, m, nt_ + fq * 4, acc[i][j]);
      }
    }
  }
}

// row-major W [N][ldw] -> decode-packed: piece (n-tile, k-step, lane = fq*16 + fr) holds W[16*nt + fr][32*ks + 8*fq .. +8]
__global__ __launch_bounds__(256) void pack_decode_w_kernel(const unsigned short* W, int64_t ldw, int N, int K, u32x4* out) {
  const int64_t pieces = (int64_t)((N + 15) >> 4) * (K >> 5) * 64;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < pieces; q += (int64_t)gridDim.x * 256) {
    const int l = (int)(q & 63);
    const int64_t blk = q >> 6;
    const int ks = (int)(blk % (K >> 5));
    const int64_t row = (blk / (K >> 5)) * 16 + (l & 15);
    out[q] = row < N ? *(const u32x4*)(W + row * ldw + ks * 32 + (l >> 4) * 8) : u32x4{0u, 0u, 0u, 0u};
  }
}

template <int DEPTH, int MT>
int launch_m128(GemmParams& p, hipStream_t stream) {
  constexpr int BN = 128, SMEM = (DEPTH + 1) * MT * 16 * 128;
  auto kern = gemm_m128_kernel<DEPTH, MT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      icl_set_error("icl_gemm_bf16: hipFuncSetAttribute(%d) failed: %s", SMEM, hipGetErrorString(e));
      return ICL_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((p.N + BN - 1) / BN, 1, p.split_k), dim3(512), SMEM, stream, p);
  ICL_CHECK_LAUNCH("icl_gemm_bf16(m128)");
  return ICL_OK;
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


// Tile choice (measured on MI355X, profiles/r01_*):  64x64 for skinny / under-filled problems (and all split-K
// calls), the 256x256 rolling pipeline when K is deep enough to amortise its fill/drain and the tile grid
// quantises well onto the CUs (one 256x256 block per CU; K >= 1024 after the accumulator-init epilogue), otherwise the 128x128 double-buffered kernel.
extern "C" int icl_gemm_select_tile(int32_t M, int32_t N, int32_t K, int32_t batch, int32_t split_k) {
  const int64_t t128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (split_k > 1 || M <= 64 || t128 < 256) return 2;
  if (N <= 64) return 2;    // a wider tile multiplies padding columns: BEATs grouped pos-conv (N = 48) 376 us on 256x256, 215 on 64x64
  if (N <= 128) return 1;
  static int ncu = 0;
  if (ncu <= 0) {
    ncu = icl_device_cu_count();
    if (ncu <= 0) ncu = 256;
  }
  const int64_t t256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256) * batch;
  const int64_t rounds = (t256 + ncu - 1) / ncu;
  const double eff = (double)t256 / (double)(rounds * ncu);
  if (K >= 768 && eff >= 0.8) return 3;   // measured at micro-batch 128: the 256x256 tile wins from K = 768 (BEATs) upwards
  return 1;
}

static int gemm_impl(const icl_gemm_args* a, void* stream_, const RopeFuse* rope) {
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
  if (tile == 0) tile = icl_gemm_select_tile(a->M, a->N, a->K, a->batch, a->split_k);
  if (tile == 3 && a->K < 128) tile = 1;   // the 256x256 pipeline peels two K-tiles; same arithmetic on the 128x128 tile
  if (rope) {
    ICL_CHECK_ARG(tile == 3, "icl_gemm_rope_kv_bf16: only the 256x256 tile fuses RoPE (this problem resolves to tile %d; "
                             "use icl_gemm_bf16 + icl_rope_kv_bf16)", tile);
    ICL_CHECK_ARG(a->split_k == 1, "icl_gemm_rope_kv_bf16: split_k must be 1");
    return launch_tile256(p, 1, stream, rope);
  }
  int rc;
  if (tile == 1)
    rc = launch_tile<2, 2, 4, 4>(p, a->batch, stream);
  else if (tile == 2)
    rc = launch_tile<2, 2, 2, 2>(p, a->batch, stream);
  else if (tile == 3) {
    ICL_CHECK_ARG(a->split_k == 1, "icl_gemm_bf16: the 256x256 tile does not support split_k");
    rc = launch_tile256(p, a->batch, stream);
  } else if (tile == 4 || tile == 6) {   // 6: the same kernel on the decode-packed copy of W
    ICL_CHECK_ARG(a->M <= 64 && a->batch == 1, "icl_gemm_bf16: the skinny kernel needs M <= 64 and batch == 1");
    p.split_k = 1;   // K is split inside the block
    const bool sw = a->epilogue & ICL_EPI_SWIGLU, pk = tile == 6;
    const int mb = (a->M + 15) / 16;
    if (sw) rc = mb <= 1 ? launch_skinny<1, 2, 4>(p, stream, pk) : mb == 2 ? launch_skinny<2, 2, 2>(p, stream, pk) : launch_skinny<4, 2, 2>(p, stream, pk);
    else    rc = mb <= 1 ? launch_skinny<1, 1, 8>(p, stream, pk) : mb == 2 ? launch_skinny<2, 1, 4>(p, stream, pk) : launch_skinny<4, 1, 2>(p, stream, pk);
    return rc;
  } else if (tile == 5) {
    ICL_CHECK_ARG(a->M <= 256 && a->batch == 1, "icl_gemm_bf16: the decode tile needs M <= 256 and batch == 1");
    // 256-column blocks when that still gives every CU most of a block, 128-column blocks otherwise
    // depth 3 / 4 / 6 measured alike: the CU's vector-memory path is the limit, not latency.  64-row blocks stage half the A bytes
    // 256-row blocks (two micro-batches decoded together): a weight byte then serves twice the rows
    rc = a->M <= 64 ? launch_m128<3, 4>(p, stream) : a->M <= 128 ? launch_m128<3, 8>(p, stream) : launch_m128<3, 16>(p, stream);
  } else {
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

extern "C" int icl_gemm_bf16(const icl_gemm_args* a, void* stream) { return gemm_impl(a, stream, nullptr); }

extern "C" int icl_gemm_rope_kv_bf16(const icl_gemm_args* a, int64_t k_off, int64_t v_off, const float* cosT,
                                     const float* sinT, const int32_t* pos, const int32_t* seq_ids, void* kcache,
                                     void* vcache, int32_t n_heads, int32_t head_dim, int32_t max_len, int32_t kv_rows_to_c,
                                     void* stream) {
  ICL_CHECK_ARG(a != nullptr && cosT && sinT && pos, "icl_gemm_rope_kv_bf16: NULL pointer");
  ICL_CHECK_ARG(head_dim == 128, "icl_gemm_rope_kv_bf16: head_dim=%d (the fused epilogue is built for 128)", head_dim);
  ICL_CHECK_ARG(n_heads > 0 && k_off == (int64_t)n_heads * 128 && v_off == 2 * k_off && a->N == 3 * k_off && k_off % 256 == 0,
                "icl_gemm_rope_kv_bf16: need q|k|v blocks of n_heads*128 columns each, a multiple of 256 (N=%d k_off=%lld v_off=%lld)",
                a->N, (long long)k_off, (long long)v_off);
  ICL_CHECK_ARG(a->batch == 1 && a->out_dtype == ICL_BF16 && (a->epilogue & ~ICL_EPI_BIAS) == 0,
                "icl_gemm_rope_kv_bf16: batch 1, bf16 output, bias-only epilogue");
  ICL_CHECK_ARG(a->C && ((uintptr_t)a->C & 15) == 0 && a->ldc % 8 == 0, "icl_gemm_rope_kv_bf16: C must be 16-byte aligned, ldc %% 8 == 0");
  ICL_CHECK_ARG(!a->bias || ((uintptr_t)a->bias & 15) == 0, "icl_gemm_rope_kv_bf16: bias must be 16-byte aligned");
  ICL_CHECK_ARG(((uintptr_t)cosT & 15) == 0 && ((uintptr_t)sinT & 15) == 0, "icl_gemm_rope_kv_bf16: cos/sin misaligned");
  ICL_CHECK_ARG((kcache == nullptr) == (vcache == nullptr), "icl_gemm_rope_kv_bf16: kcache and vcache must both be set or both NULL");
  if (kcache) {
    ICL_CHECK_ARG(seq_ids && max_len > 0, "icl_gemm_rope_kv_bf16: cache append needs seq_ids and max_len");
    ICL_CHECK_ARG(((uintptr_t)kcache & 15) == 0 && ((uintptr_t)vcache & 15) == 0, "icl_gemm_rope_kv_bf16: cache misaligned");
  }
  ICL_CHECK_ARG(kv_rows_to_c || kcache, "icl_gemm_rope_kv_bf16: kv_rows_to_c = 0 needs a cache to hold k / v");
  RopeFuse rf;
  rf.kv_rows_to_c = kv_rows_to_c;
  rf.cosT = cosT; rf.sinT = sinT; rf.pos = pos; rf.seq_ids = seq_ids;
  rf.kc = (unsigned short*)kcache; rf.vc = (unsigned short*)vcache;
  rf.k_off = (int)k_off; rf.v_off = (int)v_off; rf.H = n_heads; rf.max_len = max_len;
  return gemm_impl(a, stream, &rf);
}

extern "C" int icl_pack_decode_weights(const void* W, int64_t ldw, int32_t N, int32_t K, void* out, void* stream) {
  ICL_CHECK_ARG(W && out && N > 0 && K > 0, "icl_pack_decode_weights: bad arguments");
  ICL_CHECK_ARG(K % 64 == 0 && ldw % 8 == 0 && ldw >= K, "icl_pack_decode_weights: K=%d must be a multiple of 64, ldw=%lld a multiple of 8", K, (long long)ldw);
  ICL_CHECK_ARG(((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0, "icl_pack_decode_weights: misaligned pointer");
  const int64_t pieces = (int64_t)((N + 15) >> 4) * (K >> 5) * 64;
  const int blocks = (int)std::min<int64_t>((pieces + 255) / 256, 65535);
  hipLaunchKernelGGL(pack_decode_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)W, ldw, N, K, (u32x4*)out);
  ICL_CHECK_LAUNCH("icl_pack_decode_weights");
  return ICL_OK;
}
