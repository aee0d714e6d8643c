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
#include "gemm_common.h"

using namespace iclg;

namespace {


// NST = K-tiles of A and W resident in LDS per block.  2: the launch has blocks to spare and occupancy hides the load latency
// (64x64: 32 KiB, five blocks per CU).  Deep (4 x 16 KiB for the 64x64 tile, 3 x 32 KiB for the 128x128 tile): every block of the
// launch is resident at once and each keeps NST - 1 K-tiles of loads in flight.  Same MFMA sequence either way (bit-identical).
#define TILE_STAGES_DEEP(BM_, BN_) (((BM_) + (BN_)) <= 128 ? 4 : 3)

template <int WAVES_M, int WAVES_N, int MI, int NI, int NST>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  constexpr int BM = WAVES_M * MI * 16, BN = WAVES_N * NI * 16;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / 32, W_INSTR = BN / 32;  // glds wave-instructions per wave
  constexpr int G_STAGE = A_INSTR + W_INSTR;            // VMEM instructions per wave and K-tile
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

  auto stage = [&](int buf_off, int kt) {      // buf_off: byte offset of the ring slot
    char* base = smem + buf_off;
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
    // NST-deep ring of K-tiles: NST - 1 tiles of LDS-DMA in flight while one is consumed, counted waits (loads past the end of
    // the K range are clamped to its last tile so that the count is the same in every iteration), one barrier per K-tile.
    // With two stages and a full drain per K-tile a block spent most of each tile waiting for its one load in flight: the
    // launches that cannot fill the chip with blocks (prefill of a few sequences, BEATs / Q-Former shapes) were latency-bound.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_) stage(s_ * BUF, min(kt0 + s_, kt1 - 1));
    __builtin_amdgcn_sched_barrier(0);
    if (fold_bias) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = acc[i][j] + bias_f[j];
    }
    int cur = 0, fill = (NST - 1) * BUF;        // byte offsets: slot being consumed, slot to refill (= the one consumed last)
    for (int kt = kt0; kt < kt1; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * G_STAGE) : "memory");   // K-tile kt has landed (this wave's part)
      __syncthreads();                          // ... everyone's part; and everyone is done reading the buffer of K-tile kt-1
      if (NST > 2) __builtin_amdgcn_sched_barrier(0);
      if (NST > 2 || kt + 1 < kt1)              // two stages wait for vmcnt(0) anyway: no clamped over-issue needed there
        stage(fill, min(kt + NST - 1, kt1 - 1));
      const char* a_s = smem + cur;
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
      fill = cur;
      cur = cur == (NST - 1) * BUF ? 0 : cur + BUF;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped over-issue must not outlive the block's LDS
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

// sum of `split` slabs at `src` (stride `slab` floats) in slab order, four loads in flight at a time: the adds keep their order
// (bit-identical to the one-load-at-a-time loop), the loads no longer wait for each other (a row-per-block reduction runs one
// wave per SIMD: nothing else hides a ~2 us round trip per slab)
__device__ __forceinline__ f32x4 sum_slabs4(const float* src, int64_t slab, int split) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= split; s += 4) {
    const f32x4 a = *(const f32x4*)(src + (s + 0) * slab), b = *(const f32x4*)(src + (s + 1) * slab);
    const f32x4 c = *(const f32x4*)(src + (s + 2) * slab), d = *(const f32x4*)(src + (s + 3) * slab);
    v = v + a;
    v = v + b;
    v = v + c;
    v = v + d;
  }
  for (; s < split; ++s) v = v + *(const f32x4*)(src + s * slab);
  return v;
}

// split-K reduction + epilogue.  Vector form (every layout 4-element aligned — the decode GEMMs): one thread per FOUR consecutive
// output columns, 16-B loads of the slabs / residual and one 8- or 16-B store; each element still sums its slabs in slab order,
// then bias, activation, residual — the scalar form's arithmetic, bit for bit (round 4: 10.9 us per call at 256 x 12288 x 4 slabs
// with 4-B loads).  Scalar form: everything else.
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p) {
  const bool swiglu = p.epi & ICL_EPI_SWIGLU;
  const int Nout = swiglu ? p.N / 2 : p.N;
  const int64_t total = (int64_t)p.M * Nout;
  const int64_t slab = (int64_t)p.M * p.N;
  const bool vec = (p.N & 31) == 0 && (p.ldc & 3) == 0 && (!(p.epi & ICL_EPI_RESIDUAL) || ((p.ldr & 3) == 0 && p.res_dtype == ICL_F32)) &&
                   (((uintptr_t)p.ws | (uintptr_t)p.C) & 15) == 0 && (!(p.epi & ICL_EPI_BIAS) || ((uintptr_t)p.bias & 15) == 0) &&
                   (!(p.epi & ICL_EPI_RESIDUAL) || ((uintptr_t)p.R & 15) == 0);
  if (vec) {
    const int nq = Nout >> 2;
    const int64_t total4 = (int64_t)p.M * nq;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * blockDim.x) {
      const int m = (int)(idx / nq), c = (int)(idx % nq) * 4;
      f32x4 v;
      if (swiglu) {
        const int ng = (c >> 4) * 32 + (c & 15), nu = ng + 16;      // 4 consecutive outputs stay inside one 16-column gate block
        f32x4 g = sum_slabs4(p.ws + (int64_t)m * p.N + ng, slab, p.split_k);
        f32x4 u = sum_slabs4(p.ws + (int64_t)m * p.N + nu, slab, p.split_k);
        if (p.epi & ICL_EPI_BIAS) {
          g = g + *(const f32x4*)(p.bias + ng);
          u = u + *(const f32x4*)(p.bias + nu);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(g[r]) * u[r];
      } else {
        v = sum_slabs4(p.ws + (int64_t)m * p.N + c, slab, p.split_k);
        if (p.epi & ICL_EPI_BIAS) v = v + *(const f32x4*)(p.bias + c);
        if (p.epi & ICL_EPI_GELU) v = gelu_erf4(v);
        if (p.epi & ICL_EPI_RESIDUAL) v = v + *(const f32x4*)((const float*)p.R + (int64_t)m * p.ldr + c);
      }
      const int64_t coff = (int64_t)m * p.ldc + c;
      if (p.out_dtype == ICL_BF16) *(u32x2*)((unsigned short*)p.C + coff) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      else *(f32x4*)((float*)p.C + coff) = v;
    }
    return;
  }
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



// split-K reduction + f32 residual + the NEXT RMSNorm in one pass (decode: the o_proj / down_proj GEMMs are each followed by
// an RMSNorm of the row they have just completed).  One 256-thread block per row: every thread sums its 16-B pieces over the
// slabs in slab order, adds the residual LAST (the order of gemm_splitk_reduce_kernel: C is bit-identical to the two-kernel
// form), stores the f32 row, and the block then reduces sum(v^2) (wave shuffles, the four wave sums added in wave order: a
// fixed order, not the one-wave-per-row order of norm_kernel) and writes xn = bf16(v * rsqrt(mean + eps) * gamma).
// Saves a launch and a re-read of the row per instance (9.8 + 13.8 us -> one kernel at 256 rows).
template <int VPT>   // 16-B pieces per thread: N <= 1024 * VPT
__global__ __launch_bounds__(256) void splitk_reduce_rmsnorm_kernel(GemmParams p, const float* gamma, float eps,
                                                                    unsigned short* xn, int64_t ldx) {
  __shared__ float wsum[4];
  const int m = blockIdx.x, tid = threadIdx.x, nvec = p.N >> 2;
  const int64_t slab = (int64_t)p.M * p.N;
  f32x4 v[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = tid + i * 256;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nvec) v[i] = sum_slabs4(p.ws + (int64_t)m * p.N + c * 4, slab, p.split_k);
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = tid + i * 256;
    if (c < nvec) {
      if (p.epi & ICL_EPI_RESIDUAL) v[i] = v[i] + *(const f32x4*)((const float*)p.R + (int64_t)m * p.ldr + c * 4);
      *(f32x4*)((float*)p.C + (int64_t)m * p.ldc + c * 4) = v[i];
      ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
  }
  ss = wave_reduce_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  const float rstd = rsqrtf((((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]) / (float)p.N + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = tid + i * 256;
    if (c < nvec) {
      const f32x4 g = *(const f32x4*)(gamma + c * 4);
      const f32x4 o = v[i] * rstd * g;
      *(u32x2*)(xn + (int64_t)m * ldx + c * 4) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    }
  }
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

template <int WAVES_M, int WAVES_N, int MI, int NI, int NST>
int launch_tile_n(GemmParams& p, dim3 grid, hipStream_t stream) {
  constexpr int BM = WAVES_M * MI * 16, BN = WAVES_N * NI * 16;
  constexpr int SMEM = (BM + BN) * 128 * NST;
  auto kern = gemm_bf16_kernel<WAVES_M, WAVES_N, MI, NI, NST>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      icl_set_error("icl_gemm_bf16: hipFuncSetAttribute(%d) failed: %s", SMEM, hipGetErrorString(e));
      return ICL_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, stream, p);
  ICL_CHECK_LAUNCH("icl_gemm_bf16");
  return ICL_OK;
}

template <int WAVES_M, int WAVES_N, int MI, int NI>
int launch_tile(GemmParams& p, int batch, hipStream_t stream) {
  constexpr int BM = WAVES_M * MI * 16, BN = WAVES_N * NI * 16;
  constexpr int DEEP = TILE_STAGES_DEEP(BM, BN);
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.split_k > 1 ? p.split_k : batch);
  static int n_cu = 0;
  if (n_cu <= 0) {
    n_cu = icl_device_cu_count();
    if (n_cu <= 0) n_cu = 256;
  }
  // deep ring only when ALL blocks of the launch fit on the chip at once at its LDS footprint (160 KiB per CU): measured on
  // MI355X (tools/gemm_ab.py --shapes small): 384 blocks 35 -> 26 us, 480 blocks 43 -> 37 us; with blocks to spare the two-stage
  // form's occupancy wins (1152 blocks: 61 us vs 86 us).  The choice does not change a bit of the result.
  const int64_t blocks = (int64_t)grid.x * grid.z;
  const int per_cu = (160 * 1024) / ((BM + BN) * 128 * DEEP);
  if (blocks <= (int64_t)n_cu * per_cu) return launch_tile_n<WAVES_M, WAVES_N, MI, NI, DEEP>(p, grid, stream);
  return launch_tile_n<WAVES_M, WAVES_N, MI, NI, 2>(p, grid, stream);
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
  // measured at micro-batch 128: the 256x256 tile wins from K = 768 (BEATs) upwards.  One partial round of 256-tiles still beats
  // two rounds of 128-tiles down to ~60 % of the CUs (one utterance's gate/up, 376 x 22016 x 4096: 172 tiles, 76 us vs 115 us on
  // the 128-tile; at 96 tiles — its QKV projection — the 128-tile wins 67 vs 71 us; tools/gemm_small_m.py, round 4).  Tiles 1 - 3
  // sum K in the same order, so the choice never changes a row's bits.
  if (K >= 768 && eff >= 0.6) return 3;
  return 1;
}

struct NormFuse {
  const float* gamma;
  float eps;
  unsigned short* xn;
  int64_t ldx;
};

static int gemm_impl(const icl_gemm_args* a, void* stream_, const RopeFuse* rope, const NormFuse* norm = nullptr) {
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
  p.group_m = p.xcd_sync = 0;   // the 128x128 / 64x64 tiles keep GROUP_M; the 256x256 launcher picks per shape

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
    // split-K on the 256x256 tile: every K slice must feed the pipeline's two peeled K-tiles (>= 128 deep), and the slab layout
    // is that of the other tiles (N % 4 == 0 for the 16-B slab rows of the staged epilogue)
    ICL_CHECK_ARG(a->split_k == 1 || (a->split_k <= a->K / 128 && a->N % 4 == 0),
                  "icl_gemm_bf16: split_k=%d on the 256x256 tile needs K / split_k >= 128 and N %% 4 == 0 (K=%d N=%d)", a->split_k, a->K, a->N);
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
  if (a->split_k > 1 && norm) {      // reduction + residual + the next RMSNorm in one kernel (validated by the caller)
    if (a->N <= 4096) hipLaunchKernelGGL(splitk_reduce_rmsnorm_kernel<4>, dim3(a->M), dim3(256), 0, stream, p, norm->gamma, norm->eps, norm->xn, norm->ldx);
    else              hipLaunchKernelGGL(splitk_reduce_rmsnorm_kernel<8>, dim3(a->M), dim3(256), 0, stream, p, norm->gamma, norm->eps, norm->xn, norm->ldx);
    ICL_CHECK_LAUNCH("icl_gemm_rmsnorm_bf16(split-K reduce + RMSNorm)");
  } else if (a->split_k > 1) {
    const int64_t total = (int64_t)a->M * nout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    ICL_CHECK_LAUNCH("icl_gemm_bf16(split-K reduce)");
  }
  return ICL_OK;
}

extern "C" int icl_rmsnorm(const void* x, int64_t ldx, const float* gamma, void* y, int64_t ldy, int32_t M, int32_t N, float eps,
                           int32_t in_dtype, int32_t out_dtype, void* stream);

extern "C" int icl_gemm_rmsnorm_bf16(const icl_gemm_args* a, const float* gamma, float eps, void* xn, int64_t ld_xn, void* stream) {
  ICL_CHECK_ARG(a != nullptr && gamma && xn, "icl_gemm_rmsnorm_bf16: NULL pointer");
  ICL_CHECK_ARG(a->batch == 1 && a->out_dtype == ICL_F32 && (a->epilogue & ~ICL_EPI_RESIDUAL) == 0 &&
                    (!(a->epilogue & ICL_EPI_RESIDUAL) || a->res_dtype == ICL_F32),
                "icl_gemm_rmsnorm_bf16: batch 1, f32 output, residual-only epilogue with an f32 residual");
  ICL_CHECK_ARG(a->N % 4 == 0 && a->N <= 8192 && a->ldc % 4 == 0 && a->ldr % 4 == 0 && ld_xn % 4 == 0 && ld_xn >= a->N,
                "icl_gemm_rmsnorm_bf16: N=%d must be a multiple of 4 and <= 8192, leading dimensions multiples of 4", a->N);
  ICL_CHECK_ARG(((uintptr_t)gamma & 15) == 0 && ((uintptr_t)xn & 7) == 0 && ((uintptr_t)a->C & 15) == 0 &&
                    (!a->R || ((uintptr_t)a->R & 15) == 0), "icl_gemm_rmsnorm_bf16: misaligned pointer");
  // the skinny kernels (tiles 4 / 6) split K inside the block and write the finished row themselves whatever split_k says:
  // only a launch that really leaves split-K slabs behind takes the fused reduce + RMSNorm
  if (a->split_k > 1 && a->tile != 4 && a->tile != 6) {
    const NormFuse nf{gamma, eps, (unsigned short*)xn, ld_xn};
    return gemm_impl(a, stream, nullptr, &nf);
  }
  // (Round 4 tried running this norm inside the skinny launch, in the block that draws the last of one ticket per block — correct,
  // and 6-8 us SLOWER per call than the second launch inside a HIP graph: the device-scope release fence in front of the ticket
  // is an L2 write-back; tools/skinny_tail_time.py)
  const int rc = gemm_impl(a, stream, nullptr);            // no slabs to reduce: the GEMM's own epilogue, then the plain norm
  if (rc != ICL_OK) return rc;
#ifndef ICL_SMALL_M_BLOCKNORM
#define ICL_SMALL_M_BLOCKNORM 1   // round 4: 58.7 -> 56.7 ms per utterance at one sequence (tools/ab_bench_b1.sh, profiles/r04_skinny_tail_ab.txt)
#endif
  if (ICL_SMALL_M_BLOCKNORM && a->M <= 64 && a->ldc == a->N) {
    // few rows (a decode step of <= 64 sequences): the row-per-block kernel of the split-K path (four waves per row) over the
    // finished rows as a single "slab" — the one-wave-per-row norm_kernel is a 9-us latency chain at one row
    GemmParams q{};
    q.ws = (float*)a->C; q.C = a->C; q.R = nullptr; q.ldc = a->ldc; q.ldr = 0; q.M = a->M; q.N = a->N; q.split_k = 1; q.epi = 0;
    if (a->N <= 4096) hipLaunchKernelGGL(splitk_reduce_rmsnorm_kernel<4>, dim3(a->M), dim3(256), 0, (hipStream_t)stream, q, gamma, eps, (unsigned short*)xn, ld_xn);
    else              hipLaunchKernelGGL(splitk_reduce_rmsnorm_kernel<8>, dim3(a->M), dim3(256), 0, (hipStream_t)stream, q, gamma, eps, (unsigned short*)xn, ld_xn);
    ICL_CHECK_LAUNCH("icl_gemm_rmsnorm_bf16(row-per-block RMSNorm)");
    return ICL_OK;
  }
  return icl_rmsnorm(a->C, a->ldc, gamma, xn, ld_xn, a->M, a->N, eps, ICL_F32, ICL_BF16, stream);
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
