// gemm256.hip — the 256x256x64 bf16 MFMA GEMM tile of libicl_hip (the dominant kernel of the hot path).
//
// One kernel INSTANTIATION per epilogue kind (GELU / f32 residual / SwiGLU / output type are template parameters; only the
// bias stays a run-time flag): with every flag a run-time branch the kernel was 85 k lines of ISA (7.4 k conditional
// branches, every accumulator fragment handled under each of them) and its epilogue ran from a cold instruction cache on
// every tile.  The arithmetic of a path is unchanged by the specialisation — outputs are bit-identical.
#include "gemm_common.h"
#include <cstdlib>

using namespace iclg;

#ifndef ICL_EPI_NT
// Non-temporal loads / stores in the staged epilogue (round 4): an output tile is written once and read by a LATER launch, after
// gigabytes of other traffic, and the f32 residual is read once — kept out of the caches they stop evicting the A / W panels the
// main loops of the other CUs re-read (Whisper o +3.5 %, qkv +2.2 %, Llama gate/up +1.4 % in tools/gemm_ab.py; in situ -5.5 ms
// of a 1791 ms step: the consumer of an output loses the few Infinity-Cache hits it had; profiles/r04_gemm_nt_ab.txt).
// bit 0 = the f32 residual stream (load + store), bit 1 = the other staged tiles (not split-K slabs: their reduction re-reads
// them immediately), bit 2 = the fused RoPE epilogue's stores (measured: no difference, off).
#define ICL_EPI_NT 3
#endif

namespace {

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

__device__ __forceinline__ void st16(unsigned short* dst, u32x4 v) {      // a 16-B piece of a staged output row
  if (ICL_EPI_NT & 4) __builtin_nontemporal_store(v, (u32x4*)dst);
  else *(u32x4*)dst = v;
}

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
        if (rf.kv_rows_to_c) st16(C + (int64_t)m * p.ldc + n0 + cc * 8, v);
        if (rf.vc) st16(rf.vc + (crow[u] + (int64_t)(head0 + (cc >> 4)) * rf.max_len) * 128 + (cc & 15) * 8, v);
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
        st16(dst, olo);
        st16(dst + 64, ohi);
      }
      if (to_cache) {
        unsigned short* cd = rf.kc + (((int64_t)sq[u] * rf.H + head0 + hh) * rf.max_len + ps[u]) * 128 + j * 8;
        st16(cd, olo);
        st16(cd + 64, ohi);
      }
    }
  }
}

// EPI: compile-time epilogue kind = ICL_EPI_GELU | ICL_EPI_RESIDUAL | ICL_EPI_SWIGLU bits + T256_OUT_F32; the bias bit stays in p.epi
constexpr int T256_OUT_F32 = 16;
template <bool ROPE, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmParams p_in, RopeFuse rf) {
  // the run-time flags of the parameter block are replaced by the instantiation's constants: every `p.epi & FLAG` /
  // `p.out_dtype == ...` below (and inside the shared epilogue helpers) folds at compile time
  GemmParams p = p_in;
  p.epi = (p_in.epi & ICL_EPI_BIAS) | (EPI & (ICL_EPI_GELU | ICL_EPI_RESIDUAL | ICL_EPI_SWIGLU));
  p.out_dtype = (EPI & T256_OUT_F32) ? ICL_F32 : ICL_BF16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int tm, tn;
  block_to_tile(p, blockIdx.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  // grid.z: batch index, or — split_k > 1 (decode at 129..256 rows: one M-tile, too few N-tiles to fill the chip) — the K slice
  // [kt0, kt0 + nk) of this block, whose raw f32 partial tile goes to slab z of the workspace (the layout of the other tiles'
  // split-K: gemm_splitk_reduce_kernel / splitk_reduce_rmsnorm_kernel sum the slabs in slab order and run the epilogue)
  int z = blockIdx.z;
  int nk = p.K >> 6, kt0 = 0;
  const bool slab_out = p.split_k > 1;      // split-K slabs are re-read by the reduction right away: ordinary (cacheable) stores
  if (p.split_k > 1) {
    kt0 = (int)(((int64_t)z * nk) / p.split_k);
    nk = (int)(((int64_t)(z + 1) * nk) / p.split_k) - kt0;
    p.C = p.ws + (int64_t)z * p.M * p.N;
    p.ldc = p.N;
    p.sC = 0;
    p.epi = 0;                 // bias / residual / activation belong to the reduction
    z = 0;
  }
  const __bf16* A = p.A + (int64_t)z * p.sA + (int64_t)kt0 * 64;
  const __bf16* Wk = p.W + (int64_t)kt0 * 64;

  // ---- staging sources: region h, round r -> rows 8*(r*8 + wave) + (lane>>3) of the region -------------------
  const __bf16* gsrc[2][2][2];  // [A|B][region][round]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = (r * 8 + wave) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      gsrc[0][h][r] = A + (int64_t)min(m0 + h * 128 + row, p.M - 1) * p.lda + chunk * 8;
      gsrc[1][h][r] = Wk + (int64_t)min(n0 + h * 128 + row, p.N - 1) * p.ldw + chunk * 8;
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
    // f32 residual stream (always the two-round form): the residual rows of a half tile are whole-row 16-B loads, 16 per thread.
    // They are issued a stage AHEAD of their use — half 0's before the first barrier and the C -> LDS writes, half 1's before
    // half 0's row phase — so their latency (every CU reaches its epilogue at about the same time: a burst of 128 MB chip-wide)
    // is covered by LDS work instead of being waited for twice; the accumulator halves they replace are dead by then.
    f32x4 rrA[16], rrB[16];
    auto load_res_rows = [&](int qa_, f32x4 (&rr)[16]) {
      const char* rbase = (const char*)p.R + ((int64_t)z * p.sR + (int64_t)(m0 + qa_ * 128) * p.ldr + n0) * 4;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int c = tid + it * 512, row = c >> 6, cc = c & 63;
        rr[it] = ICL_EPI_NT & 1 ? __builtin_nontemporal_load((const f32x4*)(rbase + (int64_t)row * p.ldr * 4 + cc * 16))
                                : *(const f32x4*)(rbase + (int64_t)row * p.ldr * 4 + cc * 16);
      }
    };
    if (has_res) load_res_rows(0, rrA);
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
      if (has_res) {
        if (qa == 0) load_res_rows(1, rrB);                      // the next half's residual rows, under this half's row phase
        f32x4 (&rr)[16] = qa == 0 ? rrA : rrB;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int c = tid + it * 512, row = c >> 6, cc = c & 63;
          const f32x4 v = *(const f32x4*)(smem + row * pitch + cc * 16) + rr[it];
          if (ICL_EPI_NT & 1) __builtin_nontemporal_store(v, (f32x4*)(cbase + (int64_t)row * p.ldc * 4 + cc * 16));
          else *(f32x4*)(cbase + (int64_t)row * p.ldc * 4 + cc * 16) = v;
        }
      } else {
        // all of a thread's row pieces are read from LDS first (independent reads in flight together), then stored: the loop
        // form waited for each 16-B LDS read before issuing its store (16 dependent round trips per thread)
        const int n_iter = n_chunks >> 9;                      // 8 | 16 (a compile-time constant per instantiation)
        u32x4 v[16];
#pragma unroll
        for (int it = 0; it < 16; ++it)
          if (it < n_iter) {
            const int c = tid + it * 512, row = c / chunks_per_row, cc = c - row * chunks_per_row;
            v[it] = *(const u32x4*)(smem + row * pitch + cc * 16);
          }
#pragma unroll
        for (int it = 0; it < 16; ++it)
          if (it < n_iter) {
            const int c = tid + it * 512, row = c / chunks_per_row, cc = c - row * chunks_per_row;
            if ((ICL_EPI_NT & 2) && !slab_out) __builtin_nontemporal_store(v[it], (u32x4*)(cbase + (int64_t)row * p.ldc * es + cc * 16));
            else *(u32x4*)(cbase + (int64_t)row * p.ldc * es + cc * 16) = v[it];
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

template <bool ROPE, int EPI>
int launch256(const GemmParams& p, const dim3& grid, hipStream_t stream, const RopeFuse& rf) {
  auto kern = gemm256_bf16_kernel<ROPE, EPI>;
  static bool attr_set = false;      // one flag per instantiation
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T256_SMEM);
    if (e != hipSuccess) {
      icl_set_error("icl_gemm_bf16: hipFuncSetAttribute(%d) failed: %s", T256_SMEM, hipGetErrorString(e));
      return ICL_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), T256_SMEM, stream, p, rf);
  ICL_CHECK_LAUNCH("icl_gemm_bf16(256)");
  return ICL_OK;
}

}  // namespace

int iclg::launch_tile256(GemmParams& p, int batch, hipStream_t stream, const RopeFuse* rope) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  {
    // Super-tile height of the block -> tile map (gemm_common.h, block_to_tile): the XCD's window of ~32 concurrent tiles is
    // gm M-tiles x 32/gm N-tiles.  Both shapes with gm + 32/gm = 12 fetch the same number of panels per window, but the A panels
    // (256 rows x K, streamed from HBM, re-read by every window of their super-tile) are what has to survive in the Infinity
    // Cache between windows while all eight XCDs stream: measured on the bench shapes (tools/gemm_ab.py, profiles/r03_gemm_ab.txt)
    // the best gm keeps an XCD's A side near 8 MB — 1-2 at K = 11008, 3-4 at K = 4096-5120, 4-6 at K <= 1280; gm = 8 (rounds
    // 1-2) cost 1.5-3.5 % at K >= 4096 and gm = 16 / 32 cost 8 / 16 %.
    static const int env_gm = [] { const char* e = getenv("ICL_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();   // tuning knobs
    // XCD-synchronised super-tile map (block_to_tile): on since round 4 — with the outputs no longer cached (non-temporal stores) the
    // eight XCDs sweeping the same N range together is worth 6-9 ms of a 1790 ms step in five interleaved same-box rounds
    // (profiles/r04_gemm_xcd_sync_ab.txt; within +-0.7 % in round 3's isolated-GEMM test).  ICL_GEMM_XCD_SYNC=0 restores the contiguous split.
    static const int env_xs = [] { const char* e = getenv("ICL_GEMM_XCD_SYNC"); return e ? atoi(e) : 1; }();
    const int64_t panel = (int64_t)256 * p.K * 2;
    p.group_m = env_gm > 0 ? env_gm : (int)std::min<int64_t>(6, std::max<int64_t>(1, 8400000 / panel));
    p.xcd_sync = env_xs;
  }
  const dim3 grid(p.tiles_m * p.tiles_n, 1, p.split_k > 1 ? p.split_k : batch);
  if (rope) return launch256<true, 0>(p, grid, stream, *rope);      // bf16 output, bias-only epilogue (checked by the caller)
  const RopeFuse none{};
  constexpr int G = ICL_EPI_GELU, R = ICL_EPI_RESIDUAL, S = ICL_EPI_SWIGLU, F = T256_OUT_F32;
  if (p.split_k > 1) return launch256<false, F>(p, grid, stream, none);      // raw f32 partial tiles into the workspace slabs
  switch ((p.epi & (G | R | S)) | (p.out_dtype == ICL_F32 ? F : 0)) {
    case 0:         return launch256<false, 0>(p, grid, stream, none);
    case G:         return launch256<false, G>(p, grid, stream, none);
    case R:         return launch256<false, R>(p, grid, stream, none);
    case G | R:     return launch256<false, G | R>(p, grid, stream, none);
    case S:         return launch256<false, S>(p, grid, stream, none);
    case F:         return launch256<false, F>(p, grid, stream, none);
    case F | G:     return launch256<false, F | G>(p, grid, stream, none);
    case F | R:     return launch256<false, F | R>(p, grid, stream, none);
    case F | G | R: return launch256<false, F | G | R>(p, grid, stream, none);
    case F | S:     return launch256<false, F | S>(p, grid, stream, none);
  }
  icl_set_error("icl_gemm_bf16(256): unsupported epilogue combination 0x%x", p.epi);   // SwiGLU with GELU / residual: rejected upstream
  return ICL_EINVAL;
}
