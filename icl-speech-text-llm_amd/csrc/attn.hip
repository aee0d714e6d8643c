// attn.hip — fused (flash-style) attention forward for gfx950, head_dim 64 / 128, bf16 in/out.
//
// Structure (cdna_hip_programming.md Appendix B "Fused attention prefill", T10/T12/T14):
//   * block = 4 waves, each wave owns 32 query rows of one (sequence, head); KV tiles of 64 keys.
//   * swapped QK^T: S^T = K · Q^T on v_mfma_f32_32x32x16_bf16, so the query sits on the LANE and
//     its scores sit in that lane's accumulator registers -> row max / row sum are in-register
//     (one cross-half exchange), no LDS round trip for P.
//   * O^T = V^T · P^T: the S^T accumulator registers 8s..8s+7 (converted to bf16) ARE the B operand
//     of k-step s (guide §3 "An accumulator tile as the next MFMA's operand"); the matching,
//     k-permuted V^T A-operand comes from two ds_read_b64_tr_b16 transposed reads of the
//     row-major V tile.  The per-query rescale is then a per-lane scalar.
//   * K/V tiles arrive by LDS-DMA (global_load_lds, 16 B per lane, 1 KiB per wave-instruction, no VGPR round trip and no
//     ds_write): tile t+1 is issued at the top of iteration t into the OTHER buffer, one vmcnt(0) + barrier per tile.
//     The DMA destination is lane-linear, so rows are unpadded and bank conflicts are removed by swizzling on the SOURCE
//     side: LDS slot s of row r holds global 16-B chunk s ^ f(r); K: f = (r>>1)&7 (D=64) / r&15 (D=128) makes the
//     ds_read_b128 fragment reads conflict-free, V: f = ((r>>1)&1)<<2 (D=64) / (r&3)<<2 (D=128) does it for the
//     transposed ds_read_b64_tr_b16 reads.
// Variable-length packing (cu_seqlens), causal masking, a key-padding length per sequence and the
// BEATs gated relative-position bias are handled in the score stage.
#include "common.h"
#include <type_traits>

#ifndef ICL_ATTN_V
#define ICL_ATTN_V 1       // A/B switches of attn_fwd_kernel (tools/attn_ab.sh): bit 0 = no re-staging in the last AHEAD tiles, bit 1 = heavy-first causal q-blocks
#endif

namespace {

struct AttnParams {
  const unsigned short* Q;
  const unsigned short* K;
  const unsigned short* V;
  unsigned short* O;
  const int* cu;
  const int* kv_lens;
  const float* rel_bias;
  const float* rel_gate;
  int64_t ldq, ldk, ldv, ldo;
  int64_t kv_seq_stride, kv_head_stride;   // != 0: K/V live in a [seq][head][pos][D]-style cache (row stride ldk / ldv)
  int n_heads, rel_span, n_qblocks;
  float scale_log2e;
};

constexpr float NEG_BIG = -1.0e30f;

// Row maxima use plain fmaxf (hipcc fuses the chain into v_max3_f32 and canonicalises only the first pair).  They must NOT be
// inline asm: reading an MFMA result from a vector instruction needs software wait states on gfx950, hipcc's hazard
// recogniser does not look inside asm statements, and its scheduler is free to move one right behind the MFMA that
// produces its operand — the row maximum then comes out stale now and then (harmless to the value of a softmax, fatal to
// bit-reproducibility; caught by the packed-rows == cache-rows test).
__device__ __forceinline__ float max32(const f32x16& a, const f32x16& b) {
  float m = __builtin_fmaxf(a[0], a[1]);
#pragma unroll
  for (int r = 2; r < 16; ++r) m = __builtin_fmaxf(m, a[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) m = __builtin_fmaxf(m, b[r]);
  return m;
}
constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// max of a value with its partner in the other half of the wave (lane ^ 32), in every lane: one v_permlane32_swap (lanes
// 32..63 of the first operand trade places with lanes 0..31 of the second) instead of a ds_bpermute round trip through LDS
// in the middle of the softmax's dependent chain
__device__ __forceinline__ float max_xhalf(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned lo = r[0], hi = r[1];   // (a bit_cast applied to r[1] directly reads element 0 with this hipcc)
  return __builtin_fmaxf(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
}

template <int D, bool CAUSAL, bool BIAS>
__global__ __launch_bounds__(256, (D == 64 && BIAS) ? 3 : 2) void attn_fwd_kernel(AttnParams p) {
  // 32-query blocks per wave.  D = 64 without bias runs attn_fwd_il64_kernel below (two blocks per wave sharing every K / V
  // fragment, stages interleaved by hand); this kernel serves D = 128 and the gated-bias variant, one block per wave
  // (measured for the bias variant: its longer per-score sequence wants the third wave per SIMD more than the sharing,
  // 407 vs 385 TF/s on the BEATs shape).  The QB > 1 paths are kept: they are the reference the il64 kernel was checked
  // against bit for bit.
  constexpr int QB = 1;
  // D = 128 (the decoders' prefill / teacher-forced attention): P enters the PV product as a TWO-term bf16 split,
  // P = hi + lo with hi = bf16(P), lo = bf16(P - hi) (16 mantissa bits instead of 8), at the price of a second PV MFMA.
  // The oracle's softmax weights are f32: with one-term bf16 P this single rounding point alone put the decoder logits
  // 3.7e-3 (relative L2) from the oracle — measured by rounding P inside the ORACLE — against the north star's 1e-3; with the
  // split they sit at the level of the other, mirrored, rounding points.  The D = 64 encoder kernels keep one term: they
  // meet ~1e-3 already and are vector-issue-bound, where the extra conversions would cost 15-20 %.
  constexpr bool P2 = (D == 128);
  constexpr int BQ = 128 * QB;      // queries per workgroup
  constexpr int ROWB = D * 2;       // bytes per K / V row in LDS (unpadded: LDS-DMA writes lane-linear)
  constexpr int KS = D / 16;        // QK^T k-steps
  constexpr int DB = D / 32;        // output d-blocks
  constexpr int CPR = D / 8;        // 16-B chunks per row
  constexpr int NCH = 64 * CPR / 256;  // LDS-DMA wave-instructions per wave per tensor (1 KiB each)
  constexpr int RPI = 64 / CPR;        // rows per DMA instruction
  constexpr int BUF = 2 * 64 * ROWB;   // one K tile + one V tile
  constexpr int NBUF = D == 64 ? 3 : 2;   // ring depth: tiles are staged NBUF-1 iterations ahead (D=128: 2 x 32 KiB keeps 2 blocks/CU)
  constexpr int AHEAD = NBUF - 1;
  __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF];   // ONE barrier per KV tile
  // gated relative-position bias: per tile and q-block the 95 table entries a wave can touch (rel = key - query over
  // 64 keys x 32 queries) are staged once into a wave-private LDS window; a score then costs one ds_read_b32 at
  // base + immediate instead of clamp + 64-bit address + global gather
  __shared__ float bias_win[BIAS ? 4 * QB * 128 : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;
  // XCD-aware decode of a 1-D grid: consecutive block ids are dispatched round-robin over the 8 XCDs, each with its own L2.
  // Block b -> work item (b % 8) * ceil(n/8)-chunk + b / 8 (bijective), work items ordered q-block fastest: all q-blocks of
  // one (sequence, head) run on ONE XCD back to back, so its K/V (re-read by every q-block) is fetched into one L2 once.
  const int n_blocks = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = n_blocks >> 3, r8 = n_blocks & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  // causal: the q-blocks of a (sequence, head) are dealt last-first — block j walks j + 1 K/V tile pairs, so the long blocks of
  // the launch start first and its tail is made of short ones
  const int qblk = (CAUSAL && (ICL_ATTN_V & 2)) ? p.n_qblocks - 1 - item % p.n_qblocks : item % p.n_qblocks;
  const int head = (item / p.n_qblocks) % p.n_heads, seq = item / (p.n_qblocks * p.n_heads);
  const int row0 = p.cu[seq];
  const int len = p.cu[seq + 1] - row0;
  const int qb = qblk * BQ;
  if (qb >= len) return;
  int kvlen = len;
  if (p.kv_lens) kvlen = min(max(p.kv_lens[seq], 1), len);
  const int kv_end = CAUSAL ? min(kvlen, qb + BQ) : kvlen;
  const int n_tiles = (kv_end + 63) >> 6;

  int qw[QB], qpos[QB];           // first query of each of this wave's q-blocks / this lane's query in it
  bf16x8 qf[QB][KS];              // Q fragments (B operand of S^T = K Q^T): lane (q, hh) holds Q[q][16ks + 8hh .. +7]
  float gate[QB];
  const float* bias_row = nullptr;
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    qw[qi] = qb + (wave * QB + qi) * 32;
    qpos[qi] = qw[qi] + ql;
    const int qrow = min(qpos[qi], len - 1);
    const unsigned short* qp = p.Q + (int64_t)(row0 + qrow) * p.ldq + head * D + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qi][ks] = *(const bf16x8*)(qp + ks * 16);
    gate[qi] = 0.f;
    if (BIAS) gate[qi] = p.rel_gate[(int64_t)(row0 + qrow) * p.n_heads + head] * LOG2E;
  }
  if (BIAS) bias_row = p.rel_bias + (int64_t)head * (2 * p.rel_span - 1) + (p.rel_span - 1);

  // ---- staging: LDS-DMA with source-side swizzle --------------------------------------------------------
  // DMA instruction j = wave * NCH + i of a tensor covers rows RPI*j .. RPI*j + RPI-1; lane l lands in row RPI*j + l / CPR,
  // slot l % CPR, and fetches global chunk slot ^ f(row).  Row pointers advance by one tile per iteration; only a tile that
  // crosses the end of the sequence takes the clamped form (rows past the end are masked, the read must stay in bounds).
  auto f_k = [](int row) { return D == 64 ? (row >> 1) & 7 : row & 15; };
  auto f_v = [](int row) { return D == 64 ? ((row >> 1) & 1) << 2 : (row & 3) << 2; };
  // row 0 of this (sequence, head): packed rows [cu[seq] + j][head*D ..] of the fused QKV buffer, or — when the strides are
  // given — rows [seq][head][j][..] of a KV cache (the prefill reads back what the QKV GEMM's epilogue appended)
  const int64_t kv_off = p.kv_seq_stride ? (int64_t)seq * p.kv_seq_stride + (int64_t)head * p.kv_head_stride : -1;
  const unsigned short* kbase = kv_off >= 0 ? p.K + kv_off : p.K + (int64_t)row0 * p.ldk + head * D;
  const unsigned short* vbase = kv_off >= 0 ? p.V + kv_off : p.V + (int64_t)row0 * p.ldv + head * D;
  const unsigned short* kptr[NCH];
  const unsigned short* vptr[NCH];
  int srow[NCH], kch[NCH], vch[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int j = wave * NCH + i;
    srow[i] = RPI * j + lane / CPR;
    kch[i] = (lane % CPR) ^ f_k(srow[i]);
    vch[i] = (lane % CPR) ^ f_v(srow[i]);
    kptr[i] = kbase + (int64_t)srow[i] * p.ldk + kch[i] * 8;
    vptr[i] = vbase + (int64_t)srow[i] * p.ldv + vch[i] * 8;
  }
  const int64_t kstep = 64 * p.ldk, vstep = 64 * p.ldv;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto stage_tile = [&](int t, int buf) {
    const int k0 = t * 64;
    char* k_w = lds + buf * BUF + wave * NCH * 1024;
    char* v_w = k_w + 64 * ROWB;
    if (k0 + 64 <= len) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(kptr[i] + (int64_t)t * kstep), (lptr_t)(k_w + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vptr[i] + (int64_t)t * vstep), (lptr_t)(v_w + i * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int64_t grow = min(k0 + srow[i], len - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(kbase + grow * p.ldk + kch[i] * 8), (lptr_t)(k_w + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vbase + grow * p.ldv + vch[i] * 8), (lptr_t)(v_w + i * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 o_acc[QB][DB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    m_run[qi] = NEG_BIG;
    l_run[qi] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[qi][d][r] = 0.f;
  }

  // K fragment reads: row kb*32 + ql, logical chunk 2*ks + hh -> physical chunk ^ f_k(row) (f_k(row + 32) = f_k(row))
  int k_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_off[ks] = ql * ROWB + (((2 * ks + hh) ^ f_k(ql)) << 4);
  // transposed V reads: lane i of a 16-lane group reads row (i>>2) [+4*hh, +8 for the second read], logical chunk
  // 4*d + 2*((lane>>4)&1) + ((lane&3)>>1), half (lane&1); the rows' f_v depends only on (i>>2)
  int tr_off[DB];
  {
    const int q = (lane & 15) >> 2;
    const int c2 = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
    for (int d = 0; d < DB; ++d)
      tr_off[d] = (4 * hh + q) * ROWB + (((4 * d + c2) ^ f_v(q)) << 4) + (lane & 1) * 8;
  }

  // counted waits: the AHEAD-1 youngest tiles (2*NCH LDS-DMA instructions each) stay in flight across the barrier
  auto wait_oldest_tile = [&]() {
    if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NCH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) stage_tile(min(a, n_tiles - 1), a);
  wait_oldest_tile();
  __syncthreads();
  int cur = 0;            // ring slot of tile t

  // One KV tile.  MAYMASK = false is the interior form: every key of the tile is visible to every query of the wave, so
  // the body carries no mask, no per-q-block branch and none of the key-index arithmetic hipcc otherwise hoists above the
  // "need_mask" test and executes on every tile (35 vector instructions per iteration of the D = 64 kernel).
  auto tile = [&](const int t, auto maymask_c, auto last_c) {
    constexpr bool MAYMASK = decltype(maymask_c)::value;
    constexpr bool LAST = decltype(last_c)::value;      // one of the last AHEAD tiles: nothing left to stage
    // unconditional within an instantiation: a run-time condition around the issue makes hipcc's waitcnt pass merge the two
    // paths pessimistically.  The last AHEAD iterations are their own instantiation WITHOUT the staging (they used to stage
    // the last tile again into the idle buffer: 2 * NCH LDS-DMA instructions, ~45 ns of issue each, per wave — on the decoders'
    // prefill, 2-6 tiles per block, that was 17-50 % more staging than the block needs)
    int nxt = cur + AHEAD;
    if (nxt >= NBUF) nxt -= NBUF;
    if constexpr (!LAST) stage_tile(min(t + AHEAD, n_tiles - 1), nxt);   // that slot held tile t-1: last read in iteration t-1 (barrier passed)
    const int k0 = t * 64;
    const char* k_lds = lds + cur * BUF;
    const char* v_lds = k_lds + 64 * ROWB;
    // wave-uniform: which of this wave's q-blocks see a key of this tile?
    bool active[QB];
    bool any_active = false;
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
      active[qi] = !MAYMASK || !CAUSAL || (k0 <= qw[qi] + 31);
      any_active |= active[qi];
    }
    if (any_active) {
      // ---- S^T = K Q^T: every K fragment is read once and feeds all q-blocks --------------------------------
      f32x16 s_acc[QB][2];
#pragma unroll
      for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s_acc[qi][kb][r] = 0.f;
      // K fragments by inline-asm ds_read_b128 with hand-counted waits, LA k-steps ahead of the MFMAs that consume them
      // (same reason as the V reads below: a compiler-visible LDS read makes hipcc wait for every LDS-DMA in flight).
      // k-step outer, key-half inner: each accumulator still sums its k-steps in ascending order.
      {
        constexpr int LA = QB == 2 ? 1 : 2;
        bf16x8 kf[LA + 1][2];
        auto read_k = [&](int ks, int slot) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const unsigned a = (unsigned)(uintptr_t)(k_lds + kb * 32 * ROWB + k_off[ks]);
            asm volatile("ds_read_b128 %0, %1" : "=v"(kf[slot][kb]) : "v"(a));
          }
        };
#pragma unroll
        for (int i = 0; i < LA; ++i) read_k(i, i);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int slot = ks % (LA + 1);
          if (ks + LA < KS) read_k(ks + LA, (ks + LA) % (LA + 1));
          const int newer = 2 * (KS - 1 - ks < LA ? KS - 1 - ks : LA);      // reads issued after this k-step's pair
          if (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[slot][0]), "+v"(kf[slot][1]));
          else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kf[slot][0]), "+v"(kf[slot][1]));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[slot][0]), "+v"(kf[slot][1]));
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
              if (active[qi])
                s_acc[qi][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[slot][kb], qf[qi][ks], s_acc[qi][kb], 0, 0, 0);
        }
      }
      // ---- scores -> base-2 logits, bias, mask, online softmax; per q-block ---------------------------------------
      bf16x8 pf[QB][2][2];
      bf16x8 pl[P2 ? QB : 1][2][2];      // low halves of the two-term bf16 split of P (D = 128 only)
#pragma unroll
      for (int qi = 0; qi < QB; ++qi) {
        if (!active[qi]) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                pf[qi][kb][h2][e] = (__bf16)0.f;
                if (P2) pl[qi][kb][h2][e] = (__bf16)0.f;
              }
          continue;
        }
        // need_mask is wave-uniform: interior tiles take the branch-free fast path (raw v_exp_f32, one FMA per score)
        const bool need_mask =
            MAYMASK && __builtin_amdgcn_readfirstlane((int)((k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw[qi]))));
        float psum = 0.f, alpha;
        if (!BIAS && !need_mask) {
          float tmax = max32(s_acc[qi][0], s_acc[qi][1]);
          tmax = max_xhalf(tmax) * p.scale_log2e;   // scale > 0: max commutes with the scaling
          const float m_new = __builtin_fmaxf(m_run[qi], tmax);
          alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
          m_run[qi] = m_new;
          // two scores per vector instruction where the ISA has a packed f32 form (v_pk_fma_f32, v_pk_add_f32: the
          // accumulator registers are consecutive, so pairs are free); the row sum runs as two partial sums
          const f32x2 sc2 = {p.scale_log2e, p.scale_log2e}, nm2 = {-m_new, -m_new};
          f32x2 ps2 = {0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const f32x2 x = f32x2{s_acc[qi][kb][r], s_acc[qi][kb][r + 1]} * sc2 + nm2;
              const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
              ps2 += e;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                pf[qi][kb][r >> 3][(r & 7) + h] = (__bf16)e[h];
                if (P2) pl[qi][kb][r >> 3][(r & 7) + h] = (__bf16)(e[h] - (float)pf[qi][kb][r >> 3][(r & 7) + h]);
              }
            }
          psum = ps2[0] + ps2[1];
        } else {
          float tmax;
          const float* win = nullptr;
          if (BIAS) {
            float* w = bias_win + (wave * QB + qi) * 128;
            const int base_rel = k0 - qw[qi] - 31;          // window index i <-> rel = base_rel + i, i in [0, 95)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int rel = max(-(p.rel_span - 1), min(p.rel_span - 1, base_rel + lane + 64 * i));
              w[lane + 64 * i] = bias_row[rel];
            }
            __builtin_amdgcn_wave_barrier();                 // same wave, in-order LDS queue: the reads below see the writes
            win = w + (4 * hh - ql + 31);
          }
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              float v = s_acc[qi][kb][r] * p.scale_log2e;
              if (BIAS) v = fmaf(gate[qi], win[kb * 32 + (r & 3) + 8 * (r >> 2)], v);
              if (need_mask) {
                const bool ok = (key < kvlen) && (!CAUSAL || key <= qpos[qi]);
                v = ok ? v : NEG_BIG;
              }
              s_acc[qi][kb][r] = v;
            }
          }
          tmax = max_xhalf(max32(s_acc[qi][0], s_acc[qi][1]));
          const float m_new = __builtin_fmaxf(m_run[qi], tmax);
          alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
          m_run[qi] = m_new;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float e = __builtin_amdgcn_exp2f(s_acc[qi][kb][r] - m_new);
              if (need_mask) e = (s_acc[qi][kb][r] <= NEG_BIG * 0.5f) ? 0.f : e;
              psum += e;
              pf[qi][kb][r >> 3][r & 7] = (__bf16)e;
              if (P2) pl[qi][kb][r >> 3][r & 7] = (__bf16)(e - (float)pf[qi][kb][r >> 3][r & 7]);
            }
          }
        }
        l_run[qi] = l_run[qi] * alpha + psum;
        // the running maximum settles after the first tiles: skip the rescale when no query of the wave moved (x * 1.0f is exact)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
          for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[qi][d][r] *= alpha;
        }
      }
      // ---- O^T += V^T P^T: every transposed V fragment is read once and feeds all q-blocks ----------------------------
      // The transposed reads are inline asm with hand-counted lgkmcnt waits: through the builtin, hipcc's waitcnt pass cannot
      // tell the read from the LDS-DMA writes in flight (the tiles staged AHEAD) and puts s_waitcnt vmcnt(0) in front of the
      // first one, which turns the ring back into "wait for everything you just issued".  Tile t itself is known to have
      // landed (counted wait + barrier at the end of the previous iteration).  One unit (4 reads) is kept in flight ahead of
      // the MFMAs; the wait statement carries the fragments as operands so the MFMAs cannot be scheduled above it.
      {
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        // unit u = (d-block, key half): two fragments (k-steps s = 0, 1) = 4 reads, 8 VGPRs; one unit in flight ahead
        s16x4 vr[2][2][2];
        auto read_unit = [&](int u, int slot) {
          const int d = u >> 1, kb = u & 1;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const unsigned a = (unsigned)(uintptr_t)(v_lds + (kb * 32 + 16 * s) * ROWB + tr_off[d]);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vr[slot][s][0]) : "v"(a));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[slot][s][1]) : "v"(a), "n"(8 * ROWB));
          }
        };
        read_unit(0, 0);
#pragma unroll
        for (int u = 0; u < 2 * DB; ++u) {
          const int slot = u & 1, d = u >> 1, kb = u & 1;
          if (u + 1 < 2 * DB) {
            read_unit(u + 1, slot ^ 1);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vr[slot][0][0]), "+v"(vr[slot][0][1]), "+v"(vr[slot][1][0]), "+v"(vr[slot][1][1]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[slot][0][0]), "+v"(vr[slot][0][1]), "+v"(vr[slot][1][0]), "+v"(vr[slot][1][1]));
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const s16x8 both = __builtin_shufflevector(vr[slot][s][0], vr[slot][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
              if (active[qi]) {
                o_acc[qi][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qi][kb][s], o_acc[qi][d], 0, 0, 0);
                if (P2) o_acc[qi][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pl[qi][kb][s], o_acc[qi][d], 0, 0, 0);
              }
          }
        }
      }
    }
    if constexpr (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing was staged behind tile t+1 (if there is one)
    else wait_oldest_tile();   // this wave's share of tile t+1 has landed
    // bare s_barrier, not __syncthreads(): the fences of the latter make hipcc drain every LDS-DMA in flight (vmcnt(0)).
    // What the barrier orders is covered by hand: this wave's DMA share of the next tile (counted wait above) and its LDS
    // reads of tile t (every asm read was waited for before the MFMA that consumed it)
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == NBUF ? 0 : cur + 1;
  };
  // interior tiles first (whole tile inside kv_len and, when causal, at or below the wave's first query), then the rest;
  // the split is per wave, every wave still passes one barrier per tile
  // (with the gated bias too: its interior body is the general path minus the mask and the key-index arithmetic)
  int t = 0;
  if constexpr (ICL_ATTN_V & 1) {
    // the last AHEAD tiles run the general (may-mask) body without staging; need_mask is evaluated per tile there anyway
    const int n_staged = max(n_tiles - AHEAD, 0);
    const int n_plain = min(n_staged, CAUSAL ? min(kvlen, qw[0] + 1) >> 6 : kvlen >> 6);
    for (; t < n_plain; ++t) tile(t, std::false_type{}, std::false_type{});
    for (; t < n_staged; ++t) tile(t, std::true_type{}, std::false_type{});
    for (; t < n_tiles; ++t) tile(t, std::true_type{}, std::true_type{});
  } else {
    const int n_plain = min(n_tiles, CAUSAL ? min(kvlen, qw[0] + 1) >> 6 : kvlen >> 6);
    for (; t < n_plain; ++t) tile(t, std::false_type{}, std::false_type{});
    for (; t < n_tiles; ++t) tile(t, std::true_type{}, std::false_type{});
  }
  // ring slots are about to be reused by the epilogue: every wave's LDS-DMA must have landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: O[q][d] = O^T[d][q] / l -------------------------------------------------------------
  // A lane owns one query ROW, so direct stores are 8-B pieces at a row stride: every store instruction touches 64 cache
  // lines.  The K/V ring is dead after the loop's last barrier: each wave transposes its rows through a private LDS region
  // and stores whole head-rows (D * 2 bytes = one or two full lines), 16 B per lane.
  float inv[QB];
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
    inv[qi] = l_tot > 0.f ? 1.f / l_tot : 0.f;
  }
  const bool rows16 = (((uintptr_t)p.O | (uintptr_t)(p.ldo * 2)) & 15) == 0;
  if (rows16) {
    constexpr int PITCH = D * 2 + 16;
    constexpr int LPR = D * 2 / 16, RPI = 64 / LPR;     // lanes per row, rows per store instruction
    char* stg = lds + wave * (QB * 32 * PITCH);
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = d * 32 + 8 * g + 4 * hh;
          *(u32x2*)(stg + (qi * 32 + ql) * PITCH + d0 * 2) =
              u32x2{pack_bf16x2(o_acc[qi][d][4 * g] * inv[qi], o_acc[qi][d][4 * g + 1] * inv[qi]),
                    pack_bf16x2(o_acc[qi][d][4 * g + 2] * inv[qi], o_acc[qi][d][4 * g + 3] * inv[qi])};
        }
    __builtin_amdgcn_wave_barrier();                    // same wave, in-order LDS queue
#pragma unroll
    for (int it = 0; it < QB * 32 / RPI; ++it) {
      const int row = it * RPI + lane / LPR, cc = lane % LPR;
      const int q = qw[row >> 5] + (row & 31);
      const u32x4 v = *(const u32x4*)(stg + row * PITCH + cc * 16);
      if (q < len) *(u32x4*)(p.O + (int64_t)(row0 + q) * p.ldo + head * D + cc * 8) = v;
    }
    return;
  }
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    if (qpos[qi] < len) {
      unsigned short* op = p.O + (int64_t)(row0 + qpos[qi]) * p.ldo + head * D;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = d * 32 + 8 * g + 4 * hh;
          u32x2 pk = {pack_bf16x2(o_acc[qi][d][4 * g] * inv[qi], o_acc[qi][d][4 * g + 1] * inv[qi]),
                      pack_bf16x2(o_acc[qi][d][4 * g + 2] * inv[qi], o_acc[qi][d][4 * g + 3] * inv[qi])};
          *(u32x2*)(op + d0) = pk;
        }
      }
    }
  }
}


// =====================================================================================================================
// D = 64 without bias (Whisper / Qwen2-Audio encoder, Q-Former): the kernel above with its stages INTERLEAVED by hand.
//
// Measured on gfx950 (tools/ubench/shadow.hip, profiles/r02_ubench_mfma_shadow.txt): vector work overlaps an MFMA only when
// it comes from the SAME wave and sits right behind it in the instruction stream; v_exp / v_cvt_pk_bf16 / v_max3 / v_add hide
// 50-80 % of their cost there, the packed f32 forms (v_pk_fma / v_pk_add / v_pk_mul) hide nothing (they cost MORE next to
// an MFMA than alone).  The generic kernel's tile is QK^T -> softmax -> PV, each stage waiting for the one before, so nothing
// can sit behind its MFMAs and its time is the sum of its phases.  Here a wave's two 32-query blocks are staggered:
//     A   S0 = K Q0^T                      (8 MFMAs)
//     B   S1 = K Q1^T                      (8 MFMAs)   with   exponentials of block 0
//     C   O0 += V^T P0                     (8 MFMAs)   with   exponentials of block 1
//     D   O1 += V^T P1                     (8 MFMAs)
// K and V fragments are read once per tile into registers (the V fragments take over the K fragments' registers) and feed
// both blocks, exactly as above; the scalar (non-packed) score math keeps the shadow usable.  Arithmetic per query is that of
// the generic kernel operation for operation (k-step order per accumulator, softmax formulas, sequential row sum).
// (the causal form is not on any model's path here: it gets the registers it asks for instead of spilling at two waves per SIMD)
constexpr int IL64_NW = 4;   // waves per workgroup of the interleaved kernel (64 queries each).  An 8-wave form (512 queries share a
                             // staged K / V tile, half the LDS-DMA instructions per wave) was bit-identical and 2-3.5 % SLOWER on both
                             // encoder shapes (profiles/r02_attn_ablations.log); it is gone: one instantiation, its invariants asserted
template <bool CAUSAL, int NW>
__global__ __launch_bounds__(64 * NW, CAUSAL ? 1 : 2) void attn_fwd_il64_kernel(AttnParams p) {
  constexpr int D = 64, QB = 2, BQ = 32 * NW * QB, ROWB = D * 2, KS = D / 16, DB = D / 32, CPR = D / 8;
  constexpr int NCH = 64 * CPR / (64 * NW), RPI = 64 / CPR, BUF = 2 * 64 * ROWB, NBUF = 3;
  constexpr int EPI = NW * QB * 32 * (D * 2 + 16);       // the epilogue's per-wave transposition buffers reuse the ring
  static_assert(NW == 4 && NCH == 2, "the counted waits below (vmcnt(4) = one tile of 2 K + 2 V LDS-DMA per wave in flight) are for 4 waves");
  static_assert(EPI <= NBUF * BUF, "the epilogue staging must fit inside the K / V ring it reuses");
  __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;
  const int n_blocks = gridDim.x;                       // XCD-aware decode, as above
  const int xcd = blockIdx.x & 7, q8 = n_blocks >> 3, r8 = n_blocks & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int qblk = item % p.n_qblocks;
  const int head = (item / p.n_qblocks) % p.n_heads, seq = item / (p.n_qblocks * p.n_heads);
  const int row0 = p.cu[seq];
  const int len = p.cu[seq + 1] - row0;
  const int qb = qblk * BQ;
  if (qb >= len) return;
  int kvlen = len;
  if (p.kv_lens) kvlen = min(max(p.kv_lens[seq], 1), len);
  const int kv_end = CAUSAL ? min(kvlen, qb + BQ) : kvlen;
  const int n_tiles = (kv_end + 63) >> 6;

  int qw[QB], qpos[QB];
  bf16x8 qf[QB][KS];
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    qw[qi] = qb + (wave * QB + qi) * 32;
    qpos[qi] = qw[qi] + ql;
    const int qrow = min(qpos[qi], len - 1);
    const unsigned short* qp = p.Q + (int64_t)(row0 + qrow) * p.ldq + head * D + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qi][ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // ---- staging: LDS-DMA with source-side swizzle (identical to the kernel above) ---------------------------------------
  auto f_k = [](int row) { return (row >> 1) & 7; };
  auto f_v = [](int row) { return ((row >> 1) & 1) << 2; };
  const int64_t kv_off = p.kv_seq_stride ? (int64_t)seq * p.kv_seq_stride + (int64_t)head * p.kv_head_stride : -1;
  const unsigned short* kbase = kv_off >= 0 ? p.K + kv_off : p.K + (int64_t)row0 * p.ldk + head * D;
  const unsigned short* vbase = kv_off >= 0 ? p.V + kv_off : p.V + (int64_t)row0 * p.ldv + head * D;
  const unsigned short* kptr[NCH];
  const unsigned short* vptr[NCH];
  int srow[NCH], kch[NCH], vch[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int j = wave * NCH + i;
    srow[i] = RPI * j + lane / CPR;
    kch[i] = (lane % CPR) ^ f_k(srow[i]);
    vch[i] = (lane % CPR) ^ f_v(srow[i]);
    kptr[i] = kbase + (int64_t)srow[i] * p.ldk + kch[i] * 8;
    vptr[i] = vbase + (int64_t)srow[i] * p.ldv + vch[i] * 8;
  }
  const int64_t kstep = 64 * p.ldk, vstep = 64 * p.ldv;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto stage_tile = [&](int t, int buf) {
    const int k0 = t * 64;
    char* k_w = lds + buf * BUF + wave * NCH * 1024;
    char* v_w = k_w + 64 * ROWB;
    if (k0 + 64 <= len) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(kptr[i] + (int64_t)t * kstep), (lptr_t)(k_w + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vptr[i] + (int64_t)t * vstep), (lptr_t)(v_w + i * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int64_t grow = min(k0 + srow[i], len - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(kbase + grow * p.ldk + kch[i] * 8), (lptr_t)(k_w + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vbase + grow * p.ldv + vch[i] * 8), (lptr_t)(v_w + i * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 o_acc[QB][DB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    m_run[qi] = NEG_BIG;
    l_run[qi] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[qi][d][r] = 0.f;
  }
  int k_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_off[ks] = ql * ROWB + (((2 * ks + hh) ^ f_k(ql)) << 4);
  int tr_off[DB];
  {
    const int q = (lane & 15) >> 2;
    const int c2 = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
    for (int d = 0; d < DB; ++d) tr_off[d] = (4 * hh + q) * ROWB + (((4 * d + c2) ^ f_v(q)) << 4) + (lane & 1) * 8;
  }

  // softmax of one 32-query block in two parts: head = row maximum, alpha, rescale of O (its wave-uniform branch closes a
  // basic block, so it sits BEFORE the stretch that holds MFMAs); tail = exponentials -> P fragments, row sum
  auto sm_head = [&](const int qi, f32x16 (&s)[2], const int k0, auto maymask_c) -> bool {
    constexpr bool MAYMASK = decltype(maymask_c)::value;
    const bool need_mask =
        MAYMASK && __builtin_amdgcn_readfirstlane((int)((k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw[qi]))));
    float tmax;
    if (!need_mask) {
      tmax = max_xhalf(max32(s[0], s[1])) * p.scale_log2e;   // scale > 0: max commutes with the scaling
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = (key < kvlen) && (!CAUSAL || key <= qpos[qi]);
          s[kb][r] = ok ? s[kb][r] * p.scale_log2e : NEG_BIG;
        }
      tmax = max_xhalf(max32(s[0], s[1]));
    }
    const float m_new = __builtin_fmaxf(m_run[qi], tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
    m_run[qi] = m_new;
    l_run[qi] *= alpha;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[qi][d][r] *= alpha;
    }
    return need_mask;
  };
  auto sm_tail = [&](const int qi, f32x16 (&s)[2], const bool need_mask, bf16x8 (&pf)[2][2]) {
    float psum = 0.f;
    const float m = m_run[qi];
    if (!need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], p.scale_log2e, -m));
          psum += e;
          pf[kb][r >> 3][r & 7] = (__bf16)e;
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float e = __builtin_amdgcn_exp2f(s[kb][r] - m);
          e = (s[kb][r] <= NEG_BIG * 0.5f) ? 0.f : e;
          psum += e;
          pf[kb][r >> 3][r & 7] = (__bf16)e;
        }
    }
    l_run[qi] += psum;
  };

  stage_tile(0, 0);
  stage_tile(min(1, n_tiles - 1), 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // tile 0 (this wave's 4 oldest LDS-DMA) has landed, tile 1 stays in flight
  __syncthreads();
  int cur = 0;

  auto tile = [&](const int t, auto maymask_c) {
    int nxt = cur + 2;
    if (nxt >= NBUF) nxt -= NBUF;
    stage_tile(min(t + 2, n_tiles - 1), nxt);
    const int k0 = t * 64;
    const char* k_lds = lds + cur * BUF;
    const char* v_lds = k_lds + 64 * ROWB;
    // all eight K fragments of the tile, k-step major (asm reads, counted waits: see the generic kernel)
    bf16x8 kf[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const unsigned a = (unsigned)(uintptr_t)(k_lds + kb * 32 * ROWB + k_off[ks]);
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[ks][kb]) : "v"(a));
      }
    f32x16 s0[2], s1[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[kb][r] = s1[kb][r] = 0.f;
    // ---- A: S0 = K Q0^T ------------------------------------------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(kf[0][0]), "+v"(kf[0][1]));
      if (ks == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[1][0]), "+v"(kf[1][1]));
      if (ks == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kf[2][0]), "+v"(kf[2][1]));
      if (ks == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[3][0]), "+v"(kf[3][1]));
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s0[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks][kb], qf[0][ks], s0[kb], 0, 0, 0);
    }
    const bool nm0 = sm_head(0, s0, k0, maymask_c);
    // ---- B: S1 = K Q1^T  with  the exponentials of block 0 ---------------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s1[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks][kb], qf[1][ks], s1[kb], 0, 0, 0);
    bf16x8 pf0[2][2], pf1[2][2];
    sm_tail(0, s0, nm0, pf0);
    // pin: without it hipcc sinks block 0's exponentials below the next wave-uniform branch, next to their first use (C),
    // and stretch B is left with bare MFMAs
    asm volatile("" : "+v"(pf0[0][0]), "+v"(pf0[0][1]), "+v"(pf0[1][0]), "+v"(pf0[1][1]));
    // all eight V^T fragments (transposed reads), issued ahead of block 1's row maximum
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x4 vr[2][2][DB][2];                  // [key half][k-step][d-block][lo / hi]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const unsigned a = (unsigned)(uintptr_t)(v_lds + (kb * 32 + 16 * s) * ROWB + tr_off[d]);
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vr[kb][s][d][0]) : "v"(a));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[kb][s][d][1]) : "v"(a), "n"(8 * ROWB));
        }
    const bool nm1 = sm_head(1, s1, k0, maymask_c);
    // ---- C: O0 += V^T P0  with  the exponentials of block 1 ----------------------------------------------------------------
    bf16x8 vf[2][2][DB];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (kb == 0 && s == 0) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(vr[0][0][0][0]), "+v"(vr[0][0][0][1]), "+v"(vr[0][0][1][0]), "+v"(vr[0][0][1][1]));
        if (kb == 0 && s == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vr[0][1][0][0]), "+v"(vr[0][1][0][1]), "+v"(vr[0][1][1][0]), "+v"(vr[0][1][1][1]));
        if (kb == 1 && s == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vr[1][0][0][0]), "+v"(vr[1][0][0][1]), "+v"(vr[1][0][1][0]), "+v"(vr[1][0][1][1]));
        if (kb == 1 && s == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[1][1][0][0]), "+v"(vr[1][1][0][1]), "+v"(vr[1][1][1][0]), "+v"(vr[1][1][1][1]));
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const s16x8 both = __builtin_shufflevector(vr[kb][s][d][0], vr[kb][s][d][1], 0, 1, 2, 3, 4, 5, 6, 7);
          vf[kb][s][d] = __builtin_bit_cast(bf16x8, both);
          o_acc[0][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][s][d], pf0[kb][s], o_acc[0][d], 0, 0, 0);
        }
      }
    sm_tail(1, s1, nm1, pf1);
    // ---- D: O1 += V^T P1 -------------------------------------------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int d = 0; d < DB; ++d)
          o_acc[1][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][s][d], pf1[kb][s], o_acc[1][d], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this wave's share of tile t+1 has landed; t+2 stays in flight
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == NBUF ? 0 : cur + 1;
  };
  int t = 0;
  {
    const int n_plain = min(n_tiles, CAUSAL ? min(kvlen, qw[0] + 1) >> 6 : kvlen >> 6);
    for (; t < n_plain; ++t) tile(t, std::false_type{});
  }
  for (; t < n_tiles; ++t) tile(t, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue (as above) -----------------------------------------------------------------------------------------------
  float inv[QB];
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
    inv[qi] = l_tot > 0.f ? 1.f / l_tot : 0.f;
  }
  const bool rows16 = (((uintptr_t)p.O | (uintptr_t)(p.ldo * 2)) & 15) == 0;
  if (rows16) {
    constexpr int PITCH = D * 2 + 16;
    constexpr int LPR = D * 2 / 16, RPS = 64 / LPR;
    char* stg = lds + wave * (QB * 32 * PITCH);
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = d * 32 + 8 * g + 4 * hh;
          *(u32x2*)(stg + (qi * 32 + ql) * PITCH + d0 * 2) =
              u32x2{pack_bf16x2(o_acc[qi][d][4 * g] * inv[qi], o_acc[qi][d][4 * g + 1] * inv[qi]),
                    pack_bf16x2(o_acc[qi][d][4 * g + 2] * inv[qi], o_acc[qi][d][4 * g + 3] * inv[qi])};
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < QB * 32 / RPS; ++it) {
      const int row = it * RPS + lane / LPR, cc = lane % LPR;
      const int q = qw[row >> 5] + (row & 31);
      const u32x4 v = *(const u32x4*)(stg + row * PITCH + cc * 16);
      if (q < len) *(u32x4*)(p.O + (int64_t)(row0 + q) * p.ldo + head * D + cc * 8) = v;
    }
    return;
  }
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    if (qpos[qi] < len) {
      unsigned short* op = p.O + (int64_t)(row0 + qpos[qi]) * p.ldo + head * D;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = d * 32 + 8 * g + 4 * hh;
          *(u32x2*)(op + d0) = u32x2{pack_bf16x2(o_acc[qi][d][4 * g] * inv[qi], o_acc[qi][d][4 * g + 1] * inv[qi]),
                                     pack_bf16x2(o_acc[qi][d][4 * g + 2] * inv[qi], o_acc[qi][d][4 * g + 3] * inv[qi])};
        }
    }
  }
}

template <int D>
int launch_attn(const AttnParams& p, const icl_attn_args* a, hipStream_t stream) {
  const int bq = (D == 64 && !a->rel_bias) ? IL64_NW * 64 : 128;   // queries per workgroup
  dim3 grid(((a->max_seqlen + bq - 1) / bq) * a->n_heads * a->n_seqs, 1, 1);
  const bool bias = a->rel_bias != nullptr;
  if (D == 64 && !bias) {
    if (a->causal) hipLaunchKernelGGL((attn_fwd_il64_kernel<true, IL64_NW>), grid, dim3(64 * IL64_NW), 0, stream, p);
    else hipLaunchKernelGGL((attn_fwd_il64_kernel<false, IL64_NW>), grid, dim3(64 * IL64_NW), 0, stream, p);
    ICL_CHECK_LAUNCH("icl_attn_fwd_bf16");
    return ICL_OK;
  }
  if (a->causal) {
    if (bias)
      hipLaunchKernelGGL((attn_fwd_kernel<D, true, true>), grid, dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<D, true, false>), grid, dim3(256), 0, stream, p);
  } else {
    if (bias)
      hipLaunchKernelGGL((attn_fwd_kernel<D, false, true>), grid, dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<D, false, false>), grid, dim3(256), 0, stream, p);
  }
  ICL_CHECK_LAUNCH("icl_attn_fwd_bf16");
  return ICL_OK;
}

}  // namespace

extern "C" int icl_attn_fwd_bf16(const icl_attn_args* a, void* stream) {
  ICL_CHECK_ARG(a != nullptr, "icl_attn_fwd_bf16: args is NULL");
  ICL_CHECK_ARG(a->Q && a->K && a->V && a->O && a->cu_seqlens, "icl_attn_fwd_bf16: NULL pointer");
  ICL_CHECK_ARG(a->head_dim == 64 || a->head_dim == 128, "icl_attn_fwd_bf16: head_dim=%d (only 64 and 128)", a->head_dim);
  ICL_CHECK_ARG(a->n_seqs > 0 && a->n_seqs <= 65535 && a->n_heads > 0 && a->n_heads <= 65535 && a->max_seqlen > 0,
                "icl_attn_fwd_bf16: bad n_seqs/n_heads/max_seqlen");
  ICL_CHECK_ARG(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->ldo % 4 == 0,
                "icl_attn_fwd_bf16: leading dimensions must be multiples of 8 (ldo: 4)");
  ICL_CHECK_ARG(((uintptr_t)a->Q & 15) == 0 && ((uintptr_t)a->K & 15) == 0 && ((uintptr_t)a->V & 15) == 0 &&
                    ((uintptr_t)a->O & 7) == 0,
                "icl_attn_fwd_bf16: Q/K/V must be 16-byte and O 8-byte aligned");
  ICL_CHECK_ARG((a->rel_bias == nullptr) == (a->rel_gate == nullptr), "icl_attn_fwd_bf16: rel_bias and rel_gate go together");
  if (a->rel_bias) ICL_CHECK_ARG(a->rel_span >= 1, "icl_attn_fwd_bf16: rel_span must be >= 1");
  AttnParams p;
  p.Q = (const unsigned short*)a->Q;
  p.K = (const unsigned short*)a->K;
  p.V = (const unsigned short*)a->V;
  p.O = (unsigned short*)a->O;
  p.cu = a->cu_seqlens;
  p.kv_lens = a->kv_lens;
  p.rel_bias = a->rel_bias;
  p.rel_gate = a->rel_gate;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
  ICL_CHECK_ARG((a->kv_seq_stride == 0) == (a->kv_head_stride == 0) && a->kv_seq_stride >= 0 && a->kv_head_stride >= 0 &&
                    a->kv_seq_stride % 8 == 0 && a->kv_head_stride % 8 == 0,
                "icl_attn_fwd_bf16: kv_seq_stride / kv_head_stride must both be 0 or both positive multiples of 8");
  p.kv_seq_stride = a->kv_seq_stride; p.kv_head_stride = a->kv_head_stride;
  p.n_heads = a->n_heads;
  p.rel_span = a->rel_span;
  const int bq = (a->head_dim == 64 && !a->rel_bias) ? IL64_NW * 64 : 128;
  p.n_qblocks = (a->max_seqlen + bq - 1) / bq;
  p.scale_log2e = a->scale * LOG2E;
  return a->head_dim == 64 ? launch_attn<64>(p, a, (hipStream_t)stream) : launch_attn<128>(p, a, (hipStream_t)stream);
}
