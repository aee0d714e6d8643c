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
//   * K/V tiles are register-staged (global -> VGPR issued before the compute of the current
//     tile, ds_write into the OTHER LDS buffer after it: T14; one barrier per tile), rows padded (K: +16 B, V: +64 B) so that both the
//     ds_read_b128 K-fragment reads and the transposed V reads are bank-conflict free.
// Variable-length packing (cu_seqlens), causal masking, a key-padding length per sequence and the
// BEATs gated relative-position bias are handled in the score stage.
#include "common.h"

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
  int n_heads, rel_span;
  float scale_log2e;
};

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

template <int D, bool CAUSAL, bool BIAS>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
  constexpr int KSTR = D * 2 + 16;  // bytes per K row in LDS
  constexpr int VSTR = D * 2 + 64;  // bytes per V row in LDS
  constexpr int KS = D / 16;        // QK^T k-steps
  constexpr int DB = D / 32;        // output d-blocks
  constexpr int CPR = D / 8;        // 16-B chunks per row
  constexpr int NCH = 64 * CPR / 256;  // staging chunks per thread per tensor
  constexpr int BUF = 64 * KSTR + 64 * VSTR;   // one K tile + one V tile
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];   // double-buffered: ONE barrier per KV tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int row0 = p.cu[seq];
  const int len = p.cu[seq + 1] - row0;
  const int qb = blockIdx.x * 128;
  if (qb >= len) return;
  int kvlen = len;
  if (p.kv_lens) kvlen = min(max(p.kv_lens[seq], 1), len);
  const int kv_end = CAUSAL ? min(kvlen, qb + 128) : kvlen;
  const int n_tiles = (kv_end + 63) >> 6;

  const int qw = qb + wave * 32;  // first query of this wave
  const int qpos = qw + ql;       // this lane's query (relative to the sequence)
  const int qrow = min(qpos, len - 1);

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q, hh) holds Q[q][16ks + 8hh .. +7] ----------
  bf16x8 qf[KS];
  {
    const unsigned short* qp = p.Q + (int64_t)(row0 + qrow) * p.ldq + head * D + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }
  float gate = 0.f;
  const float* bias_row = nullptr;
  if (BIAS) {
    gate = p.rel_gate[(int64_t)(row0 + qrow) * p.n_heads + head] * LOG2E;
    bias_row = p.rel_bias + (int64_t)head * (2 * p.rel_span - 1) + (p.rel_span - 1);
  }

  // ---- staging -------------------------------------------------------------------------------------
  u32x4 kreg[NCH], vreg[NCH];
  auto load_tile = [&](int t) {
    const int k0 = t * 64;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      const int r = c / CPR, cc = c - r * CPR;
      const int64_t grow = row0 + min(k0 + r, len - 1);
      kreg[i] = *(const u32x4*)(p.K + grow * p.ldk + head * D + cc * 8);
      vreg[i] = *(const u32x4*)(p.V + grow * p.ldv + head * D + cc * 8);
    }
  };
  auto write_tile = [&](int buf) {
    char* k_w = lds + buf * BUF;
    char* v_w = k_w + 64 * KSTR;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      const int r = c / CPR, cc = c - r * CPR;
      *(u32x4*)(k_w + r * KSTR + cc * 16) = kreg[i];
      *(u32x4*)(v_w + r * VSTR + cc * 16) = vreg[i];
    }
  };

  f32x16 o_acc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  // transposed-read lane address pieces (bytes): row (i>>2), column 4*(i&3) of the 16-lane group
  const int tr_lane_off = ((lane & 15) >> 2) * VSTR + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);

  load_tile(0);
  write_tile(0);
  __syncthreads();

  for (int t = 0; t < n_tiles; ++t) {
    if (t + 1 < n_tiles) load_tile(t + 1);
    const int k0 = t * 64;
    const char* k_lds = lds + (t & 1) * BUF;
    const char* v_lds = k_lds + 64 * KSTR;
    // wave-uniform: does this wave have any visible key in this tile?
    const bool active = !CAUSAL || (k0 <= qw + 31);
    if (active) {
      // ---- S^T = K Q^T ---------------------------------------------------------------------------
      f32x16 s_acc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
        const char* kp = k_lds + (kb * 32 + ql) * KSTR + hh * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 kf = *(const bf16x8*)(kp + ks * 32);
          s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_acc[kb], 0, 0, 0);
        }
      }
      // ---- scores -> base-2 logits, bias, mask -----------------------------------------------------
      // need_mask is wave-uniform: interior tiles take the branch-free fast path (raw v_exp_f32, one FMA per score)
      const bool need_mask = __builtin_amdgcn_readfirstlane((int)((k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw))));
      float psum = 0.f, alpha;
      bf16x8 pf[2][2];
      if (!BIAS && !need_mask) {
        float tmax = fmaxf(s_acc[0][0], s_acc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s_acc[0][r], s_acc[1][r]));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * p.scale_log2e;
        const float m_new = fmaxf(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s_acc[kb][r], p.scale_log2e, -m_new));
            psum += e;
            pf[kb][r >> 3][r & 7] = (__bf16)e;
          }
      } else {
        float tmax = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            float v = s_acc[kb][r] * p.scale_log2e;
            if (BIAS) {
              int rel = key - qpos;
              rel = max(-(p.rel_span - 1), min(p.rel_span - 1, rel));
              v += gate * bias_row[rel];
            }
            if (need_mask) {
              const bool ok = (key < kvlen) && (!CAUSAL || key <= qpos);
              v = ok ? v : NEG_BIG;
            }
            s_acc[kb][r] = v;
            tmax = fmaxf(tmax, v);
          }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float e = __builtin_amdgcn_exp2f(s_acc[kb][r] - m_new);
            if (need_mask) e = (s_acc[kb][r] <= NEG_BIG * 0.5f) ? 0.f : e;
            psum += e;
            pf[kb][r >> 3][r & 7] = (__bf16)e;
          }
        }
      }
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
      // ---- O^T += V^T P^T -----------------------------------------------------------------------------
#pragma unroll
      for (int d = 0; d < DB; ++d) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const char* vp = v_lds + (kb * 32 + 16 * s + 4 * hh) * VSTR + d * 64 + tr_lane_off;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 8 * VSTR));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
            o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s], o_acc[d], 0, 0, 0);
          }
        }
      }
    }
    // the other buffer was last read in iteration t-1, and every wave has passed the barrier that ended it
    if (t + 1 < n_tiles) write_tile((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l -------------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qpos < len) {
    unsigned short* op = p.O + (int64_t)(row0 + qpos) * p.ldo + head * D;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = d * 32 + 8 * g + 4 * hh;
        u32x2 pk = {pack_bf16x2(o_acc[d][4 * g] * inv, o_acc[d][4 * g + 1] * inv),
                    pack_bf16x2(o_acc[d][4 * g + 2] * inv, o_acc[d][4 * g + 3] * inv)};
        *(u32x2*)(op + d0) = pk;
      }
    }
  }
}

template <int D>
int launch_attn(const AttnParams& p, const icl_attn_args* a, hipStream_t stream) {
  dim3 grid((a->max_seqlen + 127) / 128, a->n_heads, a->n_seqs);
  const bool bias = a->rel_bias != nullptr;
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
  p.n_heads = a->n_heads;
  p.rel_span = a->rel_span;
  p.scale_log2e = a->scale * LOG2E;
  return a->head_dim == 64 ? launch_attn<64>(p, a, (hipStream_t)stream) : launch_attn<128>(p, a, (hipStream_t)stream);
}
