// attn_decode.hip — single-token decode attention over the KV cache (HBM-bound KV stream).
// Cache layout [n_seqs][n_heads][max_len][D] bf16: the keys of one (sequence, head) are one
// contiguous stream, read with 16-B loads straight to VGPRs (no LDS round trip: guide §5 table row
// "GEMV / M <= 16 decode").  D/8 lanes cover one key row, so a wave-instruction fetches 64/(D/8)
// consecutive keys (1 KiB).  Every lane keeps an online-softmax stream (m, l, o[8]) for its
// (key slot, d-chunk); the 4*64/(D/8) streams of a block are merged once through LDS.
#include "common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

// UNR = key rounds (4 * KPW keys each) whose K and V loads are issued before the first of them is consumed.  A stream's keys and
// their order do not depend on it, so the result is bit-identical for any UNR: few workgroups (a small decode batch: one block
// per (sequence, head), nothing else on the CU to hide a 2-3 us round trip per loop iteration) take 16, a full chip takes 4.
// FUSE (icl_attn_decode_rope_bf16): the step's RoPE + cache append run HERE instead of in a launch of their own (rope_kv_kernel:
// ~5 us per layer at any decode batch, one of the nine launches of a one-sequence decode layer).  Q holds the projection's raw
// q | k | v row of sequence b; every lane rotates the q chunk it needs (and the new key's chunk) with rope_rot8 — the function and
// the bf16 rounding points of rope_kv_kernel, so the result is bit-identical to the two launches — the first LPR lanes append the
// rotated key and the value to the cache at position pos[b], and in the key loop position pos[b] is served from registers, never
// from the cache line that is being written.
struct DecodeRope {
  const float* cosT;
  const float* sinT;
  const int* pos;
  const int* seq_ids;     // cache row of sequence b (NULL: b)
  unsigned short* kc;     // the caches, writable (Kc / Vc of the kernel are these)
  unsigned short* vc;
  int64_t k_off, v_off;   // column offsets of k / v in the qkv row
};

template <int D, int UNR, bool FUSE>
__global__ __launch_bounds__(256) void attn_decode_kernel(const unsigned short* Q, int64_t ldq,
                                                           const unsigned short* Kc, const unsigned short* Vc,
                                                           unsigned short* O, int64_t ldo, const int* lens,
                                                           int n_heads, int max_len, float scale_log2e, DecodeRope rp) {
  constexpr int LPR = D / 8;     // lanes per key row
  constexpr int KPW = 64 / LPR;  // keys per wave-instruction
  constexpr int NSTREAM = 4 * KPW;
  __shared__ float sm[NSTREAM][D + 2];  // per stream: o[D], m, l
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dc = lane % LPR, sub = lane / LPR;
  const int len = min(lens[b], max_len);
  const int crow = FUSE && rp.seq_ids ? rp.seq_ids[b] : b;
  const int64_t base = ((int64_t)crow * n_heads + h) * (int64_t)max_len * D;
  const unsigned short* kp = Kc + base + dc * 8;
  const unsigned short* vp = Vc + base + dc * 8;

  u32x4 q_raw, k_new = {0u, 0u, 0u, 0u}, v_new = {0u, 0u, 0u, 0u};
  int pos_new = -1;
  if constexpr (FUSE) {
    constexpr int HALF = D / 2;
    pos_new = rp.pos[b];
    const int i0 = (dc % (LPR / 2)) * 8;                 // this lane's 8 elements of the low half (its partner chunk: + HALF)
    const bool is_hi = dc >= LPR / 2;
    const unsigned short* row = Q + (int64_t)b * ldq + h * D;
    const u32x4 qlo = *(const u32x4*)(row + i0), qhi = *(const u32x4*)(row + i0 + HALF);
    const u32x4 klo = *(const u32x4*)(row + rp.k_off + i0), khi = *(const u32x4*)(row + rp.k_off + i0 + HALF);
    v_new = *(const u32x4*)(row + rp.v_off + dc * 8);
    const float* cp = rp.cosT + (int64_t)pos_new * HALF + i0;
    const float* sp = rp.sinT + (int64_t)pos_new * HALF + i0;
    const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4), s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
    u32x4 olo, ohi;
    rope_rot8(qlo, qhi, c0, c1, s0, s1, olo, ohi);
    q_raw = is_hi ? ohi : olo;
    rope_rot8(klo, khi, c0, c1, s0, s1, olo, ohi);
    k_new = is_hi ? ohi : olo;
    if (wave == 0 && sub == 0 && pos_new >= 0 && pos_new < max_len) {   // one lane per 16-B chunk: the appended row
      *(u32x4*)(rp.kc + base + (int64_t)pos_new * D + dc * 8) = k_new;
      *(u32x4*)(rp.vc + base + (int64_t)pos_new * D + dc * 8) = v_new;
    }
  } else {
    q_raw = *(const u32x4*)(Q + (int64_t)b * ldq + h * D + dc * 8);
  }
  float q[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    q[2 * t] = __uint_as_float(q_raw[t] << 16) * scale_log2e;
    q[2 * t + 1] = __uint_as_float(q_raw[t] & 0xffff0000u) * scale_log2e;
  }
  float m = NEG_BIG, l = 0.f, o[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) o[t] = 0.f;

  // A stream folds its keys in GROUPS of G = 4 (the same groups whatever UNR is: bit-identical for any UNR): the four scores of
  // a group are independent dot products, ONE running-maximum update and ONE rescale serve all four, and their exponentials and
  // the o / l updates are independent again.  The per-key form was one serial chain per key — maximum, two exponentials, nine
  // dependent FMAs — 24 links deep for a 385-key prompt at one sequence per block (22 us per call at a decode batch of 1).
  constexpr int G = 4;
  static_assert(UNR % G == 0, "key rounds are consumed in groups of four");
  for (int j0 = wave * KPW; j0 < len; j0 += 4 * KPW * UNR) {
    u32x4 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int key = min(j0 + u * 4 * KPW + sub, len - 1);
      kr[u] = *(const u32x4*)(kp + (int64_t)key * D);
      vr[u] = *(const u32x4*)(vp + (int64_t)key * D);
      if (FUSE && key == pos_new) {      // the row this launch appends: from registers (the store above may not have landed)
        kr[u] = k_new;
        vr[u] = v_new;
      }
    }
#pragma unroll
    for (int g = 0; g < UNR / G; ++g) {
      float s[G];
      bool ok[G];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int u = g * G + i;
        float d = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          d += q[2 * t] * __uint_as_float(kr[u][t] << 16);
          d += q[2 * t + 1] * __uint_as_float(kr[u][t] & 0xffff0000u);
        }
#pragma unroll
        for (int x = 1; x < LPR; x <<= 1) d += __shfl_xor(d, x, 64);
        ok[i] = j0 + u * 4 * KPW + sub < len;
        s[i] = ok[i] ? d : NEG_BIG;
      }
      const float m_new = fmaxf(fmaxf(m, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      float pe[G];
#pragma unroll
      for (int i = 0; i < G; ++i) pe[i] = ok[i] ? __builtin_amdgcn_exp2f(s[i] - m_new) : 0.f;
      m = m_new;
      l = l * alpha + ((pe[0] + pe[1]) + (pe[2] + pe[3]));
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float a0 = o[2 * t] * alpha, a1 = o[2 * t + 1] * alpha;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          a0 = fmaf(pe[i], __uint_as_float(vr[g * G + i][t] << 16), a0);
          a1 = fmaf(pe[i], __uint_as_float(vr[g * G + i][t] & 0xffff0000u), a1);
        }
        o[2 * t] = a0;
        o[2 * t + 1] = a1;
      }
    }
  }
  const int stream = wave * KPW + sub;
#pragma unroll
  for (int t = 0; t < 8; ++t) sm[stream][dc * 8 + t] = o[t];
  if (dc == 0) {
    sm[stream][D] = m;
    sm[stream][D + 1] = l;
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const int d = threadIdx.x;
    float M = NEG_BIG;
#pragma unroll
    for (int s = 0; s < NSTREAM; ++s) M = fmaxf(M, sm[s][D]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int s = 0; s < NSTREAM; ++s) {
      const float w = __builtin_amdgcn_exp2f(sm[s][D] - M);
      L += sm[s][D + 1] * w;
      acc += sm[s][d] * w;
    }
    O[(int64_t)b * ldo + h * D + d] = f32_to_bf16_bits(L > 0.f ? acc / L : 0.f);
  }
}

}  // namespace

static int launch_attn_decode(const void* Q, int64_t ldq, const void* Kc, const void* Vc, void* O, int64_t ldo, const int32_t* lens,
                              int32_t n_seqs, int32_t n_heads, int32_t head_dim, int32_t max_len, float scale, const DecodeRope* rope,
                              void* stream, const char* who) {
  dim3 grid(n_heads, n_seqs);
  const bool few = (int64_t)n_seqs * n_heads <= 1024;      // at most four workgroups per CU: latency-bound, not bandwidth-bound
  const DecodeRope rp = rope ? *rope : DecodeRope{};
#define ICL_DECODE_CASE(DD, UU, FF)                                                                                       \
  hipLaunchKernelGGL((attn_decode_kernel<DD, UU, FF>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)Q, ldq, \
                     (const unsigned short*)Kc, (const unsigned short*)Vc, (unsigned short*)O, ldo, lens, n_heads, max_len, \
                     scale * LOG2E, rp)
  if (rope) {
    if (head_dim == 64) { if (few) ICL_DECODE_CASE(64, 16, true); else ICL_DECODE_CASE(64, 4, true); }
    else                { if (few) ICL_DECODE_CASE(128, 16, true); else ICL_DECODE_CASE(128, 4, true); }
  } else {
    if (head_dim == 64) { if (few) ICL_DECODE_CASE(64, 16, false); else ICL_DECODE_CASE(64, 4, false); }
    else                { if (few) ICL_DECODE_CASE(128, 16, false); else ICL_DECODE_CASE(128, 4, false); }
  }
#undef ICL_DECODE_CASE
  ICL_CHECK_LAUNCH(who);
  return ICL_OK;
}

extern "C" int icl_attn_decode_bf16(const void* Q, int64_t ldq, const void* Kc, const void* Vc, void* O,
                                    int64_t ldo, const int32_t* lens, int32_t n_seqs, int32_t n_heads,
                                    int32_t head_dim, int32_t max_len, float scale, void* stream) {
  ICL_CHECK_ARG(Q && Kc && Vc && O && lens, "icl_attn_decode_bf16: NULL pointer");
  ICL_CHECK_ARG(head_dim == 64 || head_dim == 128, "icl_attn_decode_bf16: head_dim=%d (only 64 and 128)", head_dim);
  ICL_CHECK_ARG(n_seqs > 0 && n_seqs <= 65535 && n_heads > 0 && max_len > 0, "icl_attn_decode_bf16: bad sizes");
  ICL_CHECK_ARG(ldq % 8 == 0 && ((uintptr_t)Q & 15) == 0 && ((uintptr_t)Kc & 15) == 0 && ((uintptr_t)Vc & 15) == 0,
                "icl_attn_decode_bf16: misaligned operands");
  return launch_attn_decode(Q, ldq, Kc, Vc, O, ldo, lens, n_seqs, n_heads, head_dim, max_len, scale, nullptr, stream,
                            "icl_attn_decode_bf16");
}

extern "C" int icl_attn_decode_rope_bf16(const void* qkv, int64_t ld, int64_t k_off, int64_t v_off, const float* cosT,
                                         const float* sinT, const int32_t* pos, const int32_t* seq_ids, void* kcache, void* vcache,
                                         void* O, int64_t ldo, const int32_t* lens, int32_t n_seqs, int32_t n_heads,
                                         int32_t head_dim, int32_t max_len, float scale, void* stream) {
  ICL_CHECK_ARG(qkv && cosT && sinT && pos && kcache && vcache && O && lens, "icl_attn_decode_rope_bf16: NULL pointer");
  ICL_CHECK_ARG(head_dim == 64 || head_dim == 128, "icl_attn_decode_rope_bf16: head_dim=%d (only 64 and 128)", head_dim);
  ICL_CHECK_ARG(n_seqs > 0 && n_seqs <= 65535 && n_heads > 0 && max_len > 0, "icl_attn_decode_rope_bf16: bad sizes");
  ICL_CHECK_ARG(ld % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && k_off >= (int64_t)n_heads * head_dim &&
                    v_off >= k_off + (int64_t)n_heads * head_dim && ld >= v_off + (int64_t)n_heads * head_dim,
                "icl_attn_decode_rope_bf16: q | k | v column blocks must be 8-element aligned and disjoint inside a row");
  ICL_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)kcache & 15) == 0 && ((uintptr_t)vcache & 15) == 0 &&
                    ((uintptr_t)cosT & 15) == 0 && ((uintptr_t)sinT & 15) == 0, "icl_attn_decode_rope_bf16: misaligned operands");
  DecodeRope rp;
  rp.cosT = cosT; rp.sinT = sinT; rp.pos = pos; rp.seq_ids = seq_ids;
  rp.kc = (unsigned short*)kcache; rp.vc = (unsigned short*)vcache; rp.k_off = k_off; rp.v_off = v_off;
  return launch_attn_decode(qkv, ld, kcache, vcache, O, ldo, lens, n_seqs, n_heads, head_dim, max_len, scale, &rp, stream,
                            "icl_attn_decode_rope_bf16");
}
