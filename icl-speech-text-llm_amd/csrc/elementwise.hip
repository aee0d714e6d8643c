// elementwise.hip — HBM-bound glue kernels of the ICL path (gfx950): RoPE + KV append, embedding
// gather/interleave, greedy argmax + EOS bookkeeping, LoRA down-projection, BEATs gate, the
// window-level Q-Former cross attention and a generic axpby/cast.  All loads/stores are 8-16 B
// per lane (Guideline 13); none of these is reshaped into a GEMM.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// RoPE (HF rotate_half pairing) in place on q,k of a fused QKV row + KV-cache append.
// One block per row; work items of 8 elements (16 B).
__global__ __launch_bounds__(256) void rope_kv_kernel(unsigned short* qkv, int64_t ld, int64_t k_off,
                                                       int64_t v_off, const float* cosT,
                                                       const float* sinT, const int* pos,
                                                       const int* seq_ids, unsigned short* kc,
                                                       unsigned short* vc, int H, int D, int max_len) {
  const int64_t m = blockIdx.x;
  const int p = pos[m];
  const int half = D >> 1;
  const int per_head = half >> 3;          // 8-element items per (tensor, head)
  const int rope_items = 2 * H * per_head; // q and k
  unsigned short* row = qkv + m * ld;
  const int64_t cache_row = kc ? ((int64_t)seq_ids[m] * H) * max_len + p : 0;
  // Two rope items and two V items per thread per pass, ALL loads issued before the first use: with one small block per
  // row the kernel is latency-bound on bytes in flight per thread (2.8 TB/s with the loads interleaved with their uses).
  constexpr int U = 2;
  const int v_items = vc ? H * (D >> 3) : 0;
  for (int base = threadIdx.x; base < rope_items || base < v_items; base += U * blockDim.x) {
    u32x4 lo[U], hi[U], vv[U];
    f32x4 c0[U], c1[U], s0[U], s1[U];
    unsigned short* rb[U];
    int i0s[U], hs[U], whichs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = base + u * blockDim.x;
      if (it < rope_items) {
        const int which = it / (H * per_head);          // 0 = q, 1 = k
        const int rem = it - which * H * per_head;
        const int h = rem / per_head, i0 = (rem - h * per_head) * 8;
        unsigned short* bp = row + (which ? k_off : 0) + h * D;
        rb[u] = bp; i0s[u] = i0; hs[u] = h; whichs[u] = which;
        lo[u] = *(const u32x4*)(bp + i0);
        hi[u] = *(const u32x4*)(bp + i0 + half);
        c0[u] = *(const f32x4*)(cosT + (int64_t)p * half + i0);
        c1[u] = *(const f32x4*)(cosT + (int64_t)p * half + i0 + 4);
        s0[u] = *(const f32x4*)(sinT + (int64_t)p * half + i0);
        s1[u] = *(const f32x4*)(sinT + (int64_t)p * half + i0 + 4);
      }
      if (it < v_items) {
        const int h = it / (D >> 3), i0 = (it - h * (D >> 3)) * 8;
        vv[u] = *(const u32x4*)(row + v_off + h * D + i0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = base + u * blockDim.x;
      if (it < rope_items) {
        u32x4 olo, ohi;
        rope_rot8(lo[u], hi[u], c0[u], c1[u], s0[u], s1[u], olo, ohi);
        *(u32x4*)(rb[u] + i0s[u]) = olo;
        *(u32x4*)(rb[u] + i0s[u] + half) = ohi;
        if (whichs[u] == 1 && kc) {
          unsigned short* dst = kc + (cache_row + (int64_t)hs[u] * max_len) * D;
          *(u32x4*)(dst + i0s[u]) = olo;
          *(u32x4*)(dst + i0s[u] + half) = ohi;
        }
      }
      if (it < v_items) {
        const int h = it / (D >> 3), i0 = (it - h * (D >> 3)) * 8;
        *(u32x4*)(vc + (cache_row + (int64_t)h * max_len) * D + i0) = vv[u];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_gather_kernel(const int* src_idx, const unsigned short* table,
                                                            const float* speech, float* out, int H) {
  const int64_t r = blockIdx.x;
  const int idx = src_idx[r];
  float* dst = out + r * H;
  if (idx >= 0) {
    const unsigned short* src = table + (int64_t)idx * H;
    for (int c = threadIdx.x * 8; c < H; c += blockDim.x * 8) {
      const u32x4 raw = *(const u32x4*)(src + c);
      f32x4 a, b;
      a[0] = __uint_as_float(raw[0] << 16); a[1] = __uint_as_float(raw[0] & 0xffff0000u);
      a[2] = __uint_as_float(raw[1] << 16); a[3] = __uint_as_float(raw[1] & 0xffff0000u);
      b[0] = __uint_as_float(raw[2] << 16); b[1] = __uint_as_float(raw[2] & 0xffff0000u);
      b[2] = __uint_as_float(raw[3] << 16); b[3] = __uint_as_float(raw[3] & 0xffff0000u);
      *(f32x4*)(dst + c) = a;
      *(f32x4*)(dst + c + 4) = b;
    }
  } else {
    const float* src = speech + (int64_t)(-idx - 1) * H;
    for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) *(f32x4*)(dst + c) = *(const f32x4*)(src + c);
  }
}

// ---------------------------------------------------------------------------------------------
// greedy argmax (lowest index on ties) + EOS/pad bookkeeping; one block per sequence.
__global__ __launch_bounds__(256) void argmax_eos_kernel(const float* logits, int64_t ldl, int V, int eos_id,
                                                          int eos_id2, int pad_id, int* finished, int* out_tokens,
                                                          int out_stride, int step, int* next_ids) {
  __shared__ float smax[4];
  __shared__ int sidx[4];
  const int b = blockIdx.x;
  const float* row = logits + (int64_t)b * ldl;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    const float x = row[v];
    if (x > best || (x == best && v < bi)) {
      best = x;
      bi = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    smax[w] = best;
    sidx[w] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (smax[i] > best || (smax[i] == best && sidx[i] < bi)) {
        best = smax[i];
        bi = sidx[i];
      }
    if (bi == 0x7fffffff) bi = 0;  // all-NaN row: keep a valid id
    int fin = finished[b];
    const int tok = fin ? pad_id : bi;
    if (tok == eos_id || tok == eos_id2) fin = 1;
    finished[b] = fin;
    out_tokens[(int64_t)b * out_stride + step] = tok;
    next_ids[b] = tok;
  }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_any(const void* p, int64_t off, int dt) {
  return dt == ICL_F32 ? ((const float*)p)[off] : bf16_bits_to_f32(((const unsigned short*)p)[off]);
}
__global__ __launch_bounds__(256) void axpby_cast_kernel(const void* in, int64_t ldi, int in_dtype,
                                                          const void* add, int64_t lda, int add_dtype,
                                                          float alpha, void* out, int64_t ldo,
                                                          int out_dtype, int M, int N) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i - m * N;
    float v = ld_any(in, m * ldi + n, in_dtype) * alpha;
    if (add) v += ld_any(add, m * lda + n, add_dtype);
    if (out_dtype == ICL_F32)
      ((float*)out)[m * ldo + n] = v;
    else
      ((unsigned short*)out)[m * ldo + n] = f32_to_bf16_bits(v);
  }
}

// ---------------------------------------------------------------------------------------------
// LoRA down-projection into the K-augmentation columns for SMALL M (decode): one block per row; wave w owns the rank-rows
// j = w, w + 4, ... in groups of four, and a group's K sweep loads the x chunk once and the four A chunks next to it — five
// independent 16-B loads per lane and step, unrolled four deep, instead of one rank-row at a time behind its own chain of
// L2 round trips (15.8 -> ~5 us at K = 4096, r = 16: the kernel is latency, not bandwidth).  Sums run per lane in ascending
// k, then across the wave.  Large M goes through icl_gemm_bf16 (N = r_total) instead.
__global__ __launch_bounds__(256) void lora_down_kernel(unsigned short* X, int64_t ldx, int K0,
                                                         const unsigned short* A, int64_t lda,
                                                         int r_total, float scale) {
  const int64_t m = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned short* x = X + m * ldx;
  for (int j0 = w; j0 < r_total; j0 += 16) {          // rank-rows j0, j0 + 4, j0 + 8, j0 + 12 (clamped; extra sums are dropped)
    const unsigned short* a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) a[g] = A + (int64_t)min(j0 + 4 * g, r_total - 1) * lda;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = lane * 8; k < K0; k += 64 * 8) {
      const u32x4 xv = *(const u32x4*)(x + k);
      u32x4 av[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) av[g] = *(const u32x4*)(a[g] + k);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[g] += __uint_as_float(xv[t] << 16) * __uint_as_float(av[g][t] << 16);
          s[g] += __uint_as_float(xv[t] & 0xffff0000u) * __uint_as_float(av[g][t] & 0xffff0000u);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float tot = wave_reduce_sum(s[g]);
      if (lane == 0 && j0 + 4 * g < r_total) X[m * ldx + K0 + j0 + 4 * g] = f32_to_bf16_bits(tot * scale);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BEATs gated relative-position gate: one thread per (row, head).
__global__ __launch_bounds__(256) void beats_gate_kernel(const unsigned short* qkv, int64_t ld,
                                                          const float* grep_w, const float* grep_b,
                                                          const float* grep_a, float* gate, int M, int H) {
  __shared__ float w[8 * 64];
  __shared__ float bsh[8];
  for (int i = threadIdx.x; i < 512; i += blockDim.x) w[i] = grep_w[i];
  if (threadIdx.x < 8) bsh[threadIdx.x] = grep_b[threadIdx.x];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * H) return;
  const int64_t m = idx / H;
  const int h = (int)(idx - m * H);
  const unsigned short* q = qkv + m * ld + h * 64;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = bsh[j];
  for (int d0 = 0; d0 < 64; d0 += 8) {
    const u32x4 raw = *(const u32x4*)(q + d0);
    float qv[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qv[2 * t] = __uint_as_float(raw[t] << 16);
      qv[2 * t + 1] = __uint_as_float(raw[t] & 0xffff0000u);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[j] += w[j * 64 + d0 + t] * qv[t];
  }
  const float sa = acc[0] + acc[1] + acc[2] + acc[3], sb = acc[4] + acc[5] + acc[6] + acc[7];
  const float ga = 1.f / (1.f + __expf(-sa)), gb = 1.f / (1.f + __expf(-sb));
  gate[idx] = ga * (gb * grep_a[h] - 1.f) + 2.f;
}

// ---------------------------------------------------------------------------------------------
// window-level Q-Former cross attention: 1 query x `win` (<= 64) keys, head_dim 64.
// One wave per (window, head): lane j < win scores key j; then lane d accumulates output dim d.
__global__ __launch_bounds__(256) void qformer_xattn_kernel(const unsigned short* q, int64_t ldq,
                                                             const unsigned short* kv, int64_t ldkv,
                                                             int64_t v_off, unsigned short* out, int64_t ldo,
                                                             int win_per_audio, int win, int rows_per_audio,
                                                             int H, float scale, int n_pairs) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const int wdx = pair / H, h = pair - wdx * H;
  const int a = wdx / win_per_audio, wi = wdx - a * win_per_audio;
  const int64_t row0 = (int64_t)a * rows_per_audio + (int64_t)wi * win;
  const unsigned short* qp = q + (int64_t)wdx * ldq + h * 64;
  float s = -INFINITY;
  if (lane < win) {
    const unsigned short* kp = kv + (row0 + lane) * ldkv + h * 64;
    float acc = 0.f;
#pragma unroll
    for (int d0 = 0; d0 < 64; d0 += 8) {
      const u32x4 kr = *(const u32x4*)(kp + d0);
      const u32x4 qr = *(const u32x4*)(qp + d0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc += __uint_as_float(kr[t] << 16) * __uint_as_float(qr[t] << 16);
        acc += __uint_as_float(kr[t] & 0xffff0000u) * __uint_as_float(qr[t] & 0xffff0000u);
      }
    }
    s = acc * scale;
  }
  const float mx = wave_reduce_max(s);
  const float e = lane < win ? __expf(s - mx) : 0.f;
  const float denom = wave_reduce_sum(e);
  const float pj = e / denom;
  float o = 0.f;
  for (int j = 0; j < win; ++j) {
    const float pw = __shfl(pj, j, 64);
    o += pw * bf16_bits_to_f32(kv[(row0 + j) * ldkv + v_off + h * 64 + lane]);
  }
  out[(int64_t)wdx * ldo + h * 64 + lane] = f32_to_bf16_bits(o);
}

}  // namespace

extern "C" int icl_rope_kv_bf16(void* qkv, int64_t ld, int64_t k_off, int64_t v_off, const float* cosT,
                                const float* sinT, const int32_t* pos, const int32_t* seq_ids,
                                void* kcache, void* vcache, int32_t M, int32_t n_heads, int32_t head_dim,
                                int32_t max_len, void* stream) {
  ICL_CHECK_ARG(qkv && cosT && sinT && pos, "icl_rope_kv_bf16: NULL pointer");
  ICL_CHECK_ARG(M > 0 && n_heads > 0, "icl_rope_kv_bf16: M and n_heads must be > 0");
  ICL_CHECK_ARG(head_dim % 16 == 0 && head_dim >= 16, "icl_rope_kv_bf16: head_dim=%d must be a multiple of 16", head_dim);
  ICL_CHECK_ARG(ld % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ((uintptr_t)qkv & 15) == 0,
                "icl_rope_kv_bf16: qkv must be 16-byte aligned with ld/k_off/v_off multiples of 8");
  ICL_CHECK_ARG(((uintptr_t)cosT & 15) == 0 && ((uintptr_t)sinT & 15) == 0, "icl_rope_kv_bf16: cos/sin misaligned");
  ICL_CHECK_ARG((kcache == nullptr) == (vcache == nullptr), "icl_rope_kv_bf16: kcache and vcache must both be set or both NULL");
  if (kcache) {
    ICL_CHECK_ARG(seq_ids && max_len > 0, "icl_rope_kv_bf16: cache append needs seq_ids and max_len");
    ICL_CHECK_ARG(((uintptr_t)kcache & 15) == 0 && ((uintptr_t)vcache & 15) == 0, "icl_rope_kv_bf16: cache misaligned");
  }
  hipLaunchKernelGGL(rope_kv_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (unsigned short*)qkv, ld,
                     k_off, v_off, cosT, sinT, pos, seq_ids, (unsigned short*)kcache,
                     (unsigned short*)vcache, n_heads, head_dim, max_len);
  ICL_CHECK_LAUNCH("icl_rope_kv_bf16");
  return ICL_OK;
}

extern "C" int icl_embed_gather_interleave(const int32_t* src_idx, const void* table, const float* speech,
                                           float* out, int32_t rows, int32_t H, int32_t vocab,
                                           int32_t n_speech_rows, void* stream) {
  (void)vocab;
  (void)n_speech_rows;  // range is validated by the caller on the host copy of src_idx
  ICL_CHECK_ARG(src_idx && table && out, "icl_embed_gather_interleave: NULL pointer");
  ICL_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0, "icl_embed_gather_interleave: rows>0 and H%%8==0 required");
  ICL_CHECK_ARG(((uintptr_t)table & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)speech & 15) == 0,
                "icl_embed_gather_interleave: misaligned pointer");
  hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, src_idx,
                     (const unsigned short*)table, speech, out, H);
  ICL_CHECK_LAUNCH("icl_embed_gather_interleave");
  return ICL_OK;
}

extern "C" int icl_argmax_eos(const float* logits, int64_t ldl, int32_t B, int32_t V, int32_t eos_id, int32_t eos_id2,
                              int32_t pad_id, int32_t* finished, int32_t* out_tokens, int32_t out_stride,
                              int32_t step, int32_t* next_ids, void* stream) {
  ICL_CHECK_ARG(logits && finished && out_tokens && next_ids, "icl_argmax_eos: NULL pointer");
  ICL_CHECK_ARG(B > 0 && V > 0 && ldl >= V, "icl_argmax_eos: bad sizes");
  ICL_CHECK_ARG(step >= 0 && step < out_stride, "icl_argmax_eos: step=%d outside out_stride=%d", step, out_stride);
  hipLaunchKernelGGL(argmax_eos_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, eos_id,
                     eos_id2, pad_id, finished, out_tokens, out_stride, step, next_ids);
  ICL_CHECK_LAUNCH("icl_argmax_eos");
  return ICL_OK;
}

extern "C" int icl_axpby_cast(const void* in, int64_t ldi, int32_t in_dtype, const void* add, int64_t lda_,
                              int32_t add_dtype, float alpha, void* out, int64_t ldo, int32_t out_dtype,
                              int32_t M, int32_t N, void* stream) {
  ICL_CHECK_ARG(in && out && M > 0 && N > 0, "icl_axpby_cast: bad arguments");
  const int64_t total = (int64_t)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(axpby_cast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ldi, in_dtype,
                     add, lda_, add_dtype, alpha, out, ldo, out_dtype, M, N);
  ICL_CHECK_LAUNCH("icl_axpby_cast");
  return ICL_OK;
}

extern "C" int icl_lora_down_bf16(void* X, int64_t ldx, int32_t K0, const void* A, int64_t lda_,
                                  int32_t r_total, float scale, int32_t M, void* stream) {
  ICL_CHECK_ARG(X && A && M > 0, "icl_lora_down_bf16: bad arguments");
  ICL_CHECK_ARG(K0 % 8 == 0 && ldx % 8 == 0 && lda_ % 8 == 0 && ldx >= K0 + r_total && r_total > 0 && r_total <= 64,
                "icl_lora_down_bf16: bad shape (K0=%d ldx=%lld r=%d)", K0, (long long)ldx, r_total);
  ICL_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)A & 15) == 0, "icl_lora_down_bf16: misaligned pointer");
  hipLaunchKernelGGL(lora_down_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (unsigned short*)X, ldx,
                     K0, (const unsigned short*)A, lda_, r_total, scale);
  ICL_CHECK_LAUNCH("icl_lora_down_bf16");
  return ICL_OK;
}

extern "C" int icl_beats_gate(const void* qkv, int64_t ld, const float* grep_w, const float* grep_b,
                              const float* grep_a, float* gate, int32_t M, int32_t n_heads, void* stream) {
  ICL_CHECK_ARG(qkv && grep_w && grep_b && grep_a && gate && M > 0 && n_heads > 0, "icl_beats_gate: bad arguments");
  ICL_CHECK_ARG(ld % 8 == 0 && ((uintptr_t)qkv & 15) == 0, "icl_beats_gate: qkv misaligned");
  const int64_t total = (int64_t)M * n_heads;
  hipLaunchKernelGGL(beats_gate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const unsigned short*)qkv, ld, grep_w, grep_b, grep_a, gate, M, n_heads);
  ICL_CHECK_LAUNCH("icl_beats_gate");
  return ICL_OK;
}

extern "C" int icl_qformer_window_xattn(const void* q, int64_t ldq, const void* kv, int64_t ldkv, int64_t v_off,
                                        void* out, int64_t ldo, int32_t n_audio, int32_t win_per_audio,
                                        int32_t win, int32_t rows_per_audio, int32_t n_heads, float scale,
                                        void* stream) {
  ICL_CHECK_ARG(q && kv && out, "icl_qformer_window_xattn: NULL pointer");
  ICL_CHECK_ARG(n_audio > 0 && win_per_audio > 0 && n_heads > 0, "icl_qformer_window_xattn: bad sizes");
  ICL_CHECK_ARG(win >= 1 && win <= 64, "icl_qformer_window_xattn: win=%d must be in [1,64]", win);
  ICL_CHECK_ARG((int64_t)win_per_audio * win <= rows_per_audio, "icl_qformer_window_xattn: windows exceed rows_per_audio");
  ICL_CHECK_ARG(ldq % 8 == 0 && ldkv % 8 == 0 && v_off % 8 == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)kv & 15) == 0,
                "icl_qformer_window_xattn: misaligned operands");
  const int n_pairs = n_audio * win_per_audio * n_heads;
  hipLaunchKernelGGL(qformer_xattn_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)q, ldq, (const unsigned short*)kv, ldkv, v_off,
                     (unsigned short*)out, ldo, win_per_audio, win, rows_per_audio, n_heads, scale, n_pairs);
  ICL_CHECK_LAUNCH("icl_qformer_window_xattn");
  return ICL_OK;
}

// ---------------------------------------------------------------------------------------------
// K12: shift-by-one causal-LM cross entropy (HF LlamaForCausalLM loss, ignore_index = -100).
// Row r of `logits` predicts labels[r] (the caller passes labels already shifted); one block per row.
namespace {
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* logits, int64_t ldl, const int* labels, int V,
                                                       float* row_loss) {
  __shared__ float red[4];
  const int64_t r = blockIdx.x;
  const int y = labels[r];
  if (y < 0 || y >= V) {
    if (threadIdx.x == 0) row_loss[r] = 0.f;
    return;
  }
  const float* row = logits + r * ldl;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, row[v]);
  mx = wave_reduce_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += __expf(row[v] - mx);
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) row_loss[r] = logf(red[0] + red[1] + red[2] + red[3]) + mx - row[y];
}
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* row_loss, const int* labels, int V, int M, float* out) {
  __shared__ float rs[4];
  __shared__ float rc[4];
  float s = 0.f, c = 0.f;
  for (int r = threadIdx.x; r < M; r += blockDim.x) {
    const int y = labels[r];
    if (y >= 0 && y < V) {
      s += row_loss[r];
      c += 1.f;
    }
  }
  s = wave_reduce_sum(s);
  c = wave_reduce_sum(c);
  if ((threadIdx.x & 63) == 0) {
    rs[threadIdx.x >> 6] = s;
    rc[threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S = rs[0] + rs[1] + rs[2] + rs[3], C = rc[0] + rc[1] + rc[2] + rc[3];
    out[0] = C > 0.f ? S / C : nanf("");
  }
}
}  // namespace

extern "C" int icl_cross_entropy(const float* logits, int64_t ldl, const int32_t* labels, int32_t M, int32_t V,
                                 float* row_loss, float* mean_loss, void* stream) {
  ICL_CHECK_ARG(logits && labels && row_loss && mean_loss && M > 0 && V > 0 && ldl >= V, "icl_cross_entropy: bad arguments");
  hipLaunchKernelGGL(ce_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ldl, labels, V, row_loss);
  ICL_CHECK_LAUNCH("icl_cross_entropy(rows)");
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)row_loss, labels, V, M, mean_loss);
  ICL_CHECK_LAUNCH("icl_cross_entropy(mean)");
  return ICL_OK;
}
