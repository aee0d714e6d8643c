// sampling.hip — the sampled variant of the decode tail (reference: models/custom_salmon.py:705-721 passes
// do_sample / temperature / top_p / repetition_penalty to HF generate, whose logits pipeline is
// RepetitionPenalty -> Temperature -> TopK (generation-config default 50) -> TopP -> softmax -> multinomial).
//
// One block per sequence:
//   1. work[v] = logits[v] / temperature, then work[t] = f(logits[t]) / temperature for every token t generated so far
//      with f(x) = x < 0 ? x * penalty : x / penalty (once per distinct token, as HF's gather/scatter does);
//   2. exact k-th largest value by a 4-pass 8-bit radix select over order-preserving integer keys (LDS histograms);
//   3. every token >= that value (ties kept, as HF's `scores < kth` mask does) goes to an LDS candidate list, which is
//      bitonic-sorted by (value descending, token id ascending);
//   4. softmax over the candidates; nucleus cut: candidate j is dropped when the probability mass of candidates j.. (the
//      ascending cumulative sum HF computes) is <= 1 - top_p, the largest one always stays;
//   5. inverse-CDF draw over the kept candidates in that order with the caller's uniform u in [0,1);
//   6. the same EOS / pad bookkeeping as the greedy tail (argmax_eos_kernel).
// top_k == V is HF's "top-k switched off" (top_k = 0 / None in a generation config; TopKLogitsWarper clamps to V): the
// probabilities are then normalised over the WHOLE row (one more pass), the candidate list holds the SAMPLE_CAP most likely
// tokens and the nucleus cut works on "1 - mass before candidate j"; a nucleus wider than SAMPLE_CAP tokens is truncated to
// them (a row that flat carries no label information; documented in include/icl_hip.h).
// HBM-bound on 5 reads of one f32 logits row per sequence (L2 resident after the first); algorithmic bytes = 4*V per row.
#include "common.h"

#define SAMPLE_CAP 1024

__device__ __forceinline__ unsigned int f32_order_key(float x) {
  if (x != x) return 0u;                       // NaN sorts below everything
  const unsigned int u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void sample_eos_kernel(
    const float* logits, int64_t ldl, int V, float* work, int64_t ldw, const int* prev_tokens, int prev_stride,
    int n_prev, float penalty, float temperature, int top_k, float top_p, const float* uniforms, int eos_id, int eos_id2,
    int pad_id, int* finished, int* out_tokens, int out_stride, int step, int* next_ids, int* dbg_ids,
    float* dbg_probs, int* dbg_count, int dbg_cap) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_remaining, s_count;
  __shared__ unsigned int ckey[SAMPLE_CAP];
  __shared__ int cidx[SAMPLE_CAP];
  __shared__ float cprob[SAMPLE_CAP];
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (int64_t)b * ldl;
  float* w = work + (int64_t)b * ldw;

  // 1. processed scores
  for (int v = tid; v < V; v += 256) w[v] = row[v] / temperature;
  __syncthreads();
  if (penalty != 1.0f)
    for (int i = tid; i < n_prev; i += 256) {
      const int t = prev_tokens[(int64_t)b * prev_stride + i];
      if (t >= 0 && t < V) {
        const float x = row[t];
        w[t] = (x < 0.0f ? x * penalty : x / penalty) / temperature;   // duplicates write the same value
      }
    }
  __syncthreads();

  // 2. radix select: key of the k-th largest score
  const bool full = top_k >= V;                   // top-k off: candidates = the SAMPLE_CAP largest, full-row denominator
  if (tid == 0) {
    s_prefix = 0u;
    s_remaining = (unsigned int)(full ? min(V, SAMPLE_CAP) : top_k);
    s_count = 0u;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0u;
    __syncthreads();
    const unsigned int prefix = s_prefix;
    const unsigned int mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int v = tid; v < V; v += 256) {
      const unsigned int k = f32_order_key(w[v]);
      if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int need = s_remaining, bin = 255u;
      for (;; --bin) {
        if (hist[bin] >= need || bin == 0u) break;
        need -= hist[bin];
      }
      s_prefix = prefix | (bin << shift);
      s_remaining = need;
    }
    __syncthreads();
  }
  const unsigned int kth = s_prefix;

  // 3. candidates (all scores >= the k-th largest), sorted by (score desc, token asc)
  for (int v = tid; v < V; v += 256) {
    const unsigned int k = f32_order_key(w[v]);
    if (k >= kth) {
      const unsigned int slot = atomicAdd(&s_count, 1u);
      if (slot < SAMPLE_CAP) {
        ckey[slot] = k;
        cidx[slot] = v;
      }
    }
  }
  __syncthreads();
  const int n_cand = (int)min(s_count, (unsigned int)SAMPLE_CAP);
  int n_sort = 1;
  while (n_sort < n_cand) n_sort <<= 1;
  for (int i = n_cand + tid; i < n_sort; i += 256) {
    ckey[i] = 0u;
    cidx[i] = 0x7fffffff;
  }
  __syncthreads();
  for (int size = 2; size <= n_sort; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < n_sort; i += 256) {
        const int j = i ^ stride;
        if (j > i) {
          const bool up = (i & size) == 0;             // "up" = this run ends sorted in the wanted (descending) order
          const unsigned int ki = ckey[i], kj = ckey[j];
          const int ii = cidx[i], ij = cidx[j];
          const bool i_first = ki > kj || (ki == kj && ii < ij);
          if (i_first != up) {
            ckey[i] = kj; ckey[j] = ki;
            cidx[i] = ij; cidx[j] = ii;
          }
        }
      }
      __syncthreads();
    }

  // 4. softmax over the candidates
  const float top = w[cidx[0]];
  float part = 0.0f;
  for (int i = tid; i < n_cand; i += 256) {
    const float e = expf(w[cidx[i]] - top);
    cprob[i] = e;
    part += e;
  }
  if (full) {                                     // every token of the row is in the distribution, listed or not
    part = 0.0f;
    for (int v = tid; v < V; v += 256) {
      const float x = w[v];
      part += x == x ? expf(x - top) : 0.0f;
    }
  }
  part = wave_reduce_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  const float denom = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = tid; i < n_cand; i += 256) cprob[i] = cprob[i] / denom;
  __syncthreads();

  // 5. nucleus cut + inverse-CDF draw (one thread: <= 1024 candidates, typically 50)
  if (tid == 0) {
    int keep = n_cand;
    if (top_p < 1.0f && full) {                   // mass of candidates j.. (+ the unlisted tail) = 1 - mass before j
      float before = cprob[0];
      keep = 1;
      for (int j = 1; j < n_cand; ++j) {
        if (1.0f - before <= 1.0f - top_p) break;
        before += cprob[j];
        keep = j + 1;
      }
    } else if (top_p < 1.0f) {
      float tail = 0.0f;
      keep = 1;
      for (int j = n_cand - 1; j >= 1; --j) {
        tail += cprob[j];
        if (tail > 1.0f - top_p) {
          keep = j + 1;
          break;
        }
      }
    }
    float total = 0.0f;
    for (int j = 0; j < keep; ++j) total += cprob[j];
    const float target = uniforms[b] * total;
    float acc = 0.0f;
    int pick = keep - 1;
    for (int j = 0; j < keep; ++j) {
      acc += cprob[j];
      if (acc > target) {
        pick = j;
        break;
      }
    }
    int fin = finished[b];
    const int tok = fin ? pad_id : cidx[pick];
    if (tok == eos_id || tok == eos_id2) fin = 1;
    finished[b] = fin;
    out_tokens[(int64_t)b * out_stride + step] = tok;
    next_ids[b] = tok;
    if (dbg_count) {
      dbg_count[b] = keep;
      for (int j = 0; j < keep && j < dbg_cap; ++j) {
        dbg_ids[(int64_t)b * dbg_cap + j] = cidx[j];
        dbg_probs[(int64_t)b * dbg_cap + j] = cprob[j] / total;
      }
    }
  }
}

extern "C" int icl_sample_eos(const float* logits, int64_t ldl, int32_t B, int32_t V, float* work, int64_t ldw,
                              const int32_t* prev_tokens, int32_t prev_stride, int32_t n_prev,
                              float repetition_penalty, float temperature, int32_t top_k, float top_p,
                              const float* uniforms, int32_t eos_id, int32_t eos_id2, int32_t pad_id, int32_t* finished,
                              int32_t* out_tokens, int32_t out_stride, int32_t step, int32_t* next_ids,
                              int32_t* dbg_ids, float* dbg_probs, int32_t* dbg_count, int32_t dbg_cap, void* stream) {
  ICL_CHECK_ARG(logits && work && uniforms && finished && out_tokens && next_ids, "icl_sample_eos: NULL pointer");
  ICL_CHECK_ARG(B > 0 && V > 0 && ldl >= V && ldw >= V, "icl_sample_eos: bad sizes");
  ICL_CHECK_ARG(step >= 0 && step < out_stride, "icl_sample_eos: step=%d outside out_stride=%d", step, out_stride);
  ICL_CHECK_ARG(temperature > 0.0f, "icl_sample_eos: temperature must be > 0");
  ICL_CHECK_ARG(top_p > 0.0f && top_p <= 1.0f, "icl_sample_eos: top_p must be in (0,1]");
  ICL_CHECK_ARG(top_k >= 1 && (top_k <= SAMPLE_CAP || top_k == V) && top_k <= V,
                "icl_sample_eos: top_k=%d must be in [1,%d] or equal to V=%d (top-k off)", top_k, SAMPLE_CAP, V);
  ICL_CHECK_ARG(repetition_penalty > 0.0f, "icl_sample_eos: repetition_penalty must be > 0");
  ICL_CHECK_ARG(n_prev == 0 || (prev_tokens && n_prev > 0 && n_prev <= prev_stride), "icl_sample_eos: bad prev_tokens");
  ICL_CHECK_ARG(!dbg_count || (dbg_ids && dbg_probs && dbg_cap > 0), "icl_sample_eos: incomplete debug outputs");
  hipLaunchKernelGGL(sample_eos_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, work, ldw,
                     prev_tokens, prev_stride, n_prev, repetition_penalty, temperature, top_k, top_p, uniforms,
                     eos_id, eos_id2, pad_id, finished, out_tokens, out_stride, step, next_ids, dbg_ids, dbg_probs, dbg_count,
                     dbg_cap);
  ICL_CHECK_LAUNCH("icl_sample_eos");
  return ICL_OK;
}
