// beam.hip — beam search bookkeeping on the device (gfx950): one step of HF's static-shaped beam search per launch, and the
// K/V-cache span copy that stands in for HF's cache.reorder_cache.
//
// Reference call site: models/custom_salmon.py:704-715 forwards num_beams / length_penalty to HF generate(inputs_embeds=...),
// models/multi_task_model.py:142 sets them per task.  The algorithm restated here is transformers/generation/utils.py
// `_beam_search` (early_stopping=False, one or two EOS ids, do_sample=False): per row, the 2K (3K with two EOS ids) best continuations of the K running
// beams; the K best of them that do not stop run on; those among the first K that stop (EOS or the length limit) compete for
// the K finished slots at sum-of-log-probs / len**length_penalty; a row stops taking finished hypotheses once its best
// running score / cur_len**length_penalty cannot beat its worst finished one.  The prompt length is 0 for the scorer
// (inputs_embeds only).  All state lives in HBM and every step is enqueued without a host round trip: a row that is done
// keeps stepping with its finished slots masked, which changes nothing (HF's loop stops early instead).
//
// One workgroup per batch row.  The candidate scan is 2K rounds of a block-wide lexicographic arg-max over the K*V
// accumulated log-probabilities (recomputed from the logits each round: K*V*4 B <= 1 MB per row stays in L2); ties go to
// the lower flat index (torch.topk leaves them unspecified).  Not a hot kernel: beam search multiplies the decode batch by K,
// and that cost is in the GEMMs and the attention.
#include "common.h"

namespace {

constexpr int BEAM_MAX = 8;      // num_beams
constexpr int BEAM_TMAX = 64;    // max_new_tokens
constexpr float BEAM_NEG = -1.0e9f;

__device__ __forceinline__ bool lex_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }
__device__ __forceinline__ float nan_to_neg_inf(float x) { return x == x ? x : -INFINITY; }

__global__ __launch_bounds__(256) void beam_step_kernel(
    const float* __restrict__ logits, int64_t ldl, int rows_per_batch, int V, int K, int T, int step, int eos_id, int eos_id2, float rep_pen,
    float lenpen_next,   // (step + 1) ** length_penalty: divides a hypothesis finishing now AND the best running score after it
    float* __restrict__ run_score, int* __restrict__ run_seq, float* __restrict__ fin_score, int* __restrict__ fin_seq,
    int* __restrict__ fin_len, int* __restrict__ fin_flag, int* __restrict__ unsat, int* __restrict__ next_ids,
    int* __restrict__ parent) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float s_m[BEAM_MAX], s_ls[BEAM_MAX], s_rs[BEAM_MAX];
  __shared__ float s_redv[4];
  __shared__ int s_redi[4];
  __shared__ float s_cv[3 * BEAM_MAX];
  __shared__ int s_ci[3 * BEAM_MAX];
  __shared__ int s_old_run[BEAM_MAX * BEAM_TMAX], s_old_fin[BEAM_MAX * BEAM_TMAX];
  __shared__ int s_run_src[BEAM_MAX];                 // candidate index feeding running beam i
  __shared__ int s_fin_src[BEAM_MAX];                 // merged index (0..K-1: old finished slot, K..: candidate) feeding slot i
  __shared__ float s_fin_newscore[BEAM_MAX];
  __shared__ int s_fin_newflag[BEAM_MAX], s_fin_newlen[BEAM_MAX];

  const int NC = (eos_id2 >= 0 ? 3 : 2) * K;          // HF: max(2, 1 + number of EOS ids) * num_beams continuations are kept
  const float* lbase = logits + (int64_t)b * rows_per_batch * ldl;      // beam k's logits: row k (one shared row at step 0)
  const int64_t lstep = rows_per_batch == 1 ? 0 : ldl;

  // ---- log-softmax statistics per beam: max, log(sum exp(x - max)) ------------------------------------------------
  for (int k = 0; k < K; ++k) {
    const float* x = lbase + k * lstep;
    float m = -INFINITY;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, nan_to_neg_inf(x[v]));     // fmaxf drops a NaN operand anyway; explicit
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) s_redv[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_redv[0], s_redv[1]), fmaxf(s_redv[2], s_redv[3]));
    if (!(m > -INFINITY)) m = 0.f;                    // a row of NaN / -inf only: every log-probability becomes -inf below
    __syncthreads();
    float s = 0.f;
    for (int v = tid; v < V; v += 256) s += expf(nan_to_neg_inf(x[v]) - m);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) s_redv[wave] = s;
    __syncthreads();
    if (tid == 0) {
      s_m[k] = m;
      s_ls[k] = logf((s_redv[0] + s_redv[1]) + (s_redv[2] + s_redv[3]));
      s_rs[k] = run_score[b * K + k];
    }
    __syncthreads();
  }
  for (int i = tid; i < K * T; i += 256) {
    s_old_run[i] = run_seq[(int64_t)b * K * T + i];
    s_old_fin[i] = fin_seq[(int64_t)b * K * T + i];
  }
  __syncthreads();
  const bool penalised = rep_pen != 1.0f && step > 0;      // HF's RepetitionPenaltyLogitsProcessor, applied to the log-probabilities

  // ---- the NC best continuations, best first -------------------------------------------------------------------------
  float pv = INFINITY;
  int pi = -1;
  for (int r = 0; r < NC; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = 0; k < K; ++k) {
      const float m = s_m[k], ls = s_ls[k], rs = s_rs[k];
      const float* x = lbase + k * lstep;
      for (int v = tid; v < V; v += 256) {
        const float xv = nan_to_neg_inf(x[v]);             // a NaN logit is a token that cannot be chosen, as in argmax_eos_kernel
        float lp = xv > -INFINITY ? (xv - m) - ls : -INFINITY;
        if (penalised) {                                   // tokens of beam k's own sequence: x < 0 ? x * penalty : x / penalty
          bool seen = false;
          for (int t = 0; t < step; ++t) seen = seen || s_old_run[k * T + t] == v;
          if (seen) lp = lp < 0.f ? lp * rep_pen : lp / rep_pen;
        }
        const float a = lp + rs;
        const int idx = k * V + v;
        if ((a < pv || (a == pv && idx > pi)) && lex_better(a, idx, bv, bi)) {
          bv = a;
          bi = idx;
        }
      }
    }
    for (int o = 32; o; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (lex_better(ov, oi, bv, bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_redv[wave] = bv;
      s_redi[wave] = bi;
    }
    __syncthreads();
    bv = s_redv[0];
    bi = s_redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (lex_better(s_redv[w], s_redi[w], bv, bi)) {
        bv = s_redv[w];
        bi = s_redi[w];
      }
    if (bi == 0x7fffffff) {       // nothing comparable was left (cannot happen once NaNs are mapped; kept so that parent /
      bv = -INFINITY;             // next_ids stay inside [0, K) x [0, V) whatever the logits hold)
      bi = r < V ? r : 0;
    }
    if (tid == 0) {
      s_cv[r] = bv;
      s_ci[r] = bi;
    }
    pv = bv;
    pi = bi;
    __syncthreads();
  }

  // ---- bookkeeping (a few dozen scalar operations) -----------------------------------------------------------------------
  if (tid == 0) {
    const bool last = step + 1 >= T;
    const bool row_open = unsat[b] != 0;
    bool stops[3 * BEAM_MAX];
    float runv[3 * BEAM_MAX];
    for (int j = 0; j < NC; ++j) {
      const int tk = s_ci[j] % V;
      stops[j] = last || tk == eos_id || tk == eos_id2;
      runv[j] = s_cv[j] + (stops[j] ? BEAM_NEG : -0.0f);
    }
    // next running beams: the K best by (score - 1e9 * stops), earlier candidate first on ties
    bool used[3 * BEAM_MAX] = {};
    for (int i = 0; i < K; ++i) {
      int best = -1;
      for (int j = 0; j < NC; ++j)
        if (!used[j] && (best < 0 || runv[j] > runv[best])) best = j;
      used[best] = true;
      s_run_src[i] = best;
    }
    // finished slots: old slots first, then the candidates; only the first K candidates may finish, only while the row is open
    float ms[4 * BEAM_MAX];
    int mflag[4 * BEAM_MAX];
    for (int k = 0; k < K; ++k) {
      ms[k] = fin_score[b * K + k];
      mflag[k] = fin_flag[b * K + k];
    }
    for (int j = 0; j < NC; ++j) {
      const bool just = stops[j] && j < K;
      float s = s_cv[j] / lenpen_next;
      s += row_open ? -0.0f : BEAM_NEG;
      s += just ? -0.0f : BEAM_NEG;
      ms[K + j] = s;
      mflag[K + j] = just ? 1 : 0;
    }
    bool mused[4 * BEAM_MAX] = {};
    float worst = INFINITY;
    for (int i = 0; i < K; ++i) {
      int best = -1;
      for (int j = 0; j < K + NC; ++j)
        if (!mused[j] && (best < 0 || ms[j] > ms[best])) best = j;
      mused[best] = true;
      s_fin_src[i] = best;
      s_fin_newscore[i] = ms[best];
      s_fin_newflag[i] = mflag[best];
      s_fin_newlen[i] = best < K ? fin_len[b * K + best] : step + 1;
      worst = fminf(worst, ms[best]);
    }
    // can the best running beam still beat the worst finished hypothesis?
    const float best_run = runv[s_run_src[0]] / lenpen_next;
    bool any = false;
    for (int i = 0; i < K; ++i) any = any || best_run > (s_fin_newflag[i] ? worst : BEAM_NEG);
    unsat[b] = (row_open && any) ? 1 : 0;
    for (int i = 0; i < K; ++i) {
      const int j = s_run_src[i];
      run_score[b * K + i] = runv[j];
      next_ids[b * K + i] = s_ci[j] % V;
      parent[b * K + i] = b * K + s_ci[j] / V;
      fin_score[b * K + i] = s_fin_newscore[i];
      fin_flag[b * K + i] = s_fin_newflag[i];
      fin_len[b * K + i] = s_fin_newlen[i];
    }
  }
  __syncthreads();
  // ---- sequences: running beam i = old running beam parent(i) + its token; finished slot i from its merged source --------
  for (int e = tid; e < K * T; e += 256) {
    const int i = e / T, t = e % T;
    const int j = s_run_src[i];
    const int par = s_ci[j] / V, tok = s_ci[j] % V;
    run_seq[(int64_t)b * K * T + e] = t < step ? s_old_run[par * T + t] : (t == step ? tok : s_old_run[par * T + t]);
    const int src = s_fin_src[i];
    int val;
    if (src < K) {
      val = s_old_fin[src * T + t];
    } else {
      const int cpar = s_ci[src - K] / V, ctok = s_ci[src - K] % V;
      val = t < step ? s_old_run[cpar * T + t] : (t == step ? ctok : s_old_run[cpar * T + t]);
    }
    fin_seq[(int64_t)b * K * T + e] = val;
  }
}

// Copy a span of positions of one (sequence, head) stream to another, for every layer and head: 16 B per thread.
__global__ __launch_bounds__(256) void kv_copy_spans_kernel(
    const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int64_t s_layer, int64_t s_seq, int64_t s_head,
    int64_t d_layer, int64_t d_seq, int64_t d_head, const int* __restrict__ src_seq, const int* __restrict__ src_t0,
    const int* __restrict__ dst_seq, const int* __restrict__ dst_t0, const int* __restrict__ n_t, int n_fixed, int H, int D,
    int src_n_seqs, int dst_n_seqs, int src_len, int dst_len) {
  const int r = blockIdx.x / H, h = blockIdx.x % H, l = blockIdx.y;
  // sequence ids and span starts come from device state (the beam step's parent array): they are clamped to the arrays the
  // caller described, so a corrupted id reads / writes a wrong but EXISTING span — never memory outside the allocation
  const int ss = min(max(src_seq ? src_seq[r] : r, 0), src_n_seqs - 1);
  const int ds = min(max(dst_seq ? dst_seq[r] : r, 0), dst_n_seqs - 1);
  const int st0 = min(max(src_t0 ? src_t0[r] : 0, 0), src_len);
  const int dt0 = min(max(dst_t0 ? dst_t0[r] : 0, 0), dst_len);
  const int n = min(min(max(n_t ? n_t[r] : n_fixed, 0), src_len - st0), dst_len - dt0);
  const int64_t so = (int64_t)l * s_layer + (int64_t)ss * s_seq + (int64_t)h * s_head + (int64_t)st0 * D;
  const int64_t dof = (int64_t)l * d_layer + (int64_t)ds * d_seq + (int64_t)h * d_head + (int64_t)dt0 * D;
  const u32x4* sp = (const u32x4*)(src + so);
  u32x4* dp = (u32x4*)(dst + dof);
  const int chunks = n * D / 8;
  for (int i = threadIdx.x; i < chunks; i += 256) dp[i] = sp[i];
}

}  // namespace

extern "C" int icl_beam_step(const float* logits, int64_t ldl, int32_t rows_per_batch, int32_t B, int32_t V, int32_t num_beams,
                             int32_t max_new_tokens, int32_t step, int32_t eos_id, int32_t eos_id2, float length_penalty,
                             float repetition_penalty, float* run_score,
                             int32_t* run_seq, float* fin_score, int32_t* fin_seq, int32_t* fin_len, int32_t* fin_flag,
                             int32_t* unsat, int32_t* next_ids, int32_t* parent, void* stream) {
  ICL_CHECK_ARG(logits && run_score && run_seq && fin_score && fin_seq && fin_len && fin_flag && unsat && next_ids && parent,
                "icl_beam_step: NULL pointer");
  ICL_CHECK_ARG(B > 0 && V > 0 && ldl >= V, "icl_beam_step: bad sizes");
  ICL_CHECK_ARG(num_beams >= 1 && num_beams <= BEAM_MAX, "icl_beam_step: num_beams=%d must be in [1,%d]", num_beams, BEAM_MAX);
  ICL_CHECK_ARG(V >= (eos_id2 >= 0 ? 3 : 2) * num_beams, "icl_beam_step: V=%d < %d * num_beams", V, eos_id2 >= 0 ? 3 : 2);
  ICL_CHECK_ARG(eos_id2 < 0 || eos_id >= 0, "icl_beam_step: eos_id2 without eos_id");
  ICL_CHECK_ARG(repetition_penalty > 0.0f, "icl_beam_step: repetition_penalty must be > 0");
  ICL_CHECK_ARG(max_new_tokens >= 1 && max_new_tokens <= BEAM_TMAX, "icl_beam_step: max_new_tokens=%d must be in [1,%d]",
                max_new_tokens, BEAM_TMAX);
  ICL_CHECK_ARG(step >= 0 && step < max_new_tokens, "icl_beam_step: step=%d outside [0,%d)", step, max_new_tokens);
  ICL_CHECK_ARG(rows_per_batch == 1 || rows_per_batch == num_beams, "icl_beam_step: rows_per_batch must be 1 or num_beams");
  ICL_CHECK_ARG((int64_t)num_beams * V < 0x7fffffffLL, "icl_beam_step: num_beams * V overflows the flat index");
  // python: (cur_len + 1 - prompt_len) ** length_penalty in double, then the f32 tensor is divided by it
  const float lenpen = (float)pow((double)(step + 1), (double)length_penalty);
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ldl, rows_per_batch, V, num_beams,
                     max_new_tokens, step, eos_id, eos_id2, repetition_penalty, lenpen, run_score, run_seq, fin_score, fin_seq, fin_len, fin_flag, unsat,
                     next_ids, parent);
  ICL_CHECK_LAUNCH("icl_beam_step");
  return ICL_OK;
}

extern "C" int icl_kv_copy_spans_bf16(const void* src, void* dst, int64_t src_layer_stride, int64_t src_seq_stride,
                                      int64_t src_head_stride, int64_t dst_layer_stride, int64_t dst_seq_stride,
                                      int64_t dst_head_stride, const int32_t* src_seq, const int32_t* src_t0,
                                      const int32_t* dst_seq, const int32_t* dst_t0, const int32_t* n_t, int32_t n_fixed,
                                      int32_t n_rows, int32_t n_layers, int32_t n_heads, int32_t head_dim,
                                      int32_t src_n_seqs, int32_t dst_n_seqs, int32_t src_len, int32_t dst_len, void* stream) {
  ICL_CHECK_ARG(src && dst, "icl_kv_copy_spans_bf16: NULL pointer");
  ICL_CHECK_ARG(src_n_seqs > 0 && dst_n_seqs > 0 && src_len > 0 && dst_len > 0,
                "icl_kv_copy_spans_bf16: src / dst extents (sequences, positions) must be > 0");
  ICL_CHECK_ARG((src_seq || n_rows <= src_n_seqs) && (dst_seq || n_rows <= dst_n_seqs),
                "icl_kv_copy_spans_bf16: n_rows=%d exceeds the %d / %d sequences of src / dst", n_rows, src_n_seqs, dst_n_seqs);
  ICL_CHECK_ARG(n_rows > 0 && n_layers > 0 && n_heads > 0 && head_dim > 0 && head_dim % 8 == 0,
                "icl_kv_copy_spans_bf16: bad sizes (head_dim must be a multiple of 8)");
  ICL_CHECK_ARG(n_t || n_fixed >= 0, "icl_kv_copy_spans_bf16: n_fixed < 0");
  ICL_CHECK_ARG(((uintptr_t)src | (uintptr_t)dst) % 16 == 0 &&
                    (src_layer_stride | src_seq_stride | src_head_stride | dst_layer_stride | dst_seq_stride | dst_head_stride) % 8 == 0,
                "icl_kv_copy_spans_bf16: pointers and strides must be 16-byte aligned");
  ICL_CHECK_ARG((int64_t)n_rows * n_heads < 0x7fffffffLL && n_layers <= 65535, "icl_kv_copy_spans_bf16: grid too large");
  if (!n_t && n_fixed == 0) return ICL_OK;
  hipLaunchKernelGGL(kv_copy_spans_kernel, dim3(n_rows * n_heads, n_layers), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, (unsigned short*)dst, src_layer_stride, src_seq_stride, src_head_stride,
                     dst_layer_stride, dst_seq_stride, dst_head_stride, src_seq, src_t0, dst_seq, dst_t0, n_t, n_fixed, n_heads,
                     head_dim, src_n_seqs, dst_n_seqs, src_len, dst_len);
  ICL_CHECK_LAUNCH("icl_kv_copy_spans_bf16");
  return ICL_OK;
}
