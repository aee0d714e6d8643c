/*
 * icl_hip.h — C-ABI of libicl_hip.so, the MI355X (gfx950 / CDNA4) kernel library behind the
 * ICL forward/generate hot path of iiscleap/ICL-speech-text-LLM.
 *
 * The reference has NO FFI of its own: every kernel it runs is reached through PyTorch /
 * transformers / the external SALMONN package.  Each entry point
 * below therefore cites the reference call site (file:line under /root/reference) whose
 * arithmetic it replaces, plus the implicit-op row of SURVEY.md §2.3 (K1..K14).
 *
 * Conventions:
 *   - plain pointers and sizes only; no torch types; every pointer is a DEVICE pointer unless
 *     its name ends in _host;
 *   - no allocation or ownership inside the library: the caller (PyTorch-ROCm on the host
 *     side) allocates inputs, outputs and workspaces;
 *   - every call is asynchronous w.r.t. the host and ordered on `stream` (a hipStream_t passed
 *     as void*); nothing here synchronises, allocates or memcpy's, so every call may be
 *     captured into a hipGraph;
 *   - return value: 0 on success, a negative ICL_E* code otherwise; icl_last_error() returns
 *     a thread-local, human-readable message for the most recent failure on this thread.
 *     A failure never aborts the process and never leaves a kernel running: arguments are
 *     validated on the host before any launch (reference behaviour to preserve: the per-batch
 *     try/except in inference/inference.py:370-373 must be able to continue).
 *   - "bf16" tensors are raw uint16 bit patterns of bfloat16; "f32" is IEEE binary32.
 *   - all matrices are row-major with an explicit leading dimension (in ELEMENTS).
 */
#ifndef ICL_HIP_H
#define ICL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICL_ABI_VERSION 5

/* error codes */
#define ICL_OK 0
#define ICL_EINVAL (-1)  /* bad argument / unsupported shape (message in icl_last_error) */
#define ICL_ELAUNCH (-2) /* hipLaunch / runtime error (message carries hipGetErrorString) */

/* element types for outputs / residuals */
#define ICL_BF16 0
#define ICL_F32 1

/* ---- library ------------------------------------------------------------------------- */
int icl_abi_version(void);
const char* icl_last_error(void);
/* Number of compute units of the current device (used by callers to size split-K). */
int icl_device_cu_count(void);

/* ---- K2/K3/K5/K7/K8/K10/K13/K14: bf16 MFMA GEMM ----------------------------------------
 * C[b][m][n] = epilogue( sum_k A[b][m][k] * W[n][k] )       (W is the PyTorch nn.Linear
 * layout [N,K], K contiguous), fp32 accumulation on v_mfma_f32_16x16x32_bf16.
 * Replaces every nn.Linear / Conv1d-as-GEMM of the path: HF WhisperEncoder / LlamaForCausalLM
 * reached from models/custom_salmon.py:550-554 (encode_speech) and :630-636 / :704-720
 * (llama_model forward / generate).
 *
 * epilogue (applied in this order):  v = acc; v += bias[n] (ICL_EPI_BIAS);
 *   v = gelu_erf(v) (ICL_EPI_GELU);  v = silu(v_gate)*v_up (ICL_EPI_SWIGLU, see below);
 *   v += R[b][m][n] (ICL_EPI_RESIDUAL, R is f32 or bf16 per res_dtype); store as out_dtype.
 * ICL_EPI_SWIGLU: W holds gate/up rows interleaved in blocks of 16 ([g0..g15,u0..u15,g16..]),
 *   N counts the interleaved rows; the output has N/2 columns (ldc refers to that matrix).
 * batch: independent problems on grid.z with element strides (stride 0 = broadcast; W has
 *   no batch stride — the weight is shared).
 * split_k > 1 (only with batch == 1): partial sums go to `workspace` (f32, at least
 *   split_k*M*N elements) and a second kernel reduces them and applies the epilogue.
 * Requirements: K % 64 == 0, lda/ldw % 8 == 0, A/W 16-byte aligned, N % 32 == 0 for SWIGLU.
 */
#define ICL_EPI_BIAS 1
#define ICL_EPI_GELU 2
#define ICL_EPI_RESIDUAL 4
#define ICL_EPI_SWIGLU 8

typedef struct icl_gemm_args {
  const void* A;        /* bf16 [batch][M][lda]            */
  const void* W;        /* bf16 [N][ldw]                   */
  void* C;              /* out_dtype [batch][M][ldc]       */
  const float* bias;    /* f32 [N] or NULL                 */
  const void* R;        /* res_dtype [batch][M][ldr] or NULL */
  float* workspace;     /* f32 split-K partials or NULL    */
  int64_t lda, ldw, ldc, ldr;
  int64_t strideA, strideC, strideR; /* batch strides in elements */
  int32_t M, N, K;
  int32_t batch;
  int32_t epilogue;     /* OR of ICL_EPI_*                 */
  int32_t out_dtype;    /* ICL_BF16 | ICL_F32              */
  int32_t res_dtype;    /* ICL_BF16 | ICL_F32              */
  int32_t split_k;      /* >= 1                            */
  int32_t tile;         /* 0 = auto, 1 = 128x128, 2 = 64x64 (+ split_k), 3 = 256x256 (+ split_k), 4 = decode
                           skinny kernel (M <= 64, batch 1: weights streamed HBM->VGPR, in-block split-K),
                           5 = decode tile for M <= 128 (batch 1, + split_k; 64-row blocks when M <= 64): W must
                           be the decode-packed copy made by icl_pack_decode_weights (ldw is ignored),
                           6 = the skinny kernel (as 4) on the decode-packed copy                       */
} icl_gemm_args;

int icl_gemm_bf16(const icl_gemm_args* args, void* stream);
/* The decode form of "projection back into the residual stream, then the next RMSNorm" (HF LlamaDecoderLayer:
 * hidden = residual + o_proj(attn) ; post_attention_layernorm(hidden) — and down_proj followed by the next layer's
 * input_layernorm / the final norm; reached from models/custom_salmon.py:704-720 through generate): C = R + A W^T in f32
 * (epilogue must be 0 or ICL_EPI_RESIDUAL with an f32 residual, batch 1, f32 output) AND xn[m][:N] = bf16(rmsnorm(C[m]) * gamma).
 * With split_k > 1 the partial slabs are reduced, the residual added, the row stored and normalised by ONE kernel (a launch
 * and a re-read of the row less than icl_gemm_bf16 + icl_rmsnorm; C is bit-identical to that pair, xn sums its squares in a
 * different — fixed — order); with split_k == 1 it is exactly that pair.  tile 3 accepts split_k > 1 since ABI 5
 * (K / split_k >= 128).  No reference counterpart (fusion). */
int icl_gemm_rmsnorm_bf16(const icl_gemm_args* args, const float* gamma, float eps, void* xn, int64_t ld_xn, void* stream);
/* Decode-packed copy of a weight matrix for tile 5: row-major bf16 W [N][ldw] -> out, (ceil(N/16)*16) x K bf16
 * elements: per 16-row block a K-long stream of 1-KB pieces (one per 32-wide k-step) in MFMA operand order, so that the
 * decode kernel's wave-loads are 1 KB contiguous (rows >= N are zero).  A layout copy made once at load time (the
 * prefill kernels keep reading the row-major original: HBM is sized for both); no reference counterpart. */
int icl_pack_decode_weights(const void* W, int64_t ldw, int32_t N, int32_t K, void* out, void* stream);
/* The tile id (1|2|3) that tile == 0 resolves to for this problem (pure host function). */
int icl_gemm_select_tile(int32_t M, int32_t N, int32_t K, int32_t batch, int32_t split_k);

/* ---- K3/K5/K10/K13/K14: fused (flash-style) attention forward ---------------------------
 * O[t][h][:] = softmax_j( scale * Q[t][h]·K[j][h] + bias ) · V[j][h]   per packed sequence.
 * Replaces the SDPA inside HF WhisperEncoderLayer / LlamaDecoderLayer / the BEATs
 * MultiheadAttention (reached from models/custom_salmon.py:550-554 and :630-636/:704-720).
 * Sequences are packed back to back: sequence s owns rows cu_seqlens[s] .. cu_seqlens[s+1]-1
 * of Q, K, V and O (queries and keys share the packing: self-attention).
 *   causal != 0   : key j visible to query i iff j <= i (positions relative to the sequence).
 *   kv_lens       : optional int32 [n_seqs]; keys >= kv_lens[s] are masked (BEATs key padding
 *                   mask; queries still run over the whole sequence).  Must be >= 1.
 *   rel_bias      : optional f32 [n_heads][2*rel_span-1] table and rel_gate f32
 *                   [total_rows][n_heads]: bias(i,j,h) = rel_gate[i][h] *
 *                   rel_bias[h][clamp(j-i, -(rel_span-1), rel_span-1) + rel_span-1]
 *                   (BEATs gated relative position bias, K5).
 * head_dim must be 64 or 128.  Q/K/V row strides are in elements (so a fused QKV buffer can
 * be addressed in place); head h lives at column h*head_dim of its row.
 */
typedef struct icl_attn_args {
  const void* Q; const void* K; const void* V; /* bf16 */
  void* O;                                     /* bf16 [total_rows][ldo] */
  const int32_t* cu_seqlens;                   /* int32 [n_seqs+1] (device) */
  const int32_t* kv_lens;                      /* int32 [n_seqs] or NULL (device) */
  const float* rel_bias;                       /* f32 [n_heads][2*rel_span-1] or NULL */
  const float* rel_gate;                       /* f32 [total_rows][n_heads] or NULL */
  int64_t ldq, ldk, ldv, ldo;
  int32_t n_seqs, max_seqlen, n_heads, head_dim;
  int32_t causal, rel_span;
  float scale;
  int32_t reserved;                            /* 0 */
  int64_t kv_seq_stride, kv_head_stride;       /* both 0: K/V rows are packed like Q (row cu_seqlens[s] + j, head h at
                                                  column h*head_dim, row strides ldk / ldv).  Both > 0: K/V are read from a
                                                  cache, key j of (sequence s, head h) at K + s*kv_seq_stride +
                                                  h*kv_head_stride + j*ldk (elements) — the Llama prefill reads back what
                                                  the fused QKV epilogue appended: [n_seqs][n_heads][max_len][head_dim],
                                                  ldk = ldv = head_dim                                              */
} icl_attn_args;

int icl_attn_fwd_bf16(const icl_attn_args* args, void* stream);

/* ---- K11: single-token decode attention over the KV cache --------------------------------
 * One new query per sequence against cache rows [0, lens[b]) (the new token's K/V must
 * already be in the cache).  Cache layout: [n_seqs][n_heads][max_len][head_dim] bf16.
 * Replaces the cached-key SDPA inside HF GenerationMixin's greedy loop,
 * models/custom_salmon.py:704-720.
 */
int icl_attn_decode_bf16(const void* Q, int64_t ldq, const void* Kc, const void* Vc, void* O,
                         int64_t ldo, const int32_t* lens, int32_t n_seqs, int32_t n_heads,
                         int32_t head_dim, int32_t max_len, float scale, void* stream);

/* ---- K11: the decode step's RoPE + KV-cache append + attention as ONE launch --------------------
 * icl_rope_kv_bf16 (M = n_seqs rows, one new position each) followed by icl_attn_decode_bf16, fused: qkv bf16 [n_seqs][ld] holds
 * the QKV projection's raw q | k | v row of every sequence (column blocks at 0 | k_off | v_off, n_heads*head_dim wide);
 * for sequence b the kernel rotates q and k at position pos[b] (cos / sin f32 [max_pos][head_dim/2]), appends the rotated k
 * and v to cache row seq_ids[b] (NULL: b) at position pos[b], and attends over lens[b] = pos[b] + 1 keys — the appended
 * one taken from registers.  Same rounding points as the two calls (rope_rot8, bf16 q / k): bit-identical output and cache.
 * The qkv buffer is NOT modified (icl_rope_kv_bf16 rotates it in place; nothing reads it afterwards).
 * Replaces apply_rotary_pos_emb + DynamicCache.update + the attention of transformers' LlamaAttention in a decode step
 * (HF generate loop behind models/custom_salmon.py:704-720).
 */
int icl_attn_decode_rope_bf16(const void* qkv, int64_t ld, int64_t k_off, int64_t v_off, const float* cos, const float* sin,
                              const int32_t* pos, const int32_t* seq_ids, void* kcache, void* vcache, void* O, int64_t ldo,
                              const int32_t* lens, int32_t n_seqs, int32_t n_heads, int32_t head_dim, int32_t max_len,
                              float scale, void* stream);

/* ---- K3/K5/K6/K7: LayerNorm;  K10/K14: RMSNorm -----------------------------------------
 * LayerNorm: v = x[m][:] + (res ? alpha * res[m][:] : 0);
 *            y[m][:] = (v - mean(v)) * rsqrt(var(v) + eps) * gamma + beta   (biased variance)
 *   The optional pre-add is BEATs' deep-norm post-LN, LN(x + alpha*residual) (K5); res has
 *   dtype in_dtype and leading dimension ldx.  y2 (optional, bf16, leading dimension ldy2)
 *   receives a second copy of y: post-LN blocks (BEATs, Q-Former) keep the f32 stream in y
 *   and feed the next GEMM from y2.
 * RMSNorm:   y[m][:] = x[m][:] * rsqrt(mean(x^2) + eps) * gamma
 * x is f32 or bf16 (in_dtype), y bf16 or f32 (out_dtype); gamma/beta f32; all math in f32;
 * N % 4 == 0 and N <= 8192.
 * Replaces nn.LayerNorm / LlamaRMSNorm in the HF / SALMONN modules (models/custom_salmon.py
 * :550-554, :630-636) incl. SALMONN's ln_speech / ln_audio (K6).
 */
int icl_layernorm(const void* x, int64_t ldx, const void* res, float alpha, const float* gamma,
                  const float* beta, void* y, int64_t ldy, void* y2, int64_t ldy2, int32_t M,
                  int32_t N, float eps, int32_t in_dtype, int32_t out_dtype, void* stream);
int icl_rmsnorm(const void* x, int64_t ldx, const float* gamma, void* y, int64_t ldy, int32_t M,
                int32_t N, float eps, int32_t in_dtype, int32_t out_dtype, void* stream);

/* ---- K10/K11/K14: rotary embedding + KV-cache append -------------------------------------
 * In place on the q and k column blocks of a fused bf16 QKV buffer [M][ld] (q at column 0,
 * k at column k_off, v at column v_off; n_heads x head_dim each; HF "rotate_half" pairing
 * (i, i+head_dim/2)); cos/sin f32 [max_pos][head_dim/2].  Row m belongs to sequence seq_ids[m]
 * at position pos[m].  If kcache != NULL the rotated k and the v row are also written to the
 * cache ([n_seqs][n_heads][max_len][head_dim]) at row pos[m] of sequence seq_ids[m].
 * Replaces LlamaRotaryEmbedding/apply_rotary_pos_emb + DynamicCache.update of transformers,
 * reached from models/custom_salmon.py:630-636 / :704-720.
 */
int icl_rope_kv_bf16(void* qkv, int64_t ld, int64_t k_off, int64_t v_off, const float* cos,
                     const float* sin, const int32_t* pos, const int32_t* seq_ids, void* kcache,
                     void* vcache, int32_t M, int32_t n_heads, int32_t head_dim, int32_t max_len,
                     void* stream);

/* ---- K10/K11: QKV projection with RoPE + KV-cache append fused into its epilogue -----------
 * icl_gemm_bf16(args) followed by icl_rope_kv_bf16 on its output, as ONE kernel: the 256x256 tile
 * stages its bf16 C tile through LDS, and with head_dim = 128 a tile is two whole heads of q, k or
 * v, so the row phase rotates (q, k), stores the rows and appends k/v to the cache — the QKV rows
 * are not read back from HBM.  Bit-identical to the two-call sequence (same rounding points).
 * args: batch 1, bf16 output, epilogue 0 or ICL_EPI_BIAS, N = 3*n_heads*128 with q|k|v column
 * blocks at 0 | k_off | v_off, n_heads*128 a multiple of 256; the problem must resolve to the
 * 256x256 tile (icl_gemm_select_tile(...) == 3 and K >= 128) — otherwise ICL_EINVAL, and the
 * caller issues the two calls.  Remaining arguments as icl_rope_kv_bf16.  kv_rows_to_c = 0 (needs a cache): the k / v
 * column blocks of C are NOT written — they exist only in the cache, where icl_attn_fwd_bf16 can read them
 * (kv_seq_stride / kv_head_stride); q is always written.
 * Replaces q_proj/k_proj/v_proj + apply_rotary_pos_emb + DynamicCache.update of transformers'
 * LlamaAttention, reached from models/custom_salmon.py:630-636 (prefill).
 */
int icl_gemm_rope_kv_bf16(const icl_gemm_args* args, int64_t k_off, int64_t v_off, const float* cos,
                          const float* sin, const int32_t* pos, const int32_t* seq_ids, void* kcache,
                          void* vcache, int32_t n_heads, int32_t head_dim, int32_t max_len, int32_t kv_rows_to_c,
                          void* stream);

/* ---- K9: token-embedding gather + speech interleave --------------------------------------
 * out[r][:] = src_idx[r] >= 0 ? table[src_idx[r]][:] : speech[-src_idx[r]-1][:]
 * table bf16 [vocab][H]; speech f32 [n_speech_rows][H]; out f32 [rows][H].
 * Replaces embed_tokens + torch.cat interleave of models/custom_salmon.py:189-194, :243-283.
 */
int icl_embed_gather_interleave(const int32_t* src_idx, const void* table, const float* speech,
                                float* out, int32_t rows, int32_t H, int32_t vocab,
                                int32_t n_speech_rows, void* stream);

/* ---- K11: greedy argmax + EOS/pad bookkeeping --------------------------------------------
 * tok = finished[b] ? pad_id : argmax_v logits[b][v] (lowest index on ties);
 * finished[b] |= (tok == eos_id || tok == eos_id2); out_tokens[b*out_stride + step] = tok; next_ids[b] = tok.
 * eos_id2 = -1 when the generation config names one EOS id (HF accepts a list: Qwen2-Audio's is [151643, 151645]).
 * Replaces HF GenerationMixin._sample greedy branch (models/custom_salmon.py:704-720).
 */
int icl_argmax_eos(const float* logits, int64_t ldl, int32_t B, int32_t V, int32_t eos_id, int32_t eos_id2,
                   int32_t pad_id, int32_t* finished, int32_t* out_tokens, int32_t out_stride,
                   int32_t step, int32_t* next_ids, void* stream);

/* ---- K11 (sampled): repetition penalty -> temperature -> top-k -> top-p -> inverse-CDF draw + EOS/pad bookkeeping ----
 * Per sequence b: scores = logits[b] with every token in prev_tokens[b][0..n_prev) rescaled (x<0 ? x*penalty : x/penalty),
 * all divided by temperature; candidates = scores >= the top_k-th largest score (ties kept); probabilities = softmax over
 * the candidates sorted by (score desc, token asc); candidate j is dropped when the mass of candidates j.. is <= 1-top_p
 * (the largest always stays); the token is the first kept candidate whose running mass exceeds uniforms[b] * kept mass.
 * Then exactly icl_argmax_eos's bookkeeping.  work f32 [B][ldw>=V] is scratch; top_k in [1,1024], or top_k == V = "top-k
 * off" (HF: top_k 0 / None): probabilities are then normalised over the whole row, candidates are the 1024 most likely
 * tokens and a nucleus wider than that is truncated to them; any other top_k is ICL_EINVAL.  Greedy search with a
 * repetition penalty is top_k = 1.  Optional debug outputs (NULL to skip): the kept tokens, their renormalised
 * probabilities and their count, dbg_cap entries per sequence.
 * Replaces HF generate(do_sample=True, temperature, top_p, repetition_penalty) as called at models/custom_salmon.py:705-721
 * (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper [generation-config default 50],
 * TopPLogitsWarper, softmax, multinomial).
 */
int icl_sample_eos(const float* logits, int64_t ldl, int32_t B, int32_t V, float* work, int64_t ldw,
                   const int32_t* prev_tokens, int32_t prev_stride, int32_t n_prev, float repetition_penalty,
                   float temperature, int32_t top_k, float top_p, const float* uniforms, int32_t eos_id, int32_t eos_id2,
                   int32_t pad_id, int32_t* finished, int32_t* out_tokens, int32_t out_stride, int32_t step,
                   int32_t* next_ids, int32_t* dbg_ids, float* dbg_probs, int32_t* dbg_count, int32_t dbg_cap,
                   void* stream);

/* ---- K1: Whisper log-mel (f64 STFT, f32 out) ---------------------------------------------
 * wav f32 [n_audio][wav_ld] with valid lengths wav_lens (device int32; samples past the length
 * or past 480000 are treated as zero), -> spec f32 [n_audio][80][3000] (n_mel x 3000) and,
 * if xt != NULL, the conv-stem operand bf16 [n_audio][3002][xt_ld] (time-major, rows 0 and
 * 3001 zero, columns >= n_mel zero).  n_fft=400, hop=160, periodic Hann, reflect-padded
 * centre frames, Slaney mel filters `mel_filters` f64 [n_mel][201], log10, max-8 clamp,
 * (x+4)/4 — WhisperFeatureExtractor semantics (data/model_processors.py:641-645, :659-663).
 * workspace: f32 [n_audio][n_mel][3000] + int32 [n_audio] (raw log10 values + per-audio max).
 */
int icl_logmel_whisper(const float* wav, int64_t wav_ld, const int32_t* wav_lens,
                       const double* mel_filters, int32_t n_mel, int32_t n_audio, float* spec,
                       void* xt, int64_t xt_ld, void* workspace, void* stream);
/* spec f32 [n_audio][n_mel][3000] -> conv-stem operand xt (same layout as above). */
int icl_spec_to_xt(const float* spec, int32_t n_mel, int32_t n_audio, void* xt, int64_t xt_ld,
                   void* stream);

/* ---- K4: BEATs front-end: Kaldi fbank (f64 math, f32 out) ---------------------------------
 * wav (f32, scaled by 2^15 inside) -> fbank f32 [n_audio][max_frames][128]:
 * 25 ms / 10 ms frames (snip_edges), DC removal, pre-emphasis 0.97, povey window, 512-pt
 * power spectrum, 128 Kaldi mel bins (`mel_banks` f64 [128][257]), log(max(x, FLT_EPSILON)),
 * then (x - mean) / (2*std).  Frames >= n_frames(len) are written as the value the reference
 * produces for zero-padded audio only when `wav` itself holds those zeros; rows past
 * max_frames are not touched.  torchaudio.compliance.kaldi.fbank semantics as used by
 * BEATs.preprocess (external SALMONN package; call site models/custom_salmon.py:412-416).
 */
int icl_fbank_kaldi(const float* wav, int64_t wav_ld, const int32_t* wav_lens,
                    const double* mel_banks, int32_t n_audio, int32_t max_frames, float mean,
                    float std, float* fbank, void* stream);

/* ---- K7: window-level Q-Former cross attention (1 query x win keys) ------------------------
 * q bf16 [n_win][ldq] (n_heads x 64), kv bf16 [n_audio*rows_per_audio][ldkv] with K at
 * column 0 and V at column v_off; window w of audio a covers rows a*rows_per_audio + w*win
 * .. +win-1 (SALMONN's F.unfold with kernel == stride is a free view).
 * out bf16 [n_win][ldo].  Replaces BertSelfAttention(cross) of SALMONN's speech_Qformer
 * (call site models/custom_salmon.py:550-554).
 */
int icl_qformer_window_xattn(const void* q, int64_t ldq, const void* kv, int64_t ldkv,
                             int64_t v_off, void* out, int64_t ldo, int32_t n_audio,
                             int32_t win_per_audio, int32_t win, int32_t rows_per_audio,
                             int32_t n_heads, float scale, void* stream);

/* ---- small fused element-wise helpers ----------------------------------------------------- */
/* BEATs gate (K5): from the fused bf16 QKV rows (q block = first n_heads*64 columns, WITH
 * bias, unscaled), gate[m][h] = ga*(gb*grep_a[h]-1)+2 where (ga,gb) =
 * sigmoid(sum4(grep_w[8][64]·q_h + grep_b[8])).  Output f32 [M][n_heads]. */
int icl_beats_gate(const void* qkv, int64_t ld, const float* grep_w, const float* grep_b,
                   const float* grep_a, float* gate, int32_t M, int32_t n_heads, void* stream);
/* y = x (+ y_in) with dtype conversion: out[m][n] = (float)in[m][n] * alpha (+ add[m][n]) */
int icl_axpby_cast(const void* in, int64_t ldi, int32_t in_dtype, const void* add, int64_t lda_,
                   int32_t add_dtype, float alpha, void* out, int64_t ldo, int32_t out_dtype,
                   int32_t M, int32_t N, void* stream);
/* LoRA down-projection written into the K-augmentation columns of the GEMM operand:
 * X[m][K0 + j] = bf16( scale * sum_k X[m][k] * A[j][k] ), j < r_total; A bf16 [r_total][lda_].
 * (peft LoRA restated: W x + (alpha/r) B (A x); models/custom_salmon.py:78-81 config.) */
int icl_lora_down_bf16(void* X, int64_t ldx, int32_t K0, const void* A, int64_t lda_,
                       int32_t r_total, float scale, int32_t M, void* stream);

/* ---- K4/K5: BEATs data movers (so both BEATs convolutions run on icl_gemm_bf16) ----------------
 * icl_beats_patchify: fbank f32 [n_audio][max_frames][128] -> im2col of Conv2d(1->512, k=16, s=16):
 *   out bf16 [total_rows][256], packed row cu_rows[a] + t'*8 + f', column i*16 + j =
 *   fbank[a][16t'+i][16f'+j]  (BEATs.patch_embedding; external SALMONN package, call site
 *   models/custom_salmon.py:412-416).
 * icl_beats_posconv_pack: x f32 [total_rows][channels] (packed by cu_rows; BEATs: 768 channels, 16 groups).  Rows >= valid_rows[a] of
 *   audio a are zeroed in place (backbone `x[padding_mask] = 0`), and xg (bf16) receives per audio
 *   the image [groups][T_a + 128][channels/groups] with 64 zero rows in front / 64 behind, located at
 *   element offset (cu_rows[a] + 128*a)*channels: row t of group g then sees the 128x48 taps of
 *   Conv1d(768,768,k=128,pad=64,groups=16) as ONE contiguous K = 6144 run (lda = 48).
 * icl_gather_rows_f32: out[r][:] = src[idx[r]][:]  (last-token rows for the LM head).
 */
int icl_beats_patchify(const float* fbank, int32_t max_frames, const int32_t* cu_rows, int32_t n_audio,
                       int32_t total_rows, void* out, void* stream);
int icl_beats_posconv_pack(float* x, const int32_t* cu_rows, const int32_t* valid_rows, int32_t n_audio,
                           int32_t total_rows, int32_t channels, int32_t groups, void* xg, void* stream);
int icl_gather_rows_f32(const float* src, int64_t ld_src, const int32_t* idx, float* out, int64_t ld_out,
                        int32_t rows, int32_t N, void* stream);

/* ---- K11 (beam search): one step of HF's static-shaped beam search + the cache reorder ------------------------------
 * icl_beam_step, per batch row b (num_beams K <= 8, max_new_tokens T <= 64, V >= 2K): log-softmax of the K beams' logits
 *   (row b*rows_per_batch + k; rows_per_batch == 1 at step 0, where the K beams still share the prompt's one distribution) plus
 *   run_score[b][k]; the 2K best continuations (3K when eos_id2 >= 0: HF keeps (1 + number of EOS ids) * K; ties: lower
 *   beam*V + token); a continuation stops when its token is eos_id / eos_id2 or step + 1 == T.  run_* <- the K best that do not stop (score - 1e9 if only stopping ones are left); fin_* <- the K best of
 *   {old finished slots, first-K continuations that stop, scored sum / (step+1)**length_penalty}, taken only while unsat[b];
 *   unsat[b] &= (run_score[b][0] / (step+1)**length_penalty beats the worst finished slot, or a slot is still empty).
 *   next_ids[b*K+i] / parent[b*K+i] = token and absolute source row (b*K + parent beam) of running beam i.
 *   repetition_penalty != 1: the log-probability x of every token already in running beam k's sequence becomes x < 0 ? x * p : x / p
 *   before run_score is added (HF applies RepetitionPenaltyLogitsProcessor after log_softmax in beam search).
 *   Initial state: run_score = {0, -1e9, ...}, fin_score = -1e9, fin_flag = fin_len = 0, unsat = 1, sequences = pad.
 *   The answer after T steps is fin_seq[b][0][0 .. fin_len[b][0]).
 *   Replaces HF GenerationMixin._beam_search (early_stopping=False, do_sample=False) behind generate(inputs_embeds=...,
 *   num_beams, length_penalty) at models/custom_salmon.py:704-715 (per-task values: models/multi_task_model.py:142).
 * icl_kv_copy_spans_bf16: for every layer l < n_layers, head h < n_heads and row r < n_rows, copy n (= n_t[r], or n_fixed when
 *   n_t is NULL) positions of head_dim bf16 from src[l][src_seq[r]][h][src_t0[r] ..] to dst[l][dst_seq[r]][h][dst_t0[r] ..]
 *   (NULL seq array = r, NULL t0 array = 0; strides in elements).  src and dst spans must not overlap.  src / dst hold
 *   src_n_seqs / dst_n_seqs sequences of src_len / dst_len positions: ids, starts and counts read from device memory are
 *   clamped to those extents (ABI 5), so a corrupted parent id copies a wrong span, never out of bounds.  Stands in for HF's
 *   cache.reorder_cache(beam_idx): only the positions after the prompt differ between the beams of a row.
 *   icl_beam_step treats a NaN logit as -inf (a token that cannot be chosen; icl_argmax_eos does the same), so parent[]
 *   always names a beam of its own row.
 */
int icl_beam_step(const float* logits, int64_t ldl, int32_t rows_per_batch, int32_t B, int32_t V, int32_t num_beams,
                  int32_t max_new_tokens, int32_t step, int32_t eos_id, int32_t eos_id2, float length_penalty,
                  float repetition_penalty, float* run_score,
                  int32_t* run_seq, float* fin_score, int32_t* fin_seq, int32_t* fin_len, int32_t* fin_flag,
                  int32_t* unsat, int32_t* next_ids, int32_t* parent, void* stream);
int icl_kv_copy_spans_bf16(const void* src, void* dst, int64_t src_layer_stride, int64_t src_seq_stride,
                           int64_t src_head_stride, int64_t dst_layer_stride, int64_t dst_seq_stride,
                           int64_t dst_head_stride, const int32_t* src_seq, const int32_t* src_t0,
                           const int32_t* dst_seq, const int32_t* dst_t0, const int32_t* n_t, int32_t n_fixed,
                           int32_t n_rows, int32_t n_layers, int32_t n_heads, int32_t head_dim,
                           int32_t src_n_seqs, int32_t dst_n_seqs, int32_t src_len, int32_t dst_len, void* stream);

/* ---- K12: causal-LM cross entropy (teacher-forced forward only) -------------------------------
 * row_loss[r] = logsumexp(logits[r][:]) - logits[r][labels[r]] for labels[r] in [0,V), else 0;
 * mean_loss[0] = mean of row_loss over the valid rows (NaN if none) — torch CrossEntropyLoss with
 * ignore_index=-100, reduction='mean'.  The caller passes row r = position t of a sequence together
 * with labels[r] = label of position t+1 (HF shift-by-one; models/custom_salmon.py:617-640).
 */
int icl_cross_entropy(const float* logits, int64_t ldl, const int32_t* labels, int32_t M, int32_t V,
                      float* row_loss, float* mean_loss, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICL_HIP_H */
