"""Kernel chains of the ICL hot path: each engine strings libicl_hip entry points together over
buffers that stay resident in HBM (allocated once per shape through ``Workspace``; PyTorch is used
for memory and streams only — every arithmetic step below is a C-ABI call).

Stages:  K1 log-mel -> K2/K3 Whisper encoder ‖ K4/K5 BEATs -> K6 LN+concat ->
K7 window Q-Former -> K8 projector -> K9 embedding interleave -> K10 Llama prefill -> K11 decode.
Reference call sites: models/custom_salmon.py:546-554 (encode_speech), :115-299 (prompt wrap),
:556-640 (forward), :642-739 (generate_output).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import binding as B
from .packing import PackedBeats, PackedLlama, PackedQFormer, PackedWhisper

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class Workspace:
    """Named device buffers with a CAPACITY: one flat allocation per (name, dtype) that only ever grows to the largest
    request seen, handed out as a leading ``[:numel]`` view in the requested shape.  Ragged batches (a new total row count
    on almost every call) therefore reuse the same storage — resident bytes are bounded by the largest batch, not by
    the number of distinct shapes — and the steady-state loop performs no allocation.

    ``generation`` counts reallocations of the buffers a captured decode graph points into (``GRAPH_VISIBLE`` name prefixes):
    anything that baked raw pointers must be dropped when they move (``CausalLMRuntimeMixin`` does).  ``zero=True`` buffers carry regions the kernels never write and rely on being
    zero (conv padding rows, the LoRA augmentation tail): those regions sit at fixed flat offsets for fixed inner
    dimensions, so the buffer is zero-filled when it is (re)allocated and again whenever the inner dimensions change."""

    GRAPH_VISIBLE = ("dc_", "gen_", "kv_")      # buffers a captured decode graph bakes pointers into (decode_step, generate, cache)

    def __init__(self, device):
        self.device = torch.device(device)
        self._bufs: Dict[tuple, list] = {}      # (name, dtype) -> [flat tensor, inner dims of the last zero=True request]
        self.generation = 0

    def get(self, name: str, shape: Sequence[int], dtype, zero: bool = False) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        key = (name, dtype)
        ent = self._bufs.get(key)
        if ent is None or ent[0].numel() < n:
            # 1/16 of headroom on every allocation: ragged batches break their own size record by a fraction of a percent at a
            # time (128 prompts of 373-376 positions), and every reallocation drops the captured decode graphs — without it a
            # fresh runtime spends its first batches on eager, regrow + eager, capture
            cap = max(n, 1) + (max(n, 1) >> 4)
            if ent is not None:
                ent[0] = None                   # release the old block before asking for the larger one
                if name.startswith(self.GRAPH_VISIBLE):
                    self.generation += 1        # only a move of something a graph points into retires the graphs: an encoder
                                                # or prefill buffer growing (a longer clip, a longer prompt) leaves them valid
            flat = (torch.zeros if zero else torch.empty)(cap, dtype=dtype, device=self.device)
            ent = self._bufs[key] = [flat, shape[1:]]
        elif zero and ent[1] != shape[1:]:
            ent[0].zero_()
            ent[1] = shape[1:]
        return ent[0][:n].view(shape)

    def nbytes(self) -> int:
        return sum(e[0].numel() * e[0].element_size() for e in self._bufs.values())

    def release(self, *names: str) -> None:
        """Drop the named buffers (re-created on demand).  A graph-visible name counts as a move."""
        for key in [k for k in self._bufs if k[0] in names]:
            del self._bufs[key]
            if key[0].startswith(self.GRAPH_VISIBLE):
                self.generation += 1

    def clear(self) -> None:
        """Drop every buffer (they are re-created on demand).  Counts as a move: captured graphs are retired."""
        self._bufs.clear()
        self.generation += 1


def _i32(x, device) -> torch.Tensor:
    return torch.as_tensor(x, dtype=I32).to(device, non_blocking=True)


# ================================================================================================
# K1: log-mel + conv-stem operand
# ================================================================================================
class LogMel:
    def __init__(self, n_mels: int, device):
        # Slaney filter bank, f64 [n_mels, 201] (restated in runtime/audio_tables.py; data, not arithmetic)
        from .audio_tables import slaney_mel_filters
        self.n_mels = n_mels
        self.device = torch.device(device)
        self.filters = torch.from_numpy(slaney_mel_filters(n_mels)).to(self.device)

    def __call__(self, ws: Workspace, wav: torch.Tensor, wav_lens: torch.Tensor, want_spec: bool = False):
        """wav f32 [n, L] (device), wav_lens int32 [n] -> (xt bf16 [n, 3002, 128], spec f32 [n, n_mels, 3000] | None)"""
        n = wav.shape[0]
        xt = ws.get("logmel_xt", (n, 3002, 128), BF16)
        spec = ws.get("logmel_spec", (n, self.n_mels, 3000), F32) if want_spec else None
        scratch = ws.get("logmel_ws", (n * self.n_mels * 3000 + n,), F32)
        B.logmel_whisper(wav, wav_lens, self.filters, self.n_mels, spec, xt, scratch)
        return xt, spec

    def from_spectrogram(self, ws: Workspace, spec: torch.Tensor) -> torch.Tensor:
        n = spec.shape[0]
        xt = ws.get("logmel_xt", (n, 3002, 128), BF16)
        B.spec_to_xt(spec.contiguous(), xt)
        return xt


# ================================================================================================
# K2 + K3: Whisper encoder
# ================================================================================================
class WhisperEncoderHIP:
    def __init__(self, w: PackedWhisper):
        self.w = w

    def forward(self, ws: Workspace, xt: torch.Tensor, kv_lens: Optional[torch.Tensor] = None,
                final_ln: bool = True) -> torch.Tensor:
        """xt bf16 [n, 3002, 128] -> final-LayerNorm output f32 [n*1500, d] (or the pre-LN stream if not final_ln).
        kv_lens (device int32 [n]): key padding length per audio (Qwen2-Audio masks encoder frames past the audio)."""
        w, c = self.w, self.w.cfg
        n, d, T = xt.shape[0], c.d_model, c.n_ctx
        M = n * T
        x2 = ws.get("wh_x2", (n, 3002, d), BF16, zero=True)      # conv1 output, rows 0 / 3001 stay zero
        h = ws.get("wh_h", (M, d), F32)
        xn = ws.get("wh_xn", (M, d), BF16)
        qkv = ws.get("wh_qkv", (M, 3 * d), BF16)
        att = ws.get("wh_att", (M, d), BF16)
        ff = ws.get("wh_ff", (M, c.ffn), BF16)
        out = ws.get("wh_out", (M, d), F32)
        cu = ws.get("wh_cu", (n + 1,), I32)
        cu.copy_(torch.arange(0, M + 1, T, dtype=I32), non_blocking=True)
        # conv1 (k=3,p=1) as GEMM over the time-major padded mel: row t = xt[t:t+3, :128] (K = 384)
        B.gemm(xt, w.conv1_w, x2[:, 1:], bias=w.conv1_b, gelu=True, M=3000, K=384, lda=128, batch=n,
               stride_a=3002 * 128, stride_c=3002 * d)
        # conv2 (k=3,s=2,p=1): row t = x2[2t:2t+3, :] (lda = 2d, K = 3d), + GELU, + positional embedding
        B.gemm(x2, w.conv2_w, h, bias=w.conv2_b, gelu=True, residual=w.pos, M=T, K=3 * d, lda=2 * d, batch=n,
               stride_a=3002 * d, stride_c=T * d, stride_r=0)
        D = d // c.n_heads
        for L in w.layers:
            B.layernorm(h, L.ln1_g, L.ln1_b, xn, 1e-5)
            B.gemm(xn, L.wqkv, qkv, bias=L.bqkv)
            B.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], att, cu, T, c.n_heads, D, D ** -0.5, kv_lens=kv_lens)
            B.gemm(att, L.wo, h, bias=L.bo, residual=h)
            B.layernorm(h, L.ln2_g, L.ln2_b, xn, 1e-5)
            B.gemm(xn, L.w1, ff, bias=L.b1, gelu=True)
            B.gemm(ff, L.w2, h, bias=L.b2, residual=h)
        if not final_ln:
            return h
        B.layernorm(h, w.lnf_g, w.lnf_b, out, 1e-5)
        return out


# ================================================================================================
# K13: Qwen2-Audio tower = Whisper-style encoder (128 mel, key padding) -> AvgPool1d(2,2) -> ln_post -> projector
# ================================================================================================
class QwenAudioTowerHIP:
    """HF Qwen2AudioEncoder + Qwen2AudioMultiModalProjector (modeling_qwen2_audio.py:289-421), reached from the
    reference at models/custom_qwen.py:188-195 / :228-234."""

    def __init__(self, enc: WhisperEncoderHIP, proj_w: torch.Tensor, proj_b: torch.Tensor, llm_hidden: int):
        self.enc, self.proj_w, self.proj_b, self.llm_hidden = enc, proj_w, proj_b, llm_hidden

    @staticmethod
    def output_lengths(mel_len: int):
        feat = (mel_len - 1) // 2 + 1          # after the stride-2 conv
        return feat, (feat - 2) // 2 + 1       # after AvgPool1d(2, 2)

    def forward(self, ws: Workspace, xt: torch.Tensor, mel_lens: List[int]) -> Tuple[torch.Tensor, List[int]]:
        """xt bf16 [n, 3002, 128], mel_lens (valid mel frames per audio) -> (features f32 [n*750, H_llm], valid rows per audio)."""
        w, c = self.enc.w, self.enc.w.cfg
        n, d, T = xt.shape[0], c.d_model, c.n_ctx
        lens = [self.output_lengths(int(m)) for m in mel_lens]
        kv = _i32([max(1, f) for f, _ in lens], xt.device)
        h = self.enc.forward(ws, xt, kv_lens=kv, final_ln=False)                 # [n*1500, d] f32
        P = n * (T // 2)
        pooled = ws.get("qa_pool", (P, d), F32)
        even, odd = h.view(P, 2 * d)[:, :d], h.view(P, 2 * d)[:, d:]           # frames 2t / 2t+1 as strided row views
        B.axpby_cast(odd, pooled, alpha=0.5)
        B.axpby_cast(even, pooled, alpha=0.5, add=pooled)
        pb = ws.get("qa_pool_bf", (P, d), BF16)
        B.layernorm(pooled, w.lnf_g, w.lnf_b, pb, 1e-5)
        out = ws.get("qa_out", (P, self.llm_hidden), F32)
        B.gemm(pb, self.proj_w, out, bias=self.proj_b)
        return out, [o for _, o in lens]


# ================================================================================================
# K4 + K5: BEATs
# ================================================================================================
class BeatsHIP:
    def __init__(self, w: PackedBeats, device):
        from .audio_tables import kaldi_mel_banks
        self.w = w
        self.banks = torch.from_numpy(kaldi_mel_banks()).to(device)

    @staticmethod
    def frames(n_samples: int) -> int:
        return 0 if n_samples < 400 else 1 + (n_samples - 400) // 160

    @classmethod
    def tokens(cls, n_samples: int) -> int:
        return (cls.frames(n_samples) // 16) * 8

    def forward(self, ws: Workspace, wav: torch.Tensor, padded_lens: List[int], valid_lens: List[int]):
        """wav f32 [n, L] zero padded.  Audio a is processed over padded_lens[a] samples (what the reference's
        BEATs sees: the collated, padded waveform) with keys/rows beyond valid_lens[a] masked.
        Returns (x f32 [sum T_a, d] packed, cu (host list), T list)."""
        w, c = self.w, self.w.cfg
        dev = wav.device
        n, d = wav.shape[0], c.d_model
        nf = [self.frames(L) for L in padded_lens]
        T = [(f // 16) * 8 for f in nf]
        assert min(T) > 0, "audio shorter than one BEATs patch (16 frames)"
        # relative positions beyond the packed span are clamped by the kernel: exact once the span covers max_distance, where
        # the bucket function has saturated (BEATs: 800 < 1504), so clips longer than 30 s need no larger table
        assert max(T) <= w.rel_span or w.rel_span > c.max_distance, "audio longer than the packed relative-position span"
        cu_h = [0]
        for t in T:
            cu_h.append(cu_h[-1] + t)
        M = cu_h[-1]
        # valid token count per audio: BEATs.forward_padding_mask applied twice (frames, then patches)
        valid_T = []
        for a in range(n):
            vf = self._valid_units(valid_lens[a], padded_lens[a], nf[a])
            valid_T.append(max(1, self._valid_units(vf, nf[a], T[a])))
        max_frames = max(nf)
        fb = ws.get("be_fbank", (n, max_frames, 128), F32)
        B.fbank_kaldi(wav, _i32(padded_lens, dev), self.banks, max_frames, c.fbank_mean, c.fbank_std, fb)
        cu = _i32(cu_h, dev)
        valid = _i32(valid_T, dev)
        patches = ws.get("be_patches", (M, 256), BF16)
        B.beats_patchify(fb, cu, M, patches)
        e = ws.get("be_e", (M, c.embed), F32)
        eb = ws.get("be_eb", (M, c.embed), BF16)
        B.gemm(patches, w.patch_w, e)
        B.layernorm(e, w.ln0_g, w.ln0_b, eb, 1e-5)
        x = ws.get("be_x", (M, d), F32)
        B.gemm(eb, w.proj_w, x, bias=w.proj_b)
        # positional grouped conv as 16 GEMMs per audio over the padded per-group image
        G, cpg = c.conv_groups, d // c.conv_groups
        xg = ws.get("be_xg", ((M + 128 * n) * d,), BF16)
        B.beats_posconv_pack(x, cu, valid, n, M, G, xg)
        y = ws.get("be_y", (M, d), F32)
        uniform = all(t == T[0] for t in T)
        for g in range(G):
            wg, bg = w.posconv_w[g], w.posconv_b[g * cpg:(g + 1) * cpg]
            if uniform:
                a_view = xg[g * (T[0] + 128) * cpg:]
                B.gemm(a_view, wg, y[:, g * cpg:], bias=bg, gelu=True, residual=x[:, g * cpg:], M=T[0], K=128 * cpg,
                       lda=cpg, batch=n, stride_a=(T[0] + 128) * d, stride_c=T[0] * d, stride_r=T[0] * d)
            else:
                for a in range(n):
                    base = (cu_h[a] + 128 * a) * d + g * (T[a] + 128) * cpg
                    B.gemm(xg[base:], wg, y[cu_h[a]:cu_h[a + 1], g * cpg:], bias=bg, gelu=True,
                           residual=x[cu_h[a]:cu_h[a + 1], g * cpg:], M=T[a], K=128 * cpg, lda=cpg)
        xb = ws.get("be_xb", (M, d), BF16)
        B.layernorm(y, w.enc_ln_g, w.enc_ln_b, x, 1e-5, out2=xb)
        qkv = ws.get("be_qkv", (M, 3 * d), BF16)
        att = ws.get("be_att", (M, d), BF16)
        o = ws.get("be_o", (M, d), F32)
        ff = ws.get("be_ff", (M, c.ffn), BF16)
        gate = ws.get("be_gate", (M, c.n_heads), F32)
        alpha = c.deep_norm_alpha
        for L in w.layers:
            B.gemm(xb, L.wqkv, qkv, bias=L.bqkv)
            B.beats_gate(qkv, L.extra["grep_w"], L.extra["grep_b"], L.extra["grep_a"], gate, c.n_heads)
            B.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], att, cu, max(T), c.n_heads, 64, 0.125,
                       kv_lens=valid, rel_bias=w.rel_table, rel_gate=gate, rel_span=w.rel_span)
            B.gemm(att, L.wo, o, bias=L.bo)
            B.layernorm(o, L.ln1_g, L.ln1_b, x, 1e-5, res=x, alpha=alpha, out2=xb)   # LN(alpha*x + attn)
            B.gemm(xb, L.w1, ff, bias=L.b1, gelu=True)
            B.gemm(ff, L.w2, o, bias=L.b2)
            B.layernorm(o, L.ln2_g, L.ln2_b, x, 1e-5, res=x, alpha=alpha, out2=xb)
        return x, cu_h, T

    @staticmethod
    def _valid_units(valid_in: int, total_in: int, n_units: int) -> int:
        """Number of leading units NOT fully padded under BEATs.forward_padding_mask: the mask over `total_in`
        inputs is trimmed to a multiple of n_units and a unit is padding iff ALL its inputs are padding."""
        if n_units == 0:
            return 0
        per = total_in // n_units
        if per == 0:
            return n_units
        return min(n_units, -(-valid_in // per)) if valid_in > 0 else 0


# ================================================================================================
# K6 + K7 + K8: LN + concat, window-level Q-Former, projector
# ================================================================================================
class SpeechQFormerHIP:
    def __init__(self, w: PackedQFormer, whisper_d: int, beats_d: int, llm_hidden: int):
        self.w, self.whisper_d, self.beats_d, self.llm_hidden = w, whisper_d, beats_d, llm_hidden
        c = w.cfg
        self.win = round(1500 * c.second_per_window / 30.0)
        self.stride = round(1500 * c.second_stride / 30.0)
        assert self.win == self.stride, "HIP path implements non-overlapping windows (kernel == stride, SURVEY.md A8)"

    def n_windows(self, T: int = 1500) -> int:
        return (T - self.win) // self.stride + 1

    def forward(self, ws: Workspace, speech: torch.Tensor, n: int, audio: Optional[torch.Tensor] = None,
                audio_cu: Optional[List[int]] = None) -> torch.Tensor:
        """speech f32 [n*1500, dw] (Whisper out), audio f32 [sum T_a, db] packed (BEATs out) -> f32 [n*W, H_llm], W = the largest
        window count of the batch (``last_windows[a]`` rows of audio a are its tokens, the rest zero).  SALMONN pads the SHORTER
        stream with zero frames (external package, `_encode_auditory_feature`; call site models/custom_salmon.py:420-430): a
        clip of up to 30 s has 1500 frames -> 88 windows; a longer one keeps its T_a > 1500 BEATs frames next to Whisper's
        1500 (the feature extractor truncates) + zeros -> (T_a - 17) // 17 + 1 windows."""
        T0 = 1500
        frames = [T0] * n if audio is None else [max(T0, audio_cu[a + 1] - audio_cu[a]) for a in range(n)]
        self.last_windows = [self.n_windows(f) for f in frames]
        if all(f == T0 for f in frames):
            return self._run(ws, speech, n, audio, audio_cu, T0)
        Wmax = max(self.last_windows)
        full = ws.get("qf_out_ragged", (n, Wmax, self.llm_hidden), F32)
        full.zero_()
        a0 = 0
        while a0 < n:                                  # runs of consecutive audios with the same frame count
            a1 = a0 + 1
            while a1 < n and frames[a1] == frames[a0]:
                a1 += 1
            cu = [x - audio_cu[a0] for x in audio_cu[a0:a1 + 1]]
            out = self._run(ws, speech[a0 * T0:a1 * T0], a1 - a0, audio[audio_cu[a0]:audio_cu[a1]], cu, frames[a0])
            nw = self.n_windows(frames[a0])
            full[a0:a1, :nw].copy_(out.view(a1 - a0, nw, self.llm_hidden))
            a0 = a1
        return full.view(n * Wmax, self.llm_hidden)

    def _run(self, ws: Workspace, speech: torch.Tensor, n: int, audio: Optional[torch.Tensor], audio_cu: Optional[List[int]],
             T: int) -> torch.Tensor:
        """n audios of T >= 1500 frames each (Whisper's 1500 + zero frames; BEATs' T_a <= T + zero frames) -> [n * n_windows(T), H_llm]."""
        w, c = self.w, self.w.cfg
        dw, db, hq = self.whisper_d, self.beats_d, c.hidden
        C = dw + (db if audio is not None else 0)
        assert C == c.enc_width, f"Q-Former encoder width {c.enc_width} != {C}"
        cat = ws.get("qf_cat", (n * T, C), BF16)
        if T == 1500:
            B.layernorm(speech, w.ln_speech_g, w.ln_speech_b, cat, 1e-5, N=dw)
        else:
            cat.zero_()                               # F.pad of the Whisper stream: frames 1500 .. T-1 are zero AFTER ln_speech
            for a in range(n):
                B.layernorm(speech[a * 1500:(a + 1) * 1500], w.ln_speech_g, w.ln_speech_b, cat[a * T:a * T + 1500], 1e-5, N=dw)
        if audio is not None:
            if T == 1500:
                cat[:, dw:].zero_()   # F.pad of the shorter BEATs stream (memset; rows past T_a stay zero)
            for a in range(n):
                ta = audio_cu[a + 1] - audio_cu[a]
                assert ta <= T
                B.layernorm(audio[audio_cu[a]:audio_cu[a] + ta], w.ln_audio_g, w.ln_audio_b,
                            cat[a * T:a * T + ta, dw:], 1e-5, N=db)
        nw = self.n_windows(T)
        W = n * nw
        h = ws.get("qf_h", (W, hq), F32)
        hb = ws.get("qf_hb", (W, hq), BF16)
        t32 = ws.get("qf_t32", (W, hq), F32)
        tb = ws.get("qf_tb", (W, hq), BF16)
        q = ws.get("qf_q", (W, hq), BF16)
        kv = ws.get("qf_kv", (n * T, 2 * hq), BF16)
        ff = ws.get("qf_ff", (W, c.ffn), BF16)
        # query token -> embeddings.LayerNorm, broadcast to every window (row stride 0 source)
        q0 = ws.get("qf_q0", (1, hq), F32)
        B.layernorm(w.query, w.emb_ln_g, w.emb_ln_b, q0, c.ln_eps)
        B.axpby_cast(q0.expand(W, hq), h)
        B.axpby_cast(q0.expand(W, hq), hb)
        for L in w.layers:
            # self-attention over ONE token: softmax == 1, context = value projection
            B.gemm(hb, L.sa_wv, tb, bias=L.sa_bv)
            B.gemm(tb, L.sa_wo, t32, bias=L.sa_bo, residual=h)
            B.layernorm(t32, L.sa_ln_g, L.sa_ln_b, h, c.ln_eps, out2=hb)
            # cross-attention: 1 query x 17 window frames
            B.gemm(hb, L.ca_wq, q, bias=L.ca_bq)
            B.gemm(cat, L.ca_wkv, kv, bias=L.ca_bkv)
            B.qformer_window_xattn(q, kv, hq, tb, n, nw, self.win, T, c.n_heads, 0.125)
            B.gemm(tb, L.ca_wo, t32, bias=L.ca_bo, residual=h)
            B.layernorm(t32, L.ca_ln_g, L.ca_ln_b, h, c.ln_eps, out2=hb)
            # query feed-forward
            B.gemm(hb, L.w1, ff, bias=L.b1, gelu=True)
            B.gemm(ff, L.w2, t32, bias=L.b2, residual=h)
            B.layernorm(t32, L.ff_ln_g, L.ff_ln_b, h, c.ln_eps, out2=hb)
        out = ws.get("qf_out", (W, self.llm_hidden), F32)
        B.gemm(hb, w.proj_w, out, bias=w.proj_b)
        return out


# ================================================================================================
# K9 + K10 + K11 (+K12 host side): Llama
# ================================================================================================
class LlamaHIP:
    decode_packed_weights = True     # micro-batch <= 256: decode GEMMs stream decode-packed copies of the layer weights
    fuse_decode_norms = True         # decode: o_proj / down_proj + the RMSNorm that follows them in one call (icl_gemm_rmsnorm_bf16)
    fuse_decode_rope = os.environ.get("ICL_FUSE_DECODE_ROPE", "1") != "0"   # decode: RoPE + K/V append inside the attention launch (icl_attn_decode_rope_bf16)

    # Decode GEMMs at 129..256 rows that run on the 256x256 tile with split-K instead of the decode tile (tile 5).  Measured at the
    # Llama-2-7B shapes, 256 rows, GEMM + slab reduction (tools/decode_gemm_time.py, profiles/r04_decode_gemm_ab.txt): qkv 54.8 us on
    # the decode tile -> 46.6 (split 4); o 32.2 -> 29.5 (split 8); gate/up 79.6 -> 67.1 (split 2); down 48.5 -> 49.7 stand-alone, but
    # IN SITU (tools/ab_env.sh, four interleaved rounds: profiles/r04_decode_gemm_ab.txt) the decode phase is 2.3 ms faster with `down`
    # on the 256 tile as well (split 12) — all four.  ICL_DECODE_T256 = comma list of qkv / o / gu / down overrides ("" = decode tile only).
    decode_t256 = tuple(x for x in os.environ.get("ICL_DECODE_T256", "qkv,o,gu,down").split(",") if x)

    def _t256_split(self, N: int, K: int) -> int:
        """Split-K of a 256-row decode GEMM on the 256x256 tile: one M-tile, ceil(N / 256) N-tiles; the slices fill ~80 % of the
        CUs (a second partial round of blocks costs more than idle CUs) and stay >= 8 K-tiles deep (the pipeline's fill / drain);
        0 = leave the GEMM on the decode tile (too few blocks either way)."""
        tiles = (N + 255) // 256
        split = min(K // 512, int(0.8 * self.n_cu) // tiles)
        return split if split >= 2 and tiles * split >= 0.45 * self.n_cu else 0

    def __init__(self, w: PackedLlama, device, decode_packed: Optional[bool] = None, pack_now: bool = True):
        """``decode_packed`` (default: the class attribute): keep a second, decode-packed layout of the layer weights (+12.9 GB
        at 7B, +25 GB at 13B).  ``pack_now=False`` defers the copy to the first decode step — forward-only users
        (``forward_logits`` / loss) then never pay for it; the default packs at load time, not inside a caller's first batch."""
        self.w = w
        self.device = torch.device(device)
        self.n_cu = max(B.device_cu_count(), 1)
        if decode_packed is not None:
            self.decode_packed_weights = bool(decode_packed)
        if self.decode_packed_weights and pack_now:
            self.ensure_decode_packed()

    def ensure_decode_packed(self):
        """Decode-packed copies of the layer weights (second layout of the same bytes: +12.9 GB at 7B, +25 GB at 13B)."""
        c = self.w.cfg
        for L in self.w.layers:
            if getattr(L, "decode_packed", None) is None:
                L.decode_packed = (B.pack_decode_weights(L.wqkv, K=self.w.k_aug), B.pack_decode_weights(L.wo, K=c.hidden),
                                   B.pack_decode_weights(L.wgu, K=c.hidden), B.pack_decode_weights(L.wdown, K=c.ffn))

    # ---- K9 ------------------------------------------------------------------------------------
    def embed(self, ws: Workspace, src_idx: torch.Tensor, speech: Optional[torch.Tensor], name: str = "ll_h") -> torch.Tensor:
        h = ws.get(name, (src_idx.numel(), self.w.cfg.hidden), F32)
        B.embed_gather_interleave(src_idx, self.w.embed, speech, h)
        return h

    # ---- one decoder layer over M packed rows ---------------------------------------------------
    def _layer(self, ws: Workspace, L, h, M: int, tag: str, attn_fn, pos, seq_ids, kc, vc, max_len: int,
               split: Optional[dict] = None, kv_rows_to_c: bool = True, xn_ready: bool = False, next_norm=None,
               attn_does_rope: bool = False):
        """``xn_ready``: the previous call has already written this layer's normalised input (decode: fused into the reduction
        of the previous down_proj).  ``next_norm`` = (gamma, out bf16 [M, >= hidden]) of the RMSNorm that follows this layer
        (the next layer's input norm into the same ``xn`` buffer, or the final norm): decode fuses it into the down_proj's
        split-K reduction, as it does the post-attention norm into the o_proj's (icl_gemm_rmsnorm_bf16)."""
        c, w = self.w.cfg, self.w
        hd, I, D, H = c.hidden, c.ffn, c.head_dim, c.n_heads
        xn = ws.get(tag + "xn", (M, w.k_aug), BF16, zero=True)   # augmentation tail stays zero
        qkv = ws.get(tag + "qkv", (M, 3 * hd), BF16)
        att = ws.get(tag + "att", (M, hd), BF16)
        act = ws.get(tag + "act", (M, I), BF16)
        sk = split or {}

        def tile_of(name):          # per-GEMM kernel choice of a decode step ("tile_qkv" ...), else the step's common one
            return sk.get("tile_" + name, sk.get("tile", 0))

        def weight_of(name, idx, row_major):       # the decode tiles (5 / 6) stream the decode-packed copy, every other tile the original
            return L.decode_packed[idx] if tile_of(name) in (5, 6) else row_major
        nsplit = max([v for k, v in sk.items() if not k.startswith("tile")], default=1)
        wsk = ws.get(tag + "splitk", (nsplit * M * max(3 * hd, 2 * I),), F32) if nsplit > 1 else None
        if not xn_ready:
            B.rmsnorm(h, L.rms1, xn, c.rms_eps, N=hd)
        if L.lora_a is not None:   # x_aug[:, hd:hd+2r] = x @ (s*A)^T : a skinny GEMM for prefill, a GEMV-style kernel for decode
            # prefill: always the 64x64 tile (the choice must not depend on the batch, or rows would not be batch-invariant);
            # decode: the skinny kernel (one block, K split over its 8 waves) up to 64 rows, the block-per-row kernel above
            # that (an N = 16 GEMM on the 64x64 tile is 2 blocks walking all of K: 40 us vs 15)
            r2 = L.lora_a.shape[0]
            if split is not None and M > 64:
                B.lora_down(xn, hd, L.lora_a, r2, 1.0, M=M)      # the LoRA scale is folded into lora_a at pack time
            else:
                B.gemm(xn, L.lora_a, xn[:, hd:hd + r2], K=hd, tile=4 if split is not None else 2)
        if split is None and B.rope_fusable(M, H, D, L.wqkv.shape[1]):
            # prefill on the 256x256 tile: RoPE + cache append run in the GEMM's staged epilogue (same bits as the two calls)
            B.gemm(xn, L.wqkv, qkv, bias=L.bqkv, tile=3,
                   rope=(hd, 2 * hd, w.rope_cos, w.rope_sin, pos, seq_ids, kc, vc, H, D, max_len, kv_rows_to_c))
        else:
            B.gemm(xn, weight_of("qkv", 0, L.wqkv), qkv, bias=L.bqkv, split_k=sk.get("qkv", 1),
                   workspace=wsk, tile=tile_of("qkv"), N=3 * hd, K=w.k_aug)
            if not attn_does_rope:      # decode: RoPE + cache append run inside the attention launch (icl_attn_decode_rope_bf16)
                B.rope_kv(qkv, hd, 2 * hd, w.rope_cos, w.rope_sin, pos, seq_ids, kc, vc, H, D, max_len, M=M)
        attn_fn(qkv, att)
        fuse = split is not None and self.fuse_decode_norms
        if fuse:     # decode: h += att Wo^T and the post-attention RMSNorm in one call (one kernel when the GEMM is split-K)
            B.gemm_rmsnorm(att, weight_of("o", 1, L.wo), h, L.rms2, c.rms_eps, xn, residual=h, split_k=sk.get("o", 1),
                           workspace=wsk, tile=tile_of("o"), N=hd, K=hd)
        else:
            B.gemm(att, weight_of("o", 1, L.wo), h, residual=h, split_k=sk.get("o", 1), workspace=wsk, tile=tile_of("o"),
                   N=hd, K=hd)
            B.rmsnorm(h, L.rms2, xn, c.rms_eps, N=hd)
        B.gemm(xn, weight_of("gu", 2, L.wgu), act, swiglu=True, K=hd, split_k=sk.get("gu", 1), workspace=wsk,
               tile=tile_of("gu"), N=2 * I)
        if fuse and next_norm is not None:
            B.gemm_rmsnorm(act, weight_of("down", 3, L.wdown), h, next_norm[0], c.rms_eps, next_norm[1], residual=h,
                           split_k=sk.get("down", 1), workspace=wsk, tile=tile_of("down"), N=hd, K=I)
            return True
        B.gemm(act, weight_of("down", 3, L.wdown), h, residual=h, split_k=sk.get("down", 1), workspace=wsk,
               tile=tile_of("down"), N=hd, K=I)
        return False

    # ---- K10: prefill over ragged packed sequences ------------------------------------------------
    def prefill(self, ws: Workspace, h: torch.Tensor, seq_lens: List[int], cache: Optional["KVCache"] = None) -> torch.Tensor:
        """h f32 [sum S_b, hidden] (modified in place) -> same buffer holding the final hidden states."""
        c = self.w.cfg
        dev = h.device
        M = sum(seq_lens)
        cu_h = [0]
        for s in seq_lens:
            cu_h.append(cu_h[-1] + s)
        assert max(seq_lens) <= c.max_pos
        pos = _i32([p for s in seq_lens for p in range(s)], dev)
        sid = _i32([b for b, s in enumerate(seq_lens) for _ in range(s)], dev)
        cu = _i32(cu_h, dev)
        maxS = max(seq_lens)
        H, D, hd = c.n_heads, c.head_dim, c.hidden

        # With a cache and the fused QKV epilogue, k / v are written ONCE — into the cache — and the attention reads them
        # there ([seq][head][pos][D]: 256-B rows at a 256-B stride instead of a 3*hidden stride); the k / v columns of the QKV
        # buffer are never written.  Without a cache (teacher-forced forward) they stay packed next to q.
        kv_from_cache = cache is not None and B.rope_fusable(M, H, D, self.w.k_aug)

        def attn(qkv, att):
            B.attn_fwd(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], att, cu, maxS, H, D, D ** -0.5, causal=True)

        for i, L in enumerate(self.w.layers):
            kc = cache.k[i] if cache is not None else None
            vc = cache.v[i] if cache is not None else None
            fn = attn
            if kv_from_cache:
                def fn(qkv, att, kc=kc, vc=vc):
                    B.attn_fwd(qkv[:, :hd], kc, vc, att, cu, maxS, H, D, D ** -0.5, causal=True, kv_cache_max_len=cache.max_len)
            self._layer(ws, L, h, M, "pf_", fn, pos, sid, kc, vc, cache.max_len if cache is not None else 0,
                        kv_rows_to_c=not kv_from_cache)
        return h

    def logits(self, ws: Workspace, h_rows: torch.Tensor, name: str = "ll_logits", xn_ready: bool = False) -> torch.Tensor:
        """h_rows f32 [R, hidden] -> logits f32 [R, vocab] (final RMSNorm + lm_head).  ``xn_ready``: the final norm has been
        written into ``name + "_xn"`` already (decode: fused into the last down_proj's reduction)."""
        c = self.w.cfg
        R = h_rows.shape[0]
        xn = ws.get(name + "_xn", (R, c.hidden), BF16)
        out = ws.get(name, (R, c.vocab), F32)
        if not xn_ready:
            B.rmsnorm(h_rows, self.w.norm, xn, c.rms_eps)
        B.gemm(xn, self.w.lm_head, out, tile=4 if R <= 8 else 0)
        return out

    # ---- K11: one decode step for Bn sequences -----------------------------------------------------
    def decode_step(self, ws: Workspace, cache: "KVCache", next_ids: torch.Tensor, pos: torch.Tensor,
                    lens: torch.Tensor, sid: torch.Tensor) -> torch.Tensor:
        c = self.w.cfg
        Bn = next_ids.numel()
        h = self.embed(ws, next_ids, None, name="dc_h")
        H, D = c.n_heads, c.head_dim
        # Weights are streamed once per step, from decode-packed copies of the layer weights (made once, on the first decode
        # step: a second 12.9 GB for Llama-2-7B — the prefill kernels keep the row-major originals; HBM is sized for both): a
        # wave-load is 1 KB contiguous instead of 16 rows x 64 B.  Per layer, rotating weights: Bn <= 8 the skinny kernel
        # (in-block split-K) 90-108 -> 79-87 us (5.1 TB/s at Bn = 1); 9..256 the decode tile (64-, 128- or 256-row blocks, about one
        # block per CU) 115-167 -> 97-132 us at 128 rows, 214 us at 256 (0.84 vs 1.06 us per row: a weight byte serves twice the rows);
        # above 256 the 64x64 LDS tile + split-K on the row-major weights.
        def sk(N, K):
            tiles = ((N + 63) // 64) * ((Bn + 63) // 64)
            s = max(1, min(K // 512, (2 * self.n_cu + tiles - 1) // tiles))
            return min(s, 16)

        def sk5(N, K):
            return max(1, min(self.n_cu // ((N + 127) // 128), K // 512))
        if Bn <= 256 and self.decode_packed_weights:
            self.ensure_decode_packed()
            if Bn <= 8:
                split = dict(tile=6)
            else:
                split = dict(qkv=sk5(3 * c.hidden, self.w.k_aug), o=sk5(c.hidden, c.hidden), gu=sk5(2 * c.ffn, c.hidden),
                             down=sk5(c.hidden, c.ffn), tile=5)
                if Bn > 128:
                    for name, (N, K) in (("qkv", (3 * c.hidden, self.w.k_aug)), ("o", (c.hidden, c.hidden)),
                                         ("gu", (2 * c.ffn, c.hidden)), ("down", (c.hidden, c.ffn))):
                        s256 = self._t256_split(N, K) if name in self.decode_t256 else 0
                        if s256:
                            split[name], split["tile_" + name] = s256, 3
        elif Bn <= 8:
            split = dict(tile=4)
        else:
            split = dict(qkv=sk(3 * c.hidden, self.w.k_aug), o=sk(c.hidden, c.hidden), gu=sk(2 * c.ffn, c.hidden),
                         down=sk(c.hidden, c.ffn), tile=2)

        layers = self.w.layers
        xn_next = ws.get("dc_xn", (Bn, self.w.k_aug), BF16, zero=True)          # the layers' normalised-input buffer (_layer's tag + "xn")
        xn_final = ws.get("dc_logits_xn", (Bn, c.hidden), BF16)
        ready = False
        for i, L in enumerate(layers):
            kc, vc = cache.k[i], cache.v[i]

            # One launch: rotate q / k at pos, append k / v, attend — bit-identical to the two launches, so the choice is free.
            # Same-box A/B (tools/ab_decode_rope.sh, profiles/r04_decode_rope_ab.txt): at 256 rows decode 143.2 -> 142.5 ms; at ONE
            # sequence 58.4 -> 59.6 ms per utterance — the rotation's dependent loads (pos -> cos / sin, q, k) sit in front of a
            # latency-bound attention and cost more than the 5-us launch they replace — so small batches keep the two launches.
            fuse_rope = self.fuse_decode_rope and Bn > 8
            if fuse_rope:
                def attn(qkv, att, kc=kc, vc=vc):
                    B.attn_decode_rope(qkv, c.hidden, 2 * c.hidden, self.w.rope_cos, self.w.rope_sin, pos, sid, kc, vc, att, lens,
                                       H, D, cache.max_len, D ** -0.5)
            else:
                def attn(qkv, att, kc=kc, vc=vc):
                    B.attn_decode(qkv[:, :c.hidden], kc, vc, att, lens, H, D, cache.max_len, D ** -0.5)

            nxt = (layers[i + 1].rms1, xn_next) if i + 1 < len(layers) else (self.w.norm, xn_final)
            ready = self._layer(ws, L, h, Bn, "dc_", attn, pos, sid, kc, vc, cache.max_len, split=split, xn_ready=ready,
                                next_norm=nxt, attn_does_rope=fuse_rope)
        return self.logits(ws, h, name="dc_logits", xn_ready=ready)


class KVCache:
    """bf16 K/V cache, per layer [n_seqs][n_heads][max_len][head_dim] (one contiguous stream per (seq, head)): a view of
    the workspace's single ``kv_k`` / ``kv_v`` allocations, which grow to the largest (n_seqs x max_len) seen."""

    def __init__(self, cfg, n_seqs: int, max_len: int, ws: Workspace):
        self.n_seqs, self.max_len = n_seqs, max_len
        shape = (cfg.n_layers, n_seqs, cfg.n_heads, max_len, cfg.head_dim)
        self.k = ws.get("kv_k", shape, BF16)
        self.v = ws.get("kv_v", shape, BF16)

    def rows(self, b0: int, b1: int) -> "KVCache":
        """The cache of sequences b0 .. b1-1 only (views; sequence ids inside are relative to b0): a prefill over a chunk of
        the batch appends into, and reads from, its own block of the one allocation."""
        sub = object.__new__(KVCache)
        sub.n_seqs, sub.max_len = b1 - b0, self.max_len
        sub.k, sub.v = self.k[:, b0:b1], self.v[:, b0:b1]
        return sub
