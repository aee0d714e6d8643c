"""SalmonnRuntime — the MI355X-native stand-in for the external ``SALMONN.models.salmonn_org.SALMONN``
object that the reference builds at models/custom_salmon.py:96 and drives through
``encode_speech`` (:550), ``llama_model(...)`` (:631) and ``llama_model.generate(...)`` (:705).

It owns the packed weights (HBM resident, bf16), one ``Workspace`` of activation buffers and the four
kernel chains of runtime/engines.py.  Prompts arrive as *segments* (token-id runs and speech-row runs)
so that the string work of ``custom_prompt_wrap`` stays on the host while the interleave itself is one
gather kernel (K9).  Everything arithmetic goes through libicl_hip; there is no CPU fallback.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import binding as B
from .config import SalmonnCfg
from .engines import BF16, F32, I32, BeatsHIP, KVCache, LlamaHIP, LogMel, SpeechQFormerHIP, WhisperEncoderHIP, Workspace, _i32
from .packing import normalize_keys, pack_beats, pack_llama, pack_qformer, pack_whisper

Segment = Union[Sequence[int], Tuple[str, int, int]]  # token ids | ("speech", first_row, n_rows)


def speech_segment(first_row: int, n_rows: int) -> Tuple[str, int, int]:
    return ("speech", first_row, n_rows)


@dataclass
class GenerateResult:
    tokens: torch.Tensor          # int64 [B, width] on CPU — what HF generate() returns with inputs_embeds
    first_logits: Optional[torch.Tensor] = None  # f32 [B, V] (device) logits of the first generated position
    step_logits: Optional[torch.Tensor] = None   # f32 [max_new_tokens, B, V] (device): the logits every token was chosen from
    dropped: Tuple[int, ...] = ()                # rows NOT generated (``overlong="drop"``): pad-filled tokens, NaN logits


class CausalLMRuntimeMixin:
    """K9-K12 over a packed decoder (`self.llama`: LlamaHIP, `self.lm_cfg`: LlamaCfg, `self.ws`, `self.device`):
    prompt segments -> gather/interleave -> prefill -> greedy decode / teacher-forced logits.  Shared by the SALMONN
    (Llama-2 / Vicuna) and Qwen2-Audio runtimes."""

    use_graphs = True          # capture the decode loop in a HIP graph (per batch shape) after its first eager pass

    # --------------------------------------------------------------------------------------------
    # K9: prompt segments -> gather indices
    # --------------------------------------------------------------------------------------------
    def _gather_indices(self, prompts: Sequence[Sequence[Segment]], n_speech_rows: int) -> Tuple[List[int], List[int]]:
        V = self.lm_cfg.vocab
        flat: List[int] = []
        lens: List[int] = []
        for segs in prompts:
            before = len(flat)
            for seg in segs:
                if isinstance(seg, tuple) and len(seg) == 3 and seg[0] == "speech":
                    _, first, cnt = seg
                    if first < 0 or first + cnt > n_speech_rows:
                        raise ValueError(f"speech segment [{first},{first + cnt}) outside {n_speech_rows} speech rows")
                    flat.extend(-(r + 1) for r in range(first, first + cnt))
                else:
                    for t in seg:
                        t = int(t)
                        if not 0 <= t < V:
                            raise ValueError(f"token id {t} outside the vocabulary [0,{V})")
                        flat.append(t)
            if len(flat) == before:
                raise ValueError("empty prompt")
            lens.append(len(flat) - before)
        return flat, lens

    @staticmethod
    def _prompt_lengths(prompts: Sequence[Sequence[Segment]]) -> List[int]:
        """Positions per prompt (host arithmetic only; the validation of ``generate`` runs before anything is launched)."""
        return [sum(seg[2] if (isinstance(seg, tuple) and len(seg) == 3 and seg[0] == "speech") else len(seg) for seg in segs)
                for segs in prompts]

    def embed_prompts(self, prompts, speech: Optional[torch.Tensor], name: str = "ll_h"):
        rows = 0 if speech is None else speech.shape[0] * (speech.shape[1] if speech.dim() == 3 else 1)
        sp2 = None if speech is None else speech.reshape(rows, self.lm_cfg.hidden).contiguous()
        flat, lens = self._gather_indices(prompts, rows)
        h = self.llama.embed(self.ws, _i32(flat, self.device), sp2, name=name)
        return h, lens

    # --------------------------------------------------------------------------------------------
    # K10 (+K12): teacher-forced forward
    # --------------------------------------------------------------------------------------------
    def forward_logits(self, prompts, speech: Optional[torch.Tensor]) -> Tuple[torch.Tensor, List[int]]:
        """All-position logits f32 [sum S_b, V] (packed) and the per-sequence lengths."""
        h, lens = self.embed_prompts(prompts, speech, name="fw_h")
        self.llama.prefill(self.ws, h, lens, cache=None)
        return self.llama.logits(self.ws, h, name="fw_logits"), lens

    def cross_entropy(self, logits: torch.Tensor, shifted_labels: torch.Tensor) -> torch.Tensor:
        """mean CE over rows with label >= 0; logits f32 [M, V] (device), shifted_labels int32 [M]."""
        M = logits.shape[0]
        rows = self.ws.get("ce_rows", (M,), F32)
        mean = self.ws.get("ce_mean", (1,), F32)
        B.cross_entropy(logits, shifted_labels.to(device=self.device, dtype=I32), rows, mean)
        return mean

    # --------------------------------------------------------------------------------------------
    # K10 + K11: greedy generate
    # --------------------------------------------------------------------------------------------
    MAX_GRAPHS = 16            # captured decode graphs kept (one per batch shape / cache length / knob set); oldest dropped

    def _cache(self, n_seqs: int, max_len: int) -> KVCache:
        """K/V views over the workspace's one K and one V allocation (they grow to the largest n_seqs x max_len seen; a
        move bumps ``ws.generation``, which retires every captured graph)."""
        return KVCache(self.lm_cfg, n_seqs, max_len, self.ws)

    prefill_chunk = int(os.environ.get("ICL_PREFILL_CHUNK", "128"))   # sequences per prefill pass of generate() (see there)

    def _graph_lookup(self, gkey):
        """(captured graph | None, warm?) for this key, after retiring everything captured or warmed under an older
        workspace generation: a graph bakes raw pointers into workspace buffers and must not outlive a reallocation."""
        gen = self.ws.generation
        if self._graph_gen != gen:
            self._graphs.clear()
            self._graph_warm.clear()
            self._graph_gen = gen
        return self._graphs.get(gkey), gkey in self._graph_warm

    def generate(self, prompts, speech: Optional[torch.Tensor], max_new_tokens: int = 10, eos_id: Optional[int] = None,
                 pad_id: Optional[int] = None, suppress_eos: bool = False, want_first_logits: bool = False,
                 cache_len_multiple: int = 64, do_sample: bool = False, temperature: float = 1.0, top_p: float = 1.0,
                 top_k: int = 50, repetition_penalty: float = 1.0, generator: Optional[torch.Generator] = None,
                 sample_debug=None, want_step_logits: bool = False, overlong: str = "raise", num_beams: int = 1,
                 length_penalty: float = 1.0, beam_debug: Optional[dict] = None) -> GenerateResult:
        """Greedy search with HF ``generate(inputs_embeds=…)`` semantics (models/custom_salmon.py:704-720): returns only
        the new tokens; a row that has emitted EOS is filled with pad; the width is that of the longest row
        (``min_length`` is a no-op with inputs_embeds).  All steps are enqueued without a host sync; the
        early-stop width is applied on the host afterwards (identical output, no per-token round trip).

        ``do_sample=True`` switches the tail to the sampling kernel (HF's sample mode: repetition penalty → temperature →
        top-k → top-p → draw; ``temperature`` / ``top_p`` / ``top_k`` are ignored otherwise, as in HF); a repetition penalty
        ≠ 1 in greedy mode runs the same kernel with ``top_k = 1``.  The draws are uniforms from ``generator`` (a device
        generator; default: torch's global CUDA generator): reproducible for a seed, not bit-identical to
        ``torch.multinomial``.

        ``overlong``: a row whose prompt + ``max_new_tokens`` exceeds ``max_pos`` is validated on the host BEFORE any launch.
        ``"raise"`` (default) fails the call; ``"drop"`` generates the other rows and reports the row in ``dropped`` (its
        tokens are pad, its logits NaN) — the reference runs batch 1, where one over-long prompt costs one utterance
        (inference/inference.py:370-373), not the batch it happens to be collated with.

        ``num_beams > 1``: HF beam search (``early_stopping=False``; ``length_penalty`` as in HF; see ``_generate_beam``)."""
        c, ws, dev = self.lm_cfg, self.ws, self.device
        if overlong not in ("raise", "drop"):
            raise ValueError(f"overlong must be 'raise' or 'drop', not {overlong!r}")
        limit = c.max_pos // cache_len_multiple * cache_len_multiple      # cache lengths are multiples of cache_len_multiple
        plens = self._prompt_lengths(prompts)
        bad = [b for b, n in enumerate(plens) if n + max_new_tokens > limit]
        if bad and (overlong == "raise" or len(bad) == len(plens)):
            raise ValueError(f"prompt + new tokens of row(s) {bad} ({[plens[b] + max_new_tokens for b in bad]} positions) "
                             f"exceed max_pos {c.max_pos}")
        if bad:
            keep = [b for b in range(len(plens)) if b not in set(bad)]
            sub = self.generate([prompts[b] for b in keep], speech, max_new_tokens=max_new_tokens, eos_id=eos_id, pad_id=pad_id,
                                suppress_eos=suppress_eos, want_first_logits=want_first_logits,
                                cache_len_multiple=cache_len_multiple, do_sample=do_sample, temperature=temperature, top_p=top_p,
                                top_k=top_k, repetition_penalty=repetition_penalty, generator=generator,
                                sample_debug=sample_debug, want_step_logits=want_step_logits, num_beams=num_beams,
                                length_penalty=length_penalty)
            n, kidx = len(plens), torch.tensor(keep)
            toks = torch.full((n, sub.tokens.shape[1]), c.pad_id if pad_id is None else pad_id, dtype=torch.int64)
            toks[kidx] = sub.tokens
            first = steps = None
            if sub.first_logits is not None:
                first = torch.full((n, sub.first_logits.shape[1]), float("nan"), dtype=F32, device=dev)
                first[kidx.to(dev)] = sub.first_logits
            if sub.step_logits is not None:
                steps = torch.full((sub.step_logits.shape[0], n, sub.step_logits.shape[2]), float("nan"), dtype=F32, device=dev)
                steps[:, kidx.to(dev)] = sub.step_logits
            return GenerateResult(tokens=toks, first_logits=first, step_logits=steps, dropped=tuple(bad))
        eos = c.eos_id if eos_id is None else eos_id            # an id, or HF's list form (one or two ids: binding._eos_pair)
        eos = tuple(int(e) for e in eos) if isinstance(eos, (tuple, list)) else int(eos)
        pad = c.pad_id if pad_id is None else pad_id
        if suppress_eos:
            eos = -1  # benchmark mode: exactly max_new_tokens per row
        assert max_new_tokens >= 1
        if num_beams != 1:
            if num_beams < 1:
                raise ValueError(f"num_beams must be >= 1, not {num_beams}")
            if do_sample or want_step_logits:
                raise NotImplementedError("beam search runs without sampling (HF's beam-sample mode) and without a per-step logits trace")
            if num_beams > 8 or max_new_tokens > 64:      # the step kernel's state (include/icl_hip.h); refused before any launch
                raise ValueError(f"beam search supports num_beams <= 8 and max_new_tokens <= 64 (got {num_beams}, {max_new_tokens})")
            return self._generate_beam(prompts, speech, max_new_tokens, eos, pad, num_beams, float(length_penalty),
                                       want_first_logits, cache_len_multiple, debug=beam_debug,
                                       repetition_penalty=float(repetition_penalty))
        h, lens = self.embed_prompts(prompts, speech)
        Bn = len(lens)
        need = max(lens) + max_new_tokens
        max_len = -(-need // cache_len_multiple) * cache_len_multiple
        assert max_len <= c.max_pos, f"prompt + new tokens ({need}) exceeds max_pos {c.max_pos}"   # validated above
        cache = self._cache(Bn, max_len)
        marks = getattr(self, "phase_marks", None)     # optional {name: torch.cuda.Event}: bench.py times prefill / decode with it
        if marks is not None:
            marks["prefill_start"].record()
        # Prefill in chunks of at most ``prefill_chunk`` sequences, decode ALL of them together: the decode GEMMs stream the
        # weights once per step whatever the row count (256 rows cost 0.84 us each against 1.06 at 128), while the prefill
        # GEMMs measured 1.3 % faster per utterance at 128 sequences than at 256.  A sequence's arithmetic does not depend on
        # its neighbours in either phase, so the split changes no result.
        if Bn <= self.prefill_chunk:
            self.llama.prefill(ws, h, lens, cache)
        else:
            r0 = 0
            for b0 in range(0, Bn, self.prefill_chunk):
                b1 = min(Bn, b0 + self.prefill_chunk)
                r1 = r0 + sum(lens[b0:b1])
                self.llama.prefill(ws, h[r0:r1], lens[b0:b1], cache.rows(b0, b1))
                r0 = r1
        cu_last = []
        acc = 0
        for s in lens:
            acc += s
            cu_last.append(acc - 1)
        last = ws.get("gen_last", (Bn, c.hidden), F32)
        B.gather_rows(h, _i32(cu_last, dev), last)
        logits = self.llama.logits(ws, last, name="gen_logits")
        first = logits.clone() if want_first_logits else None
        trace = ws.get("gen_logits_trace", (max_new_tokens, Bn, c.vocab), F32) if want_step_logits else None
        if trace is not None:
            trace[0].copy_(logits)
        if marks is not None:
            marks["prefill_end"].record()
        finished = ws.get("gen_finished", (Bn,), I32)
        finished.zero_()
        toks = ws.get("gen_tokens", (Bn, max_new_tokens), I32)
        nxt = ws.get("gen_next", (Bn,), I32)
        sampled = do_sample or repetition_penalty != 1.0
        if sampled:
            # HF's TopKLogitsWarper clamps top_k to the vocabulary; 0 / None switch it off.  The kernel takes 1..SAMPLE_TOP_K_MAX, or
            # the vocabulary size for "off" (full-row normalisation, nucleus over the 1024 most likely tokens); anything else is
            # a caller's argument error and is refused HERE, before a launch — never as a device-side failure
            top_k = c.vocab if not top_k else min(int(top_k), c.vocab)
            if top_k < 1 or B.SAMPLE_TOP_K_MAX < top_k < c.vocab:
                raise ValueError(f"top_k={top_k}: the sampling kernel supports 1..{B.SAMPLE_TOP_K_MAX}, or 0 / None / >= vocabulary "
                                 f"({c.vocab}) for no top-k filter")
            if (do_sample and not (0.0 < float(top_p) <= 1.0 and float(temperature) > 0.0)) or not float(repetition_penalty) > 0.0:
                raise ValueError(f"sampling knobs out of range: temperature={temperature} (> 0), top_p={top_p} (0 < p <= 1), "
                                 f"repetition_penalty={repetition_penalty} (> 0)")
            knobs = ((float(temperature), top_k, float(top_p)) if do_sample else (1.0, 1, 1.0)) + (float(repetition_penalty),)
            uni = ws.get("gen_uniform", (max_new_tokens, Bn), F32)
            if do_sample and generator is not None and generator.device.type != uni.device.type:
                uni.copy_(torch.rand(uni.shape, generator=generator))      # a host generator: draw there, same stream of uniforms
            elif do_sample:
                uni.uniform_(0.0, 1.0, generator=generator)
            else:
                uni.zero_()
            work = ws.get("gen_sample_work", (Bn, logits.shape[1]), F32)

            def tail(lg, step):
                B.sample_eos(lg, work, uni[step], eos, pad, finished, toks, step, nxt, temperature=knobs[0], top_k=knobs[1],
                             top_p=knobs[2], repetition_penalty=knobs[3], V=c.vocab,
                             debug=sample_debug if step == 0 else None)
        else:
            knobs = None

            def tail(lg, step):
                B.argmax_eos(lg, eos, pad, finished, toks, step, nxt, V=c.vocab)
        tail(logits, 0)
        if max_new_tokens > 1:
            steps = max_new_tokens - 1
            # positions of the fed token / cache length after its append, per step: static buffers (graph-replayable)
            pos_all = ws.get("gen_pos", (steps, Bn), I32)
            len_all = ws.get("gen_len", (steps, Bn), I32)
            sid = ws.get("gen_sid", (Bn,), I32)
            pos_all.copy_(torch.tensor([[s + t for s in lens] for t in range(steps)], dtype=I32), non_blocking=True)
            len_all.copy_(torch.tensor([[s + t + 1 for s in lens] for t in range(steps)], dtype=I32), non_blocking=True)
            sid.copy_(torch.arange(Bn, dtype=I32), non_blocking=True)

            def decode_loop():
                for t in range(steps):
                    lg = self.llama.decode_step(ws, cache, nxt, pos_all[t], len_all[t], sid)
                    if trace is not None:
                        trace[t + 1].copy_(lg)
                    tail(lg, t + 1)

            # The decode loop is launch-bound at small batch (~17 kernels x layers x steps): after one eager pass that
            # sizes every workspace buffer, it is captured ONCE per (batch, cache length, steps, eos, pad) into a HIP
            # graph and replayed — all pointers are workspace-stable and nothing inside synchronises or allocates.
            gkey = (Bn, max_len, steps, eos, pad, knobs, trace is not None)
            graph, warm = self._graph_lookup(gkey) if self.use_graphs else (None, False)
            if graph is not None:
                graph.replay()
            elif warm:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                prof, B.GEMM_PROFILE = B.GEMM_PROFILE, None      # no timing events inside a stream capture
                try:
                    with torch.cuda.graph(g):
                        decode_loop()
                finally:
                    B.GEMM_PROFILE = prof
                while len(self._graphs) >= self.MAX_GRAPHS:
                    self._graphs.pop(next(iter(self._graphs)))
                self._graphs[gkey] = g
                g.replay()
            else:
                decode_loop()
                if self.use_graphs:
                    self._graph_lookup(gkey)                     # the eager pass may have grown buffers: re-sync first
                    self._graph_warm.add(gkey)
        if marks is not None:
            marks["decode_end"].record()
        out = toks.cpu().to(torch.int64)                                                 # the only D2H of the call
        width = max_new_tokens
        if B._eos_pair(eos)[0] >= 0:
            is_eos = torch.isin(out, torch.tensor([e for e in B._eos_pair(eos) if e >= 0]))
            first_eos = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((Bn,), max_new_tokens))
            width = int(first_eos.max())
        return GenerateResult(tokens=out[:, :width].contiguous(), first_logits=first,
                              step_logits=trace.clone() if trace is not None else None)


    beam_rows = 256            # decode rows (sequences x beams) per pass of beam search: the widest decode tile

    def _generate_beam(self, prompts, speech, max_new_tokens: int, eos: int, pad: int, K: int, length_penalty: float,
                       want_first_logits: bool, cache_len_multiple: int, debug: Optional[dict] = None,
                       repetition_penalty: float = 1.0) -> GenerateResult:
        """HF beam search with ``inputs_embeds`` only (models/custom_salmon.py:704-715; transformers `_beam_search`,
        early_stopping=False).  The prompt is prefilled ONCE per row (HF prefills K copies), its K/V rows are copied to the
        row's K beams, and every step is `icl_beam_step` (log-softmax, the 2K best continuations, running / finished
        bookkeeping) -> copy of the generated K/V positions from each beam's parent (`icl_kv_copy_spans_bf16`: the prompt part
        is identical across a row's beams, so HF's whole-cache reorder is a copy of <= max_new_tokens positions) -> one
        decode step over rows x K sequences.  Nothing synchronises with the host until the final D2H of the best hypotheses;
        rows whose search has ended keep stepping with their finished slots frozen (HF stops its loop instead: same result).
        ``debug`` (tests; single pass only): receives per step the logits the step scored, and the running sequences / parents /
        tokens it chose."""
        c, ws, dev = self.lm_cfg, self.ws, self.device
        n_rows = len(prompts)
        per_pass = max(1, self.beam_rows // K)
        if n_rows > per_pass:                                  # rows x beams beyond the decode tile: run the rows in groups
            assert debug is None
            parts = [self._generate_beam(prompts[i:i + per_pass], speech, max_new_tokens, eos, pad, K, length_penalty,
                                         want_first_logits, cache_len_multiple, repetition_penalty=repetition_penalty)
                     for i in range(0, n_rows, per_pass)]
            width = max(p.tokens.shape[1] for p in parts)
            toks = torch.full((n_rows, width), pad, dtype=torch.int64)
            r = 0
            for p in parts:
                toks[r:r + p.tokens.shape[0], :p.tokens.shape[1]] = p.tokens
                r += p.tokens.shape[0]
            first = torch.cat([p.first_logits for p in parts]) if want_first_logits else None
            return GenerateResult(tokens=toks, first_logits=first)
        h, lens = self.embed_prompts(prompts, speech)
        Bn, T = len(lens), max_new_tokens
        BK = Bn * K
        need = max(lens) + T
        max_len = -(-need // cache_len_multiple) * cache_len_multiple
        assert max_len <= c.max_pos, f"prompt + new tokens ({need}) exceeds max_pos {c.max_pos}"   # validated by generate()
        whole = self._cache(BK + Bn, max_len)                  # beams' sequences first, then the rows the prompts prefill into
        cache, pre = whole.rows(0, BK), whole.rows(BK, BK + Bn)
        r0 = 0
        for b0 in range(0, Bn, self.prefill_chunk):
            b1 = min(Bn, b0 + self.prefill_chunk)
            r1 = r0 + sum(lens[b0:b1])
            self.llama.prefill(ws, h[r0:r1], lens[b0:b1], pre.rows(b0, b1))
            r0 = r1
        cu_last, acc = [], 0
        for s in lens:
            acc += s
            cu_last.append(acc - 1)
        last = ws.get("gen_last", (Bn, c.hidden), F32)
        B.gather_rows(h, _i32(cu_last, dev), last)
        logits = self.llama.logits(ws, last, name="gen_logits")
        first = logits.clone() if want_first_logits else None
        st = B.BeamState(ws.get, Bn, K, T, pad)

        def score(lg, step):
            B.beam_step(lg, st, step, eos, length_penalty, V=c.vocab, repetition_penalty=repetition_penalty)
            if debug is not None:
                for key, val in (("logits", lg), ("run_seq", st.run_seq), ("parent", st.parent), ("next", st.next_ids),
                                 ("fin_seq", st.fin_seq), ("fin_score", st.fin_score), ("fin_len", st.fin_len)):
                    debug.setdefault(key, []).append(val.clone())
        score(logits, 0)
        lens_rep = [s for s in lens for _ in range(K)]
        plen = ws.get("gen_beam_plen", (BK,), I32)
        plen.copy_(torch.tensor(lens_rep, dtype=I32), non_blocking=True)
        if T > 1:
            src = ws.get("gen_beam_src", (BK,), I32)           # beam b*K+k starts from the prompt rows of sequence BK + b
            src.copy_(torch.tensor([BK + b for b in range(Bn) for _ in range(K)], dtype=I32), non_blocking=True)
            for kv in (whole.k, whole.v):
                B.kv_copy_spans(kv, kv, BK, src_seq=src, n_t=plen)
            steps = T - 1
            pos_all = ws.get("gen_pos", (steps, BK), I32)
            len_all = ws.get("gen_len", (steps, BK), I32)
            sid = ws.get("gen_sid", (BK,), I32)
            pos_all.copy_(torch.tensor([[s + t for s in lens_rep] for t in range(steps)], dtype=I32), non_blocking=True)
            len_all.copy_(torch.tensor([[s + t + 1 for s in lens_rep] for t in range(steps)], dtype=I32), non_blocking=True)
            sid.copy_(torch.arange(BK, dtype=I32), non_blocking=True)
            tmp_k = ws.get("beam_tmp_k", (c.n_layers, BK, c.n_heads, steps, c.head_dim), BF16)
            tmp_v = ws.get("beam_tmp_v", tuple(tmp_k.shape), BF16)
            for t in range(steps):
                if t:                                          # the t positions generated so far follow their beam's parent
                    for kv, tmp in ((cache.k, tmp_k), (cache.v, tmp_v)):
                        B.kv_copy_spans(kv, tmp, BK, src_seq=st.parent, src_t0=plen, n_fixed=t)
                        B.kv_copy_spans(tmp, kv, BK, dst_t0=plen, n_fixed=t)
                lg = self.llama.decode_step(ws, cache, st.next_ids, pos_all[t], len_all[t], sid)
                score(lg, t + 1)
        best = st.fin_seq[:, 0].cpu().to(torch.int64)                                   # the only D2H pair of the call
        blen = st.fin_len[:, 0].cpu()
        ws.release("beam_tmp_k", "beam_tmp_v")      # 2 x layers x rows x heads x (T-1) x head_dim bf16: not kept between calls
        width = max(1, int(blen.max()))
        return GenerateResult(tokens=best[:, :width].contiguous(), first_logits=first)


class SalmonnRuntime(CausalLMRuntimeMixin):
    def __init__(self, cfg: SalmonnCfg, state_dict: Dict[str, torch.Tensor], device="cuda", consume: bool = False,
                 parts: Sequence[str] = ("whisper", "beats", "qformer", "llama")):
        if not torch.cuda.is_available():
            raise B.IclError("SalmonnRuntime needs a GPU: the HIP path has no CPU fallback")
        B.load_library()
        self.cfg = cfg
        self.lm_cfg = cfg.llama
        self.device = torch.device(device)
        sd = normalize_keys(state_dict) if not consume else state_dict
        self.ws = Workspace(self.device)
        self.whisper = self.beats = self.qformer = self.llama = None
        if "whisper" in parts:
            self.logmel = LogMel(cfg.whisper.n_mels, self.device)
            self.whisper = WhisperEncoderHIP(pack_whisper(sd, cfg.whisper, self.device, consume=consume))
        if "beats" in parts and cfg.beats is not None:
            self.beats = BeatsHIP(pack_beats(sd, cfg.beats, self.device, consume=consume), self.device)
        if "qformer" in parts:
            self.qformer = SpeechQFormerHIP(pack_qformer(sd, cfg.qformer, self.device, consume=consume),
                                            cfg.whisper.d_model, cfg.beats.d_model if self.beats is not None else 0,
                                            cfg.llama.hidden)
        if "llama" in parts:
            self.llama = LlamaHIP(pack_llama(sd, cfg.llama, self.device, consume=consume), self.device)
        self._graphs, self._graph_warm, self._graph_gen = {}, set(), 0

    # --------------------------------------------------------------------------------------------
    # K1-K8: SALMONN.encode_speech
    # --------------------------------------------------------------------------------------------
    @property
    def tokens_per_audio(self) -> int:
        return self.qformer.n_windows(1500)

    def encode_speech(self, raw_wav: Optional[torch.Tensor], wav_lens: Optional[Sequence[int]] = None,
                      spectrogram: Optional[torch.Tensor] = None, padded_lens: Optional[Sequence[int]] = None) -> torch.Tensor:
        """raw_wav f32 [n, L] (zero padded; host or device), wav_lens valid samples per audio.
        spectrogram (optional, f32 [n, n_mels, 3000]): use the caller's log-mel instead of K1.
        padded_lens: length the reference's BEATs would see per audio (defaults to wav_lens: batch-1 semantics).
        Returns speech embeddings f32 [n, 88, H_llm] on the device (atts are all ones in the reference); with a clip longer than
        30 s in the batch: [n, W_max, H_llm] and ``last_audio_windows`` = tokens per audio."""
        ws, dev = self.ws, self.device
        if raw_wav is not None:
            raw_wav = raw_wav.to(device=dev, dtype=F32)
            if raw_wav.dim() == 1:
                raw_wav = raw_wav[None]
            raw_wav = raw_wav.contiguous()
            n = raw_wav.shape[0]
            if wav_lens is None:
                wav_lens = [raw_wav.shape[1]] * n
            wav_lens = [int(x) for x in wav_lens]
        else:
            n = spectrogram.shape[0]
        if spectrogram is not None:
            xt = self.logmel.from_spectrogram(ws, spectrogram.to(device=dev, dtype=F32))
        else:
            xt, _ = self.logmel(ws, raw_wav, _i32(wav_lens, dev))
        speech = self.whisper.forward(ws, xt)
        audio = cu = None
        if self.beats is not None and raw_wav is not None:
            padded = [int(x) for x in (padded_lens if padded_lens is not None else wav_lens)]
            audio, cu, _ = self.beats.forward(ws, raw_wav, padded, wav_lens)
        out = self.qformer.forward(ws, speech, n, audio, cu)
        # tokens per audio: 88 up to 30 s; a longer clip keeps its extra BEATs frames (SALMONN pads the shorter stream) and gets
        # more windows — rows past last_audio_windows[a] of audio a are zero padding, not tokens
        self.last_audio_windows = list(self.qformer.last_windows)
        return out.view(n, max(self.last_audio_windows), self.cfg.llama.hidden)

    def log_mel(self, raw_wav: torch.Tensor, wav_lens: Sequence[int]) -> torch.Tensor:
        """K1 alone: f32 [n, n_mels, 3000] (the ``spectrogram`` batch key of the reference's processor)."""
        raw_wav = raw_wav.to(device=self.device, dtype=F32).contiguous()
        _, spec = self.logmel(self.ws, raw_wav, _i32([int(x) for x in wav_lens], self.device), want_spec=True)
        return spec

