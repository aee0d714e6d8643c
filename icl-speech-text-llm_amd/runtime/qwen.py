"""QwenAudioRuntime — MI355X-native stand-in for ``Qwen2AudioForConditionalGeneration`` as the reference drives it
(models/custom_qwen.py:51-55 construction, :186-195 forward, :228-234 generate): audio tower + projector (K13), audio
features scattered into the ``<|AUDIO|>`` token slots (one K9 gather), Qwen2 decoder prefill + greedy decode (K14).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import binding as B
from .config import QwenAudioCfg
from .engines import BF16, F32, I32, LlamaHIP, LogMel, QwenAudioTowerHIP, WhisperEncoderHIP, Workspace, _i32
from .packing import _bf, _f32, _take, pack_llama, pack_whisper
from .salmonn import CausalLMRuntimeMixin, speech_segment


def normalize_qwen_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """peft / DDP / transformers-4 / transformers-5 spellings -> the canonical (transformers-4 era, i.e. the reference's)
    names ``audio_tower.*``, ``multi_modal_projector.*``, ``language_model.model.*``, ``language_model.lm_head.weight``."""
    out = {}
    tops = ("audio_tower.", "language_model.", "multi_modal_projector.")
    for k, v in sd.items():
        changed = True
        while changed:
            changed = False
            for junk in ("module.", "base_model.model."):
                if k.startswith(junk):
                    k, changed = k[len(junk):], True
        if k.startswith("model.") and k[len("model."):].startswith(tops):
            k = k[len("model."):]
        k = k.replace(".base_layer.", ".").replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")
        if k.startswith("language_model.") and not k.startswith(("language_model.model.", "language_model.lm_head.")):
            k = "language_model.model." + k[len("language_model."):]
        if k == "lm_head.weight":
            k = "language_model.lm_head.weight"
        out[k] = v
    return out


class QwenAudioRuntime(CausalLMRuntimeMixin):
    def __init__(self, cfg: QwenAudioCfg, state_dict: Dict[str, torch.Tensor], device="cuda", consume: bool = False):
        if not torch.cuda.is_available():
            raise B.IclError("QwenAudioRuntime needs a GPU: the HIP path has no CPU fallback")
        B.load_library()
        self.cfg, self.lm_cfg = cfg, cfg.llm
        self.device = torch.device(device)
        sd = state_dict if consume else normalize_qwen_keys(state_dict)
        self.ws = Workspace(self.device)
        self.logmel = LogMel(cfg.audio.n_mels, self.device)
        enc = WhisperEncoderHIP(pack_whisper(sd, cfg.audio, self.device, prefix="audio_tower.", consume=consume))
        self.tower = QwenAudioTowerHIP(enc, _bf(_take(sd, "multi_modal_projector.linear.weight", consume), self.device),
                                       _f32(_take(sd, "multi_modal_projector.linear.bias", consume), self.device), cfg.llm.hidden)
        self.llama = LlamaHIP(pack_llama(sd, cfg.llm, self.device, prefix="language_model.", consume=consume), self.device)
        self._graphs, self._graph_warm, self._graph_gen = {}, set(), 0

    # ---- K13 ---------------------------------------------------------------------------------------
    def encode_audio(self, input_features: Optional[torch.Tensor] = None, mel_lens: Optional[Sequence[int]] = None,
                     raw_wav: Optional[torch.Tensor] = None, wav_lens: Optional[Sequence[int]] = None):
        """input_features [n, 128, 3000] (+ valid mel frames per audio), or raw 16 kHz wav [n, L] (+ sample counts: the
        log-mel then runs on the GPU).  Returns (features f32 [n, 750, H], valid feature rows per audio)."""
        if input_features is not None:
            spec = input_features.to(device=self.device, dtype=F32)
            xt = self.logmel.from_spectrogram(self.ws, spec)
            n = spec.shape[0]
            mel_lens = [3000] * n if mel_lens is None else [int(x) for x in mel_lens]
        else:
            wav = raw_wav.to(device=self.device, dtype=F32).contiguous()
            n = wav.shape[0]
            wav_lens = [int(x) for x in (wav_lens if wav_lens is not None else [wav.shape[1]] * n)]
            xt, _ = self.logmel(self.ws, wav, _i32(wav_lens, self.device))
            mel_lens = [min(3000, -(-L // 160)) for L in wav_lens]     # feature-extractor mask, sub-sampled every hop
        feats, out_lens = self.tower.forward(self.ws, xt, mel_lens)
        return feats.view(n, 750, self.cfg.llm.hidden), out_lens

    def logmel_features(self, wav: torch.Tensor, wav_lens: Sequence[int]) -> torch.Tensor:
        """K1 with 128 mel bins: f32 [n, 128, 3000] (the ``input_features`` batch key of the reference's Qwen processor)."""
        wav = wav.to(device=self.device, dtype=F32).contiguous()
        _, spec = self.logmel(self.ws, wav, _i32([int(x) for x in wav_lens], self.device), want_spec=True)
        return spec

    def segments_from_ids(self, input_ids: Sequence[Sequence[int]], out_lens: Sequence[int]) -> List[list]:
        """Replace each run of ``<|AUDIO|>`` ids by a speech segment over the next audio's valid rows (HF masked_scatter
        order: audios are consumed in order of appearance, row-major over the batch; modeling_qwen2_audio.py:727-742)."""
        aid, segs_all, audio = self.cfg.audio_token_id, [], 0
        for ids in input_ids:
            segs, run, i = [], [], 0
            ids = [int(t) for t in ids]
            while i < len(ids):
                if ids[i] == aid:
                    if run:
                        segs.append(run); run = []
                    j = i
                    while j < len(ids) and ids[j] == aid:
                        j += 1
                    count = j - i
                    while count > 0:      # a run may cover several consecutive audios
                        if audio >= len(out_lens):
                            raise ValueError("more <|AUDIO|> tokens than audio features")
                        take = out_lens[audio]
                        if take > count:
                            raise ValueError(f"audio {audio} yields {take} features but only {count} <|AUDIO|> tokens remain")
                        segs.append(speech_segment(audio * 750, take))
                        count -= take
                        audio += 1
                    i = j
                else:
                    run.append(ids[i]); i += 1
            if run:
                segs.append(run)
            segs_all.append(segs)
        if audio != len(out_lens):
            raise ValueError(f"Audio features and audio tokens do not match: {len(out_lens)} audios, {audio} consumed")
        return segs_all
